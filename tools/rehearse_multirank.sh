#!/bin/bash
# N > 1 rehearsal of bench.py on a ONE-GPU box: every rank on GPU 0, the product path (mt_comm_init / mt_gather_returns in
# the C ABI) behind the shared-memory stand-in for librccl (tests/fake_rccl), gloo as control plane.  Checks the control
# flow and the JSON of an N > 1 invocation (headline + the strong_1m / config3 legs); NOT a scaling number.
# usage: tools/rehearse_multirank.sh <ranks> <out.json> [extra bench.py flags]
set -e
ranks=$1; out=$2; shift 2
lib=tests/fake_rccl/_build/libfake_rccl.so
if [ ! -f "$lib" ] || [ tests/fake_rccl/fake_rccl.cpp -nt "$lib" ]; then
  mkdir -p tests/fake_rccl/_build
  /opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared tests/fake_rccl/fake_rccl.cpp -o "$lib" -lrt
fi
port=$((20000 + RANDOM % 20000))
MT_RCCL_LIB=$PWD/$lib HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 \
  --nproc-per-node "$ranks" --master-addr 127.0.0.1 --master-port "$port" bench.py --gpus "$ranks" --steps 20 --warmup 5 \
  --single-device --backend gloo "$@" > "$out" 2> "${out%.json}.err"
python - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("rehearsal ok:", d["n_gpus"], "ranks, scaling", d["scaling"], "value %.3g" % d["value"], "legs", sorted(k for k in d.get("secondary", {}) if k != "note"))
PY
