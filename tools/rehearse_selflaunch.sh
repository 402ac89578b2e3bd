#!/bin/bash
# N > 1 rehearsal of `python3 bench.py --gpus N --steps 20 --warmup 5` AS THE DRIVER RUNS IT -- no torchrun, a clean
# environment: bench.py starts its own N ranks (bench.self_launch) -- on a ONE-GPU box: every rank on GPU 0, the product path
# (mt_comm_init / mt_gather_returns in the C ABI) behind the shared-memory stand-in for librccl (tests/fake_rccl), gloo as
# control plane.  Checks launcher, control flow and the JSON of an N > 1 line; NOT a scaling number.
# usage: tools/rehearse_selflaunch.sh <ranks> <out.json> [extra bench.py flags]
set -e
ranks=$1; out=$2; shift 2
lib=tests/fake_rccl/_build/libfake_rccl.so
if [ ! -f "$lib" ] || [ tests/fake_rccl/fake_rccl.cpp -nt "$lib" ]; then
  mkdir -p tests/fake_rccl/_build
  /opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared tests/fake_rccl/fake_rccl.cpp -o "$lib" -lrt
fi
unset WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT
MT_RCCL_LIB=$PWD/$lib timeout -k 10 900 python3 bench.py --gpus "$ranks" --steps 20 --warmup 5 --single-device --backend gloo "$@" \
  > "$out" 2> "${out%.json}.err"
python3 - "$out" <<'PY'
import json, sys
lines = open(sys.argv[1]).read().splitlines()
assert len(lines) == 1, f"{len(lines)} lines on stdout"
d = json.loads(lines[0])
legs = sorted(k for k in d.get("secondary", {}) if k != "note")
print("self-launch rehearsal ok:", d["n_gpus"], "ranks, scaling", d["scaling"], "envs_total", d["config"]["envs_total"],
      "value %.3g" % d["value"], "device timeline %.3g" % d["value_device_timeline"], "legs", legs,
      "cpu_baseline" in d, d["config"]["launcher"])
PY
