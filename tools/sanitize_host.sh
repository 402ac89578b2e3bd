#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer run of the C ABI (sanitizers for the host shim; the
# GPU side cannot be sanitised on this pool).  Builds libmanytor_hip with -fsanitize=address,undefined for the HOST code
# only (-fno-gpu-sanitize) into /tmp and runs the CPU-side ABI tests against it through MT_LIB_OVERRIDE.
#   bash tools/sanitize_host.sh            (no GPU needed; on a GPU box add `-m gpu` tests by hand if wanted)
set -e
cd "$(dirname "$0")/.."
out=/tmp/libmanytor_hip_asan.so
/opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -I include \
    -fno-signed-zeros -ffinite-math-only -fno-slp-vectorize -ffp-contract=off \
    -fsanitize=address,undefined -fno-gpu-sanitize manytor_amd/csrc/engine.hip manytor_amd/csrc/comm.hip -o "$out"
asan=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD="$asan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 MT_LIB_OVERRIDE="$out"
# (the plain-C link test needs the sanitizer runtime at link time: not part of this run)
python -m pytest tests/test_host_api.py -q -x -k "not plain_c_program" "$@"
