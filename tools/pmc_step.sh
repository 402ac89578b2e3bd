#!/bin/bash
# Collect PMC counters for the step kernel (separate passes, no tracing domains mixed in).
# usage: tools/pmc_step.sh <outdir> [bench args...]
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --no-cpu-baseline --steps 60 --warmup 10 $BENCH_ARGS > "$out/$name.json" 2> "$out/$name.err"; }
BENCH_ARGS="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run grbm GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 tools/pmc_summary.py "$out"
