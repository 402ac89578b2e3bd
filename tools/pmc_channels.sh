#!/bin/bash
# Per-channel (TCC instance x XCC) request / stall counters of the 4 194 304-arm step kernel on several fresh arenas of ONE
# process (tools/placement_modes.py): is the fast / slow placement visible as an imbalance between L2 / memory channels?
# Counters only (no tracing domains), one pass per group.  usage: tools/pmc_channels.sh <outdir>
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
export MT_PLACE_ALLOCS=6 MT_CHAINS=1
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format json csv -d "$out/$name" -- python3 tools/placement_modes.py > "$out/$name.times.json" 2> "$out/$name.err" || echo "pass $name failed"; }
run rd TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL
run wr TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL
run req TCC_REQ TCC_TAG_STALL
python3 tools/placement_modes.py > "$out/plain.times.json" 2> "$out/plain.err"
cat "$out"/*.times.json
ls -la "$out"/*/* | head -30
