#!/bin/bash
# PMC passes over tools/placement_modes.py (separate passes, counters only).  usage: tools/pmc_placement.sh <outdir>
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
rocprofv3 -L > "$out/counters_list.txt" 2>&1 || true
grep -i -o "\b[A-Z0-9_]*\(UTCL\|TLB\|EA0_RDREQ\|EA0_WRREQ\|TAG_STALL\|EA0_RD_UNCACHED\|MC_RDREQ\|BUBBLE\)[A-Z0-9_]*" "$out/counters_list.txt" | sort -u > "$out/counters_of_interest.txt" || true
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 tools/placement_modes.py > "$out/$name.json" 2> "$out/$name.err" || echo "pass $name failed"; }
run grbm GRBM_GUI_ACTIVE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
python3 tools/placement_modes.py > "$out/plain.json" 2> "$out/plain.err"
cat "$out/plain.json"
