#!/bin/bash
# Translation / latency counters of a slow arena against a fast one of the same kernel (bench.py's two passes over its secondary
# configurations), separate rocprofv3 --pmc passes, no trace domain.   usage: tools/pmc_placement_tlb.sh <outdir>
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() { name=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > "$out/$name.json" 2> "$out/$name.err"; }
run utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum
run grbm GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum
for k in "step_kernel<mt::Dh7Table" "step_kernel<mt::Ref4Table, true, 0, false, 8, true, false>"; do
  for p in utcl1 grbm lat ea; do python3 tools/pmc_two_populations.py "$out/$p" "$k"; done
done
python3 - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print(f, {k[:20]: [round(x, 1) for x in v["us_per_step_passes"]] for k, v in d["secondary"]["other_configs"].items()})
    except Exception as e:
        print(f, "unreadable", e)
PY
for p in utcl1 grbm lat ea; do find "$out/$p" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ; done
