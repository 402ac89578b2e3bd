#!/bin/bash
# Round profile: kernel trace + stats of the driver's own bench command, then the PMC passes (separate runs).
# usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p "$out"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_under_trace.json" 2> "$out/trace.err"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
head -12 "$out/kernel_stats.csv"
trace_csv=$(find "$out/trace" -name "*kernel_trace.csv" | head -1)
python3 tools/trace_summary.py "$trace_csv" --union "step_kernel<mt::Ref4Table, true, 0, false, 8, true, true, false>" 524288 2 > "$out/kernel_stats_by_grid.csv"
tail -1 "$out/kernel_stats_by_grid.csv"
bash tools/pmc_step.sh "$out/pmc_d4" --no-secondary > "$out/pmc_d4.log" 2>&1 || echo "pmc d4 failed"
bash tools/pmc_step.sh "$out/pmc_d7" --no-secondary --dof 7 > "$out/pmc_d7.log" 2>&1 || echo "pmc d7 failed"
bash tools/pmc_step.sh "$out/pmc_131072" --no-secondary --envs-per-gpu 131072 > "$out/pmc_131072.log" 2>&1 || echo "pmc 131072 failed"
tail -12 "$out/pmc_d4.log"; tail -12 "$out/pmc_d7.log"; tail -24 "$out/pmc_131072.log"
# gpurun merges at most 64 MiB back: keep the summaries, drop the raw rocprofv3 output
for d in pmc_d4 pmc_d7 pmc_131072; do find "$out/$d" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} + ; done
rm -rf "$out/trace"
du -sh "$out"
