# Collect the judged evidence for bench.py's default workload (run on the GPU box from the repo root):
#   rocprofv3 kernel stats + per-launch summary, PMC FETCH_SIZE / WRITE_SIZE passes, the plain bench line.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
R="$GRAFT_REPO_ROOT/gpurun_out"
rm -rf "$R/pA" "$R/pF" "$R/pW"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/pA" -o a -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$R/pA.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/pF" -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$R/pF.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/pW" -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$R/pW.log" 2>&1
python scripts/pmc_traffic.py $(find "$R/pF" -name "*counter_collection.csv") $(find "$R/pW" -name "*counter_collection.csv") "$R/pmc_traffic_new.json"
rm -f $(find "$R/pF" "$R/pW" -name "*counter_collection.csv")
python scripts/trace_summary.py $(find "$R/pA" -name "*kernel_trace.csv") > "$R/pA_summary.txt"
rm -f $(find "$R/pA" -name "*kernel_trace.csv")
cp "$R/pmc_traffic_new.json" profiles/r01_pmc_traffic_b32_s512_bf16.json   # bench.py quotes roofline.traffic from this file
# (on a gpurun box only gpurun_out/ travels back: copy gpurun_out/pmc_traffic_new.json, pA/*kernel_stats.csv, pA_summary.txt and
#  the bench lines into profiles/ in the repository afterwards)
python bench.py > "$R/bench_new.log" 2>&1
tail -1 "$R/bench_new.log" | cut -c1-200
