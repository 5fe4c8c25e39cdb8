# copy the round-3 evidence of gpurun_out/r03_<tag>/ (scripts/refresh_profiles_r03.sh) and gpurun_out/r03_lines/ (scripts/bench_lines_r03.sh) into profiles/
set -e
for d in gpurun_out/r03_*/; do
  tag=$(basename "$d"); tag=${tag#r03_}
  [ "$tag" = "lines" ] && continue
  [ -f "$d/bench.json" ] || continue
  cp "$d/bench.json" profiles/r03_bench_$tag.json
  cp "$d/kernel_stats.csv" profiles/r03_${tag}_kernel_stats.csv
  cp "$d/trace_summary.txt" profiles/r03_${tag}_trace_summary.txt
  cp "$d/mfma_util.txt" profiles/r03_${tag}_mfma_util.txt
  cp "$d/pmc_traffic.json" profiles/r03_pmc_traffic_$tag.json
done
for f in gpurun_out/r03_lines/*.json; do [ -s "$f" ] && cp "$f" profiles/r03_bench_$(basename "$f"); done
ls profiles | grep r03
