# copy the round-4 evidence of gpurun_out/r04_<tag>/ (scripts/refresh_profiles_r04.sh) and gpurun_out/r04_lines/ (scripts/bench_lines_r04.sh) into profiles/
set -e
for d in gpurun_out/r04_*/; do
  tag=$(basename "$d"); tag=${tag#r04_}
  [ "$tag" = "lines" ] && continue
  [ -f "$d/bench.json" ] || continue
  cp "$d/bench.json" profiles/r04_bench_$tag.json
  cp "$d/kernel_stats.csv" profiles/r04_${tag}_kernel_stats.csv
  cp "$d/trace_summary.txt" profiles/r04_${tag}_trace_summary.txt
  cp "$d/mfma_util.txt" profiles/r04_${tag}_mfma_util.txt
  cp "$d/pmc_traffic.json" profiles/r04_pmc_traffic_$tag.json
done
for d in gpurun_out/disp_*/; do [ -s "$d/table.txt" ] && cp "$d/table.txt" profiles/r04_dispatch_table_$(basename "$d" | sed s/disp_//).txt; done
for f in gpurun_out/r04_lines/*.json; do [ -s "$f" ] && cp "$f" profiles/r04_bench_$(basename "$f"); done
ls profiles | grep r04
