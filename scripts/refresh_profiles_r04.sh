# Round-4 evidence for one bench.py workload (run on the GPU box from the repo root):
#   scripts/refresh_profiles_r02.sh <tag> [bench args...]   e.g.  scripts/refresh_profiles_r02.sh resnet50unet --net Resnet50Unet
# writes under gpurun_out/r04_<tag>/: kernel_stats.csv + trace_summary.txt (rocprofv3 --kernel-trace --stats), pmc_traffic.json
# (FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied by scripts/pmc_traffic.py), mfma_util.txt (SQ / GRBM pass), bench.json
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT/gpurun_out/r04_$TAG"
rm -rf "$R"; mkdir -p "$R"
B="$GRAFT_REPO_ROOT/bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/pA" -o a -- python3 "$B" "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timers > "$R/pA.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/pF" -o f -- python3 "$B" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > "$R/pF.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/pW" -o w -- python3 "$B" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > "$R/pW.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$R/pS" -o s -- python3 "$B" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > "$R/pS.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 scripts/pmc_traffic.py $(find "$R/pF" -name "*counter_collection.csv") $(find "$R/pW" -name "*counter_collection.csv") "$R/pmc_traffic.json" > "$R/pmc_traffic.txt"
python3 scripts/pmc_mfma.py $(find "$R/pS" -name "*counter_collection.csv") "$R/mfma_util.txt" > /dev/null
python3 scripts/trace_summary.py $(find "$R/pA" -name "*kernel_trace.csv") > "$R/trace_summary.txt"
cp $(find "$R/pA" -name "*kernel_stats.csv") "$R/kernel_stats.csv"
rm -rf "$R/pA" "$R/pF" "$R/pW" "$R/pS"
python3 bench.py "$@" > "$R/bench.log" 2>&1
tail -1 "$R/bench.log" > "$R/bench.json"
cut -c1-300 "$R/bench.json"
