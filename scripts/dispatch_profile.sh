# per-dispatch time + HBM traffic of one training step (run on the GPU box from the repo root): scripts/dispatch_profile.sh <tag> [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT/gpurun_out/disp_$TAG"
rm -rf "$R"; mkdir -p "$R"
B="$GRAFT_REPO_ROOT/bench.py"
rocprofv3 --kernel-trace --output-format csv -d "$R/pA" -o a -- python3 "$B" "$@" --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timers > "$R/pA.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/pF" -o f -- python3 "$B" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > "$R/pF.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/pW" -o w -- python3 "$B" "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timers > "$R/pW.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 scripts/dispatch_table.py $(find "$R/pA" -name "*kernel_trace.csv") $(find "$R/pF" -name "*counter_collection.csv") $(find "$R/pW" -name "*counter_collection.csv") > "$R/table.txt"
rm -rf "$R/pA" "$R/pF" "$R/pW"
tail -3 "$R/table.txt"
