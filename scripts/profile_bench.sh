# rocprofv3 kernel stats of one bench.py command (run on the GPU box from the repo root): scripts/profile_bench.sh TAG [bench args]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT/gpurun_out"
rm -rf "$R/$TAG"
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$TAG" -o p -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" --no-cpu-baseline --no-kernel-timers > "$R/$TAG.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 scripts/trace_summary.py $(find "$R/$TAG" -name "*kernel_trace.csv") > "$R/${TAG}_summary.txt" || true
rm -f $(find "$R/$TAG" -name "*kernel_trace.csv")
cp $(find "$R/$TAG" -name "*kernel_stats.csv") "$R/${TAG}_kernel_stats.csv"
