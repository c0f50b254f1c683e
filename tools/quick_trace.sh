#!/bin/bash
# kernel trace of the default bench command, reduced to the per-kernel replay table (gpurun_out/quick/profiles/<tag>_replay_kernel_stats.csv)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/quick
rm -rf "$O"; mkdir -p "$O/profiles"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline > "$O/trace_bench.json" 2> "$O/trace.err"
cd "$R"
export OTVAE_PROFILES_OUT="$O/profiles"
python3 tools/summarize_profiles.py --replay "$O/trace" ${1:-quick}
rm -rf "$O/trace"
