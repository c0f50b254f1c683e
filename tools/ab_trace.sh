#!/bin/bash
# Per-kernel A/B of one environment switch on ONE box: two rocprofv3 kernel traces of the same bench command, reduced to per-kernel
# replay statistics.   usage: tools/ab_trace.sh NAME=VALUE_A NAME=VALUE_B   (run through gpurun from the repo root)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/ab
rm -rf "$O"; mkdir -p "$O/profiles"
cd /tmp && export TMPDIR=/tmp
for side in a b; do
  if [ $side = a ]; then export "$1"; else export "$2"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$side" -- python3 "$R/bench.py" --steps 30 --warmup 5 --no-cpu-baseline > "$O/bench_$side.json" 2> "$O/trace_$side.err"
  OTVAE_PROFILES_OUT="$O/profiles" python3 "$R/tools/summarize_profiles.py" --replay "$O/trace_$side" ab_$side
  rm -rf "$O/trace_$side"
done
echo "[ab] done"
