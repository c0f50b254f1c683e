#!/bin/bash
# Collects everything profiles/ is built from, on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final/trace      rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/final/pmc_*      three separate PMC passes (MI355X_MICROARCH.md: one counter group per pass)
#   gpurun_out/final/bench*.json  un-profiled bench lines (MNIST/Gaussian default, Sinkhorn workload)
#   gpurun_out/final/pytest_gpu.log
# Each step stops the script when it fails.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- python3 "$R/bench.py" > "$O/trace_bench.json" 2> "$O/trace.err"
echo "[final] trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/pmc_fetch" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$O/pmc_fetch.json" 2> "$O/pmc_fetch.err"
echo "[final] pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/pmc_write" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$O/pmc_write.json" 2> "$O/pmc_write.err"
echo "[final] pmc write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_mfma" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$O/pmc_mfma.json" 2> "$O/pmc_mfma.err"
echo "[final] pmc mfma done"
cd "$R"
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
echo "[final] bench done"
python3 bench.py --workload sinkhorn --no-cpu-baseline > "$O/bench_sinkhorn.json" 2> "$O/bench_sinkhorn.err"
echo "[final] sinkhorn bench done"
python3 -m pytest tests -m gpu -x -q > "$O/pytest_gpu.log" 2>&1
tail -2 "$O/pytest_gpu.log"
cp gpurun_out/parity_report.txt "$O/parity_report.txt" 2>/dev/null || true
