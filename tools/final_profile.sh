#!/bin/bash
# Collects everything profiles/ is built from, on the GPU box (run through gpurun from the repo root):
#   gpurun_out/final/trace        rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/final/pmc_*        separate PMC passes (MI355X_MICROARCH.md: one counter group per pass): FETCH_SIZE, WRITE_SIZE,
#                                 MFMA busy, SQ instruction / wait counters
#   gpurun_out/final/trace_sk     kernel trace of the Sinkhorn workload (configs[2])
#   gpurun_out/final/bench*.json  un-profiled bench lines (MNIST/Gaussian default, Sinkhorn workload)
#   gpurun_out/final/*.txt        OT path timings, eigensolver timings, OT gradient micro-benchmark, per-call conv table
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
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$O/pmc_sq" -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$O/pmc_sq.json" 2> "$O/pmc_sq.err"
echo "[final] pmc sq done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_sk" -- python3 "$R/bench.py" --workload sinkhorn --no-cpu-baseline --steps 20 --warmup 5 > "$O/trace_sk.json" 2> "$O/trace_sk.err"
echo "[final] sinkhorn trace done"
cd "$R"
python3 bench.py > "$O/bench.json" 2> "$O/bench.err"
echo "[final] bench done"
python3 bench.py --workload sinkhorn --no-cpu-baseline > "$O/bench_sinkhorn.json" 2> "$O/bench_sinkhorn.err"
echo "[final] sinkhorn bench done"
python3 tests/ot_paths_timing.py > "$O/ot_paths.txt" 2>&1
python3 tools/eigh_bench.py > "$O/eigh_bench.txt" 2>&1
python3 tools/ot_grad_bench.py > "$O/ot_grad_bench.txt" 2>&1
python3 tools/conv_jobs_table.py > "$O/conv_jobs_table.txt" 2>&1
echo "[final] side timings done"
python3 -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1
tail -2 "$O/pytest_gpu.log"
cp gpurun_out/parity_report.txt "$O/parity_report.txt" 2>/dev/null || true
# reduce the raw traces here (gpurun merges at most 64 MiB back) and drop them
export OTVAE_PROFILES_OUT="$O/profiles"; mkdir -p "$OTVAE_PROFILES_OUT"
T=${OTVAE_ROUND_TAG:-r04}
python3 tools/summarize_profiles.py "$O/trace" ${T}_final
python3 tools/summarize_profiles.py --replay "$O/trace" ${T}_final
python3 tools/summarize_profiles.py --replay "$O/trace_sk" ${T}_sinkhorn_wl
python3 tools/summarize_profiles.py --pmc "$O" ${T}
python3 tools/summarize_profiles.py --pmc-sq "$O/pmc_sq" ${T}
python3 tools/summarize_profiles.py --roofline ${T}
cp "$O/bench.json" "$OTVAE_PROFILES_OUT/${T}_bench.json"
cp "$O/bench_sinkhorn.json" "$OTVAE_PROFILES_OUT/${T}_bench_sinkhorn.json"
for f in ot_paths eigh_bench ot_grad_bench conv_jobs_table parity_report; do cp "$O/$f.txt" "$OTVAE_PROFILES_OUT/${T}_$f.txt"; done
rm -rf "$O/trace" "$O/trace_sk" "$O"/pmc_fetch "$O"/pmc_write "$O"/pmc_mfma "$O"/pmc_sq
echo "[final] summaries written to $OTVAE_PROFILES_OUT"
