#!/bin/bash
# A/B of the one-launch AttentionBlock (otvae_attn_stage_fwd / _bwd) on the headline step and the side workloads, one box, one file:
# gpurun_out/attn_stage_ab.txt
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/attn_stage_ab.txt
ms() { python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
{
echo "# headline step (MNIST-32 CNN VAE, GaussianPrior, batch 1024), ms per captured step, two runs each, same box"
echo "three launches per AttentionBlock and direction (OTVAE_ATTN_STAGE=0):  $(OTVAE_ATTN_STAGE=0 ms)  $(OTVAE_ATTN_STAGE=0 ms)"
echo "one-launch forward, three-launch backward (OTVAE_ATTN_STAGE_BWD=0):    $(OTVAE_ATTN_STAGE_BWD=0 ms)  $(OTVAE_ATTN_STAGE_BWD=0 ms)"
echo "one-launch forward and backward, qkv written and read (OTVAE_ATTN_STAGE_QKV=1): $(OTVAE_ATTN_STAGE_QKV=1 ms)  $(OTVAE_ATTN_STAGE_QKV=1 ms)"
echo "one-launch forward and backward, q / k / v formed again in the backward kernel (default): $(ms)  $(ms)"
echo "# Sinkhorn-prior step (configs[2])"
echo "three launches: $(OTVAE_ATTN_STAGE=0 ms --workload sinkhorn)   default: $(ms --workload sinkhorn)"
echo "# side workloads, default switches"
python3 tools/w2_prior_bench.py 2>/dev/null | tail -2
python3 tools/cifar_bench.py 2>/dev/null | tail -2
python3 tools/vit_bench.py 2>/dev/null | tail -3
python3 tools/lightning_route_probe.py 2>/dev/null | tail -4
} > "$O" 2>&1
cat "$O"
