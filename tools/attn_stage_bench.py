"""The AttentionBlock's forward at the benchmark's stage shapes (batch 1024), isolated: one fused launch (``otvae_attn_stage_fwd``) against
the three launches it replaces (1x1 qkv convolution with the BatchNorm prologue, attention, 1x1 projection with the residual sum and the
statistics epilogue).  Every variant is 20 dependent repetitions of the module's forward inside one hipGraph, timed with HIP events on
the launch stream; us per repetition.  (The backward pass is compared inside the captured training step: profiles/r03_attn_stage_ab.txt.)
Usage: python tools/attn_stage_bench.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import ot_vae_lightning_amd as A  # noqa: E402
from ot_vae_lightning_amd import functional as HF  # noqa: E402
from ot_vae_lightning_amd.networks.cnn import AttentionBlock  # noqa: E402

STAGES = [(32, 1, 1), (16, 8, 4), (8, 16, 4), (4, 32, 8), (2, 64, 8)]  # (side, width, heads) of the MNIST-32 autoencoder's blocks


def graph_time(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(iters):
            fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * iters) * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    print(f"# batch {n}; us per AttentionBlock forward (BatchNorm statistics pass + finalize included on both sides), 20 repetitions per hipGraph")
    print(f"{'stage':<28}{'one launch':>12}{'three launches':>16}")
    for side, width, heads in STAGES:
        blk = AttentionBlock(width, heads=heads, normalization="batchnorm").cuda().train()
        x = HF.as_nhwc(torch.randn(n, width, side, side, device="cuda"))
        res = HF.as_nhwc(torch.randn(n, width, side, side, device="cuda"))
        row = []
        with torch.no_grad():
            for fused in (True, False):
                HF.ATTN_STAGE = fused
                row.append(graph_time(lambda: blk(x, residual=res)))
            HF.ATTN_STAGE = True
            fused_ok = HF.attention_stage(x, blk.qkv.branch(None, False), heads, blk.proj_out.branch(res, True), training=False) is not None
        name = f"{side}x{side}, width {width}, {heads} heads" + ("" if fused_ok else " *")
        print(f"{name:<28}{row[0]:>12.1f}{row[1]:>16.1f}")
    print("* shape the fused kernel does not take: both columns are the three launches")


if __name__ == "__main__":
    main()
