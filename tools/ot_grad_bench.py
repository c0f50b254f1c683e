"""otvae_ot_cost_grad (fp32 MFMA, hand-written) against the same arithmetic with the library GEMM (`pi @ y` through
hipBLASLt/rocBLAS + ATen elementwise), at the benchmark's 1024 x 1024 x 128 and the per-GPU 256 x 256 x 256 of configs[3].
Both are timed as 20 launches captured into one hipGraph (HIP events around 5 replays).  Also otvae_sqdist_max against
torch.cdist(...)**2."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ot_vae_lightning_amd import _lib as L  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    lib = L.load()
    for n, m, d in ((1024, 1024, 128), (256, 256, 256), (2048, 2048, 256)):
        gen = torch.Generator().manual_seed(1)
        z = torch.randn(n, d, generator=gen).cuda()
        y = torch.randn(m, d, generator=gen).cuda()
        pi = torch.rand(n, m, generator=gen).cuda() / (n * m)
        g = torch.full((n,), 1.0 / n, device="cuda")
        gz = torch.empty_like(z)
        st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

        def ours():
            L.check(lib.otvae_ot_cost_grad(0, L.ptr(z), L.ptr(y), L.ptr(pi), L.ptr(g), n, 1.0, None, n, m, d, L.ptr(gz), st()), "grad")

        def library():
            return 2.0 * g.sum() * (pi.sum(1, keepdim=True) * z - pi @ y)

        ours()
        ref = library()
        err = float((gz - ref).abs().max() / ref.abs().max())
        t_ours, t_lib = timed(ours), timed(library)
        t_gemm = timed(lambda: pi @ y)
        C = torch.empty(n, m, device="cuda")
        pm = torch.empty(lib.otvae_sqdist_max_parts(0, n, m), device="cuda")
        t_sq = timed(lambda: L.check(lib.otvae_sqdist_max(0, L.ptr(z), L.ptr(y), 1, n, m, d, L.ptr(C), L.ptr(pm), st()), "sqdist"))
        t_cd = timed(lambda: torch.cdist(z, y) ** 2)
        print(f"N={n} M={m} D={d}: otvae_ot_cost_grad {t_ours:.1f} us | library (GEMM + elementwise) {t_lib:.1f} us, GEMM alone {t_gemm:.1f} us"
              f" | rel err {err:.1e} || otvae_sqdist_max {t_sq:.1f} us | torch.cdist**2 {t_cd:.1f} us", flush=True)


if __name__ == "__main__":
    main()
