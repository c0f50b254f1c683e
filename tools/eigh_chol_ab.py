"""A/B of the D <= 128 eigensolver with / without the in-LDS Cholesky start (run twice: OTVAE_EIGH_NO_CHOL unset / =1):
time per call and residuals against the fp64 input on well- and ill-conditioned covariances."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ot_vae_lightning_amd.ot import matrix_utils as MU  # noqa: E402


def t_gpu(fn, reps=30):
    """median of per-call device times (HIP events): a lone workgroup lets the clocks wander, means are useless here"""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    tag = "no-chol" if os.environ.get("OTVAE_EIGH_NO_CHOL") else "chol"
    for D in ([int(a) for a in sys.argv[1:]] or (8, 31, 32, 64, 65, 96, 127, 128)):
        for cond in (1e1, 1e6, 1e12):
            g = torch.Generator().manual_seed(D)
            q, _ = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))
            lam = torch.logspace(0, -torch.log10(torch.tensor(cond)).item(), D, dtype=torch.float64)
            cov = (q * lam) @ q.T
            cov = 0.5 * (cov + cov.T)
            cg = cov.cuda()[None]
            w, v = MU.eigh_vectors(cg)
            w, v = w[0].cpu(), v[0].cpu().T.contiguous()
            res = ((cov @ v - v * w).norm() / cov.norm()).item()
            orth = (v.T @ v - torch.eye(D, dtype=torch.float64)).abs().max().item()
            rel = ((w.sort().values - lam.sort().values).abs() / lam.sort().values).max().item()
            ms = t_gpu(lambda: MU.eigh_vectors(cg))
            print(f"[{tag}] D={D:4d} cond={cond:7.0e}: {ms:7.3f} ms  residual {res:.2e}  orth {orth:.2e}  max rel eigval err {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
