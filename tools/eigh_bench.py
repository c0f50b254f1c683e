"""Eigensolver timing: one-sided (default) vs two-sided (OTVAE_EIGH_TWOSIDED=1) vs torch.linalg.eigh on the host cores."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ot_vae_lightning_amd.ot import matrix_utils as MU  # noqa: E402


def t_gpu(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cases = ((32, 1), (64, 1), (128, 1), (128, 8), (256, 1), (512, 1), (1024, 1))
    if len(sys.argv) > 1:  # python tools/eigh_bench.py 1024 [nb]
        cases = ((int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1),)
    for D, nb in cases:
        g = torch.Generator().manual_seed(D)
        x = torch.randn(nb, 3 * D, D, generator=g, dtype=torch.float64)
        cov = x.transpose(-1, -2) @ x / x.shape[-2]
        cg = cov.cuda()
        os.environ.pop("OTVAE_EIGH_TWOSIDED", None)
        one = t_gpu(lambda: MU.eigh_vectors(cg))
        two = float("nan")
        if D <= 128:
            os.environ["OTVAE_EIGH_TWOSIDED"] = "1"
            two = t_gpu(lambda: MU.eigh_vectors(cg), reps=2)
            os.environ.pop("OTVAE_EIGH_TWOSIDED", None)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.linalg.eigh(cov)
        cpu = (time.perf_counter() - t0) / 3 * 1e3
        print(f"D={D:5d} nb={nb}: default {one:8.3f} ms | two-sided {two:8.3f} ms | torch.linalg.eigh CPU {cpu:8.3f} ms", flush=True)


if __name__ == "__main__":
    main()
