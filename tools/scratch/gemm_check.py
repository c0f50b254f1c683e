import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ot_vae_lightning_amd.ot import matrix_utils as MU
g = torch.Generator().manual_seed(3)
for (nb, m, n, k) in ((1, 1024, 1024, 1024), (2, 300, 257, 129), (1, 256, 64, 64), (3, 65, 700, 1000), (1, 512, 512, 512)):
    for ta in (False, True):
        for tb in (False, True):
            a = torch.randn((nb, k, m) if ta else (nb, m, k), generator=g, dtype=torch.float64).cuda()
            b = torch.randn((nb, n, k) if tb else (nb, k, n), generator=g, dtype=torch.float64).cuda()
            c = MU.matmul64(a, b, trans_a=ta, trans_b=tb)
            ref = (a.transpose(1, 2) if ta else a) @ (b.transpose(1, 2) if tb else b)
            err = float((c - ref).abs().max() / ref.abs().max())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                MU.matmul64(a, b, trans_a=ta, trans_b=tb)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print(f"nb={nb} m={m} n={n} k={k} tA={int(ta)} tB={int(tb)}: rel err {err:.2e}  {ms:.3f} ms  {2*nb*m*n*k/ms/1e9:.2f} TFLOP/s")
            assert err < 1e-13
