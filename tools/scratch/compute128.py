import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ot_vae_lightning_amd as A
g = torch.Generator().manual_seed(1)
op = A.GaussianTransport(128, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double),
                         transport_cfg=dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)).cuda()
op.update(source_samples=(torch.randn(1024, 128, generator=g) * 1.5 + 0.3).cuda(), target_samples=torch.randn(1024, 128, generator=g).cuda())
for _ in range(3):
    op.compute()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    op.compute()
torch.cuda.synchronize()
print("compute ms", (time.perf_counter() - t0) / 5 * 1e3)
