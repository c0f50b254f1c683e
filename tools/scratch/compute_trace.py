"""GaussianTransport.compute at D = 128 a few times (for a kernel trace: where do the 7.7 ms go?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ot_vae_lightning_amd as A

D = int(os.environ.get("D", "128"))
g = torch.Generator().manual_seed(1)
src = torch.randn(1024, D, generator=g).cuda()
tgt = (torch.randn(1024, D, generator=g) * 1.5 + 0.3).cuda()
op = A.GaussianTransport(D, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double), transport_cfg=dict(make_pd=True)).cuda()
op.update(source_samples=src, target_samples=tgt)
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    op.compute()
    torch.cuda.synchronize()
    print("compute %.3f ms" % ((time.perf_counter() - t0) * 1e3))
