import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ot_vae_lightning_amd as A
from ot_vae_lightning_amd.ot import matrix_utils as MU
D, N = 128, 10000
g = torch.Generator().manual_seed(1)
m = torch.randn(D, D, generator=g, dtype=torch.float64)
zs = (torch.randn(N, D, generator=g, dtype=torch.float64) @ m.T + torch.randn(D, generator=g, dtype=torch.float64)).cuda()
mean_all = zs.mean(0)
cov = (zs - mean_all).T @ (zs - mean_all) / N
lam, vt = MU.eigh_vectors(cov)
print("cov eig min/max", float(lam.min()), float(lam.max()), "nan", bool(torch.isnan(lam).any()))
rt = MU.sqrtm(cov)
print("sqrtm nan", bool(torch.isnan(rt).any()))
mix = rt @ cov @ rt
lm, _ = MU.eigh_vectors(mix)
print("mix eig min/max", float(lm.min()), float(lm.max()), bool(torch.isnan(lm).any()))
print("cpu mix eig min", float(torch.linalg.eigvalsh(mix.cpu()).min()))
print("w2", float(A.w2_gaussian(mean_all, mean_all, cov, cov, make_pd=True)))
