import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ot_vae_lightning_amd.ot import matrix_utils as MU
g = torch.Generator().manual_seed(128)
x = torch.randn(2, 3072, 1024, generator=g, dtype=torch.float64)
cov = (x.transpose(-1, -2) @ x / 3072).cuda()
for _ in range(2):
    MU.eigh_vectors(cov)
torch.cuda.synchronize()
