import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ot_vae_lightning_amd.ot import matrix_utils as MU
D = int(os.environ.get("D", "1024"))
g = torch.Generator().manual_seed(D)
x = torch.randn(1, 3 * D, D, generator=g, dtype=torch.float64)
cg = (x.transpose(-1, -2) @ x / x.shape[-2]).cuda()
for _ in range(2):
    MU.eigh_vectors(cg)
torch.cuda.synchronize()
print("done")
