import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ot_vae_lightning_amd.ot import matrix_utils as MU
torch.manual_seed(0)
for D in (144, 160, 129):
    x = torch.randn(3 * D, D, dtype=torch.float64)
    cov = (x.T @ x / x.shape[0])
    lam = torch.linalg.eigvalsh(cov)
    ev, vt = MU.eigh_vectors(cov.cuda())
    ev, vt = ev.cpu(), vt.cpu()
    print(D, "eig err", float((torch.sort(ev)[0] - lam).abs().max() / lam.max()),
          "orth", float((vt @ vt.T - torch.eye(D, dtype=torch.float64)).abs().max()),
          "recon", float((vt.T @ (ev.unsqueeze(-1) * vt) - cov).abs().max() / lam.max()),
          "sum ev", float(ev.sum()), "trace", float(cov.trace()))
