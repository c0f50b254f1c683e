import os, sys, ctypes as C, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ot_vae_lightning_amd import _lib as L
lib = L.load()
MAXS = 24
def run(A):
    D = A.shape[-1]
    a = A.double().cuda().reshape(1, D, D).contiguous()
    ws = torch.zeros(lib.otvae_eigh_ws(1, D), device="cuda", dtype=torch.uint8)
    ev = torch.empty(1, D, device="cuda", dtype=torch.float64)
    out = torch.empty_like(a)
    L.check(lib.otvae_eigh_fn(L.ptr(a), 1, D, 3, L.ptr(out), L.ptr(ev), L.ptr(ws), L.stream()), "eigh")
    torch.cuda.synchronize()
    n = (D + 1) & ~1; half = n // 2
    off = 4 * n * n * 8 + MAXS * n * half * 16
    ctl = ws[off:off + 16].cpu().numpy().view(np.int32)
    lam = torch.linalg.eigvalsh(A.double())
    err = float((torch.sort(ev.cpu()[0])[0] - lam).abs().max() / lam.abs().max())
    vt = out[0].cpu()
    ortho = float((vt @ vt.T - torch.eye(D, dtype=torch.float64)).abs().max())
    return ctl[:2].tolist(), err, ortho, bool(torch.isnan(ev).any())
g = torch.Generator().manual_seed(0)
for D in (8, 32, 64, 128):
    x = torch.randn(3 * D, D, generator=g, dtype=torch.float64)
    print(D, "well", run(x.T @ x / (3 * D)))
    m = torch.randn(D, D, generator=g, dtype=torch.float64)
    z = torch.randn(5000, D, generator=g, dtype=torch.float64) @ m.T
    print(D, "ill ", run(torch.cov(z.T)))
