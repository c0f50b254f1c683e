import os, sys, ctypes as C, torch, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ot_vae_lightning_amd import _lib as L
lib = L.load()
def run(A):
    D = A.shape[-1]
    a = A.double().cuda().reshape(1, D, D).contiguous()
    ws = torch.zeros(lib.otvae_eigh_ws(1, D), device="cuda", dtype=torch.uint8)
    ev = torch.empty(1, D, device="cuda", dtype=torch.float64)
    out = torch.empty_like(a)
    L.check(lib.otvae_eigh_fn(L.ptr(a), 1, D, 3, L.ptr(out), L.ptr(ev), L.ptr(ws), L.stream()), "eigh")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L.check(lib.otvae_eigh_fn(L.ptr(a), 1, D, 3, L.ptr(out), L.ptr(ev), L.ptr(ws), L.stream()), "eigh")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    Dp = (D + 15) // 16 * 16
    off = (2 * Dp * D + 2 * Dp + D * D) * 8
    ctl = ws[off:off + 16].cpu().numpy().view(np.int32)
    lam = torch.linalg.eigvalsh(A.double())
    err = float((torch.sort(ev.cpu()[0])[0] - lam).abs().max() / lam.abs().max())
    vt = out[0].cpu()
    ortho = float((vt @ vt.T - torch.eye(D, dtype=torch.float64)).abs().max())
    recon = float((vt.T @ (ev.cpu()[0][:, None] * vt) - A.double()).abs().max() / lam.abs().max())
    return ctl.tolist(), round(ms, 2), err, ortho, recon
g = torch.Generator().manual_seed(0)
for D in (256, 1024):
    x = torch.randn(3 * D, D, generator=g, dtype=torch.float64)
    print(D, "well", run(x.T @ x / (3 * D)))
    m = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    z = torch.randn(4 * D, D, generator=g, dtype=torch.float64) @ m.T
    print(D, "ill ", run(torch.cov(z.T)))
