import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from ot_vae_lightning_amd import _lib as L
lib = L.load()
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
def run(n, cs, cn, hs, k, s, p, up, reps=200):
    ho = (hs * up + 2 * p - k) // s + 1
    x = nhwc(torch.randn(n, cs, hs, hs, device="cuda")); w = torch.randn(k, k, cs, cn, device="cuda")
    sc = torch.rand(cs, device="cuda") + 0.5; sh = torch.randn(cs, device="cuda")
    y = nhwc(torch.empty(n, cn, ho, ho, device="cuda"))
    g = L.ConvGeom(n, hs, hs, cs, up, ho, ho, cn, k, k, s, p)
    f = lambda: L.check(lib.otvae_conv_fwd(C.byref(g), L.ptr(x), L.ptr(sc), L.ptr(sh), 1, L.ptr(w), None, None, L.ptr(y), None, L.stream()), "f")
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): f()
    gr.replay(); torch.cuda.synchronize()
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("dbg", os.environ.get("OTVAE_DBG"), "64->64 3x3 @2x2: %.2f us | 256->256 1x1map 3x3: %.2f | 32->32 3x3 @4x4: %.2f | 8->8 3x3 @16 (tile off): %.2f" % (
    run(1024, 64, 64, 2, 3, 1, 1, 1), run(1024, 256, 256, 1, 3, 1, 1, 1), run(1024, 32, 32, 4, 3, 1, 1, 1), run(1024, 8, 8, 16, 3, 1, 1, 1)))
