"""Times one ConvLayer's three passes in isolation (hipGraph of 200 back-to-back launches through the C ABI)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, '.')
from ot_vae_lightning_amd import _lib as L
lib = L.load()
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
def graph_time(f, reps=200):
    for _ in range(5): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def run(n, cs, cn, hs, k, s, p, up):
    ho = (hs * up + 2 * p - k) // s + 1
    x = nhwc(torch.randn(n, cs, hs, hs, device="cuda")); w = torch.randn(k, k, cs, cn, device="cuda")
    sc = torch.rand(cs, device="cuda") + 0.5; sh = torch.randn(cs, device="cuda")
    y = nhwc(torch.empty(n, cn, ho, ho, device="cuda")); gy = nhwc(torch.randn(n, cn, ho, ho, device="cuda"))
    g = L.ConvGeom(n, hs, hs, cs, up, ho, ho, cn, k, k, s, p)
    wd = torch.empty(k * k * cn * cs, device="cuda")
    L.check(lib.otvae_weight_transpose(L.ptr(w), L.ptr(wd), k * k, cs, cn, L.stream()), "t")
    pf, ld = C.c_int(0), C.c_int(0)
    L.check(lib.otvae_conv_fwd_stats_ws(C.byref(g), C.byref(pf), C.byref(ld)), "ws")
    part = torch.empty((2, ld.value, pf.value), device="cuda", dtype=torch.float64)
    tf = graph_time(lambda: L.check(lib.otvae_conv_fwd(C.byref(g), L.ptr(x), L.ptr(sc), L.ptr(sh), 1, L.ptr(w), None, None, L.ptr(y), L.ptr(part), L.stream()), "f"))
    pd, cp = C.c_int(0), C.c_int(0)
    L.check(lib.otvae_conv_bwd_data_ws(C.byref(g), C.byref(pd), C.byref(cp)), "ws")
    gv = nhwc(torch.empty(n, cs, hs, hs, device="cuda")); mean = torch.zeros(cs, device="cuda"); inv = torch.ones(cs, device="cuda")
    bpart = torch.empty((2, cp.value, pd.value), device="cuda", dtype=torch.float64)
    td = graph_time(lambda: L.check(lib.otvae_conv_bwd_data(C.byref(g), L.ptr(gy), L.ptr(wd), L.ptr(x), L.ptr(sc), L.ptr(sh), 1, L.ptr(mean), L.ptr(inv), L.ptr(gv), L.ptr(bpart), L.stream()), "d"))
    pw = C.c_int(0)
    L.check(lib.otvae_conv_bwd_weight_ws(C.byref(g), 1, C.byref(pw)), "ws")
    wpart = torch.empty((pw.value, k * k * cs + 1, cn), device="cuda"); gw = torch.empty_like(w); gb = torch.empty(cn, device="cuda")
    tw = graph_time(lambda: L.check(lib.otvae_conv_bwd_weight(C.byref(g), L.ptr(x), L.ptr(sc), L.ptr(sh), 1, L.ptr(gy), 1, L.ptr(wpart), L.ptr(gw), L.ptr(gb), 1, L.stream()), "w"))
    return "%d->%d %dx%d k%d s%d up%d: fwd %.1f dgrad %.1f wgrad %.1f (P=%d)" % (cs, cn, hs, hs, k, s, up, tf, td, tw, pw.value)
for cfg in [(1024, 64, 64, 2, 3, 1, 1, 1), (1024, 256, 256, 1, 3, 1, 1, 1), (1024, 32, 32, 4, 3, 1, 1, 1), (1024, 64, 256, 2, 4, 2, 1, 1),
            (1024, 128, 64, 1, 3, 1, 1, 2), (1024, 16, 16, 8, 3, 1, 1, 1), (1024, 8, 8, 16, 3, 1, 1, 1), (1024, 256, 768, 1, 1, 1, 0, 1)]:
    print(run(*cfg))
