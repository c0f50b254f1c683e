"""Stand-alone timing of the two large attention shapes of the MNIST network through the C ABI (20 launches per hipGraph)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from ot_vae_lightning_amd import _lib as L, functional as HF
lib = L.load()
def run(n, t, heads, c, tag):
    qkv = torch.randn(n, t, 3*heads*c, device="cuda")
    out = torch.empty(n, t, heads*c, device="cuda"); lse = torch.empty(n, heads, t, device="cuda"); aux = torch.empty(n, heads, t, c*c, device="cuda")
    g = torch.randn_like(out); gq = torch.empty_like(qkv)
    def f(): L.check(lib.otvae_attn_fwd(L.ptr(qkv), n, t, heads, c, L.ptr(out), L.ptr(lse), L.ptr(aux), L.stream()), "f")
    def b(): L.check(lib.otvae_attn_bwd(L.ptr(qkv), L.ptr(out), L.ptr(lse), L.ptr(g), L.ptr(aux), n, t, heads, c, L.ptr(gq), L.stream()), "b")
    for name, fn in (("fwd", f), ("bwd", b)):
        for _ in range(3): fn()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20): fn()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gr.replay()
        e1.record(); torch.cuda.synchronize()
        print(tag, name, (n, t, heads, c), "%.1f us" % (e0.elapsed_time(e1) / 100 * 1e3), flush=True)
tag = os.environ.get("OTVAE_LIB", "default")
run(1024, 1024, 1, 1, tag + " mnist dec L4")
run(1024, 256, 4, 2, tag + " mnist enc L0 / dec L3")
run(1024, 64, 4, 4, tag + " mnist enc L1 / dec L2")
run(1024, 16, 8, 4, tag + " mnist enc L2 / dec L1")
run(1024, 4, 8, 8, tag + " mnist enc L3 / dec L0")
run(256, 256, 4, 4, tag + " cifar enc L0")
run(256, 1024, 3, 1, tag + " cifar dec L4")
