"""Would splitting K over more workgroups pay on the deep layers?  Proxy with the EXISTING kernels: the same FLOPs as (rows, K) and as
(4 x rows, K / 4) -- the second shape has 4 x the workgroups, each with a quarter of the K chunks (both uniform-tap: channels % 32 == 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import importlib.util
spec = importlib.util.spec_from_file_location("layer_time", os.path.join(os.path.dirname(os.path.abspath(__file__)), "layer_time.py"))
src = open(spec.origin).read().split("for cfg in [")[0]
ns = {}
exec(compile(src, spec.origin, "exec"), ns)
run = ns["run"]
for a, b in (((1024, 128, 64, 2, 3, 1, 1, 1), (4096, 32, 64, 2, 3, 1, 1, 1)),
             ((1024, 256, 256, 1, 3, 1, 1, 1), (4096, 64, 256, 1, 3, 1, 1, 1)),
             ((1024, 128, 128, 2, 3, 1, 1, 1), (4096, 32, 128, 2, 3, 1, 1, 1)),
             ((1024, 64, 64, 4, 3, 1, 1, 1), (2048, 32, 64, 4, 3, 1, 1, 1))):
    print("base   ", run(*a), flush=True)
    print("k-split", run(*b), "(same FLOPs, rows x%d, K /%d)" % (b[0] // a[0], a[1] // b[1]), flush=True)
