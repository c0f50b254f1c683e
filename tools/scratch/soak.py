"""Determinism soak: N captured steps, twice with the weight-gradient side stream and once without: identical bits expected."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
import ot_vae_lightning_amd as A
from ot_vae_lightning_amd import functional as HF
from ot_vae_lightning_amd.utils.synthetic import mnist_like

N = int(os.environ.get("STEPS", "600"))
def run(mode, workload):
    HF.WGRAD_SIDE_STREAM = mode
    model = B.build_model(A, seed=2, workload=workload).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(256, 1, 32, 32), data_parallel=False)
    xs = [mnist_like(256, seed=70 + i).cuda() for i in range(4)]
    last = None
    for i in range(N):
        last = tr.step(xs[i % 4])
    torch.cuda.synchronize()
    out = tr.pflat.clone(), last.clone()
    tr.close()
    return out
for wl in ("gaussian", "sinkhorn"):
    a, b, c = run(1, wl), run(1, wl), run(0, wl)
    print(wl, "loss", a[1].tolist(), "side==side", torch.equal(a[0], b[0]), "side==single", torch.equal(a[0], c[0]),
          "finite", bool(torch.isfinite(a[0]).all()))
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], c[0]) and torch.isfinite(a[0]).all()
print("soak ok")
