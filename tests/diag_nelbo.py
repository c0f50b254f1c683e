"""Diagnostic (GPU box): per-parameter gradient-norm error of the MNIST VAE (B=6 golden) in backward order."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # a test-side diagnostic: it uses the seeded fills under oracle/
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from conftest import group, load_golden
from detfill import fill_state_dict, mnist_like, normal
import ot_vae_lightning_amd as A

res = None if (len(sys.argv) > 1 and sys.argv[1] == "None") else "add"
g = group(load_golden("nelbo_mnist.npz"), str(res))
enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual=res)
dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual=res)
fill_state_dict(enc.state_dict()); fill_state_dict(dec.state_dict())
model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
x, eps = mnist_like(6, 42).cuda(), normal((6, 128, 1, 1), 43).cuda()
loss, logs, art = model.nelbo({"samples": x, "target": x, "kwargs": {"eps": eps}}, 0)
loss.backward()
names = [str(s) for s in g["param_names"]]
params = [p for net in (model.encoder, model.decoder) for _, p in net.named_parameters()]
l32, l64 = g["grad_l2"], g["grad_l2_f64"]
for i in reversed(range(len(names))):
    mine = params[i].grad.double().norm().item()
    e32 = abs(mine - l32[i].item()) / max(l32[i].item(), 1e-30)
    noise = abs(l32[i].item() - l64[i].item()) / max(l64[i].item(), 1e-30)
    flag = "  <<<" if e32 > max(3e-4, 2 * noise) else ""
    print(f"{names[i]:55s} |g|={l32[i].item():10.3e} err_vs_ref32={e32:9.2e} ref_noise={noise:9.2e}{flag}")
