"""Where the time of the graphed Lightning route goes (engine/graphed.py): the same loop with pieces removed / swapped.
    python tools/lightning_route_probe.py            # all variants, ms per step at batch 1024"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ot_vae_lightning_amd as A  # noqa: E402
from ot_vae_lightning_amd.utils.synthetic import mnist_like  # noqa: E402


def build():
    torch.manual_seed(1)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train().enable_graphed_step()


def run(tag, make_opt, do_backward=True, steps=40, warmup=10, batch=1024):
    model = build()
    opt = make_opt(model) if make_opt else None
    pool = [mnist_like(batch, seed=5 + i).cuda() for i in range(4)]
    labels = torch.zeros(batch, dtype=torch.long, device="cuda")

    def step(i):
        if opt is not None:
            opt.zero_grad()
        else:
            for p in model.parameters():
                p.grad = None
        out = model.training_step((pool[i % 4], labels), i)
        if do_backward:
            out["loss"].backward()
        if opt is not None:
            opt.step()

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    host = (time.perf_counter() - t0) / steps * 1e3     # host time to ENQUEUE a step
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag:58s} {total:7.3f} ms/step   (host enqueue {host:6.3f})", flush=True)


if __name__ == "__main__":
    params = lambda m: list(m.optim_parameters())  # noqa: E731
    run("forward replay only", None, do_backward=False)
    run("forward + backward replay, no optimizer", None)
    run("+ torch.optim.Adam (default: foreach)", lambda m: torch.optim.Adam(params(m), lr=1e-3))
    run("+ torch.optim.Adam(fused=True)", lambda m: torch.optim.Adam(params(m), lr=1e-3, fused=True))
    run("+ torch.optim.Adam(foreach=False)", lambda m: torch.optim.Adam(params(m), lr=1e-3, foreach=False))
    run("+ torch.optim.SGD(momentum=0.9)", lambda m: torch.optim.SGD(params(m), lr=1e-3, momentum=0.9))
