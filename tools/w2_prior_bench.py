"""Training step of the MNIST-32 CNN VAE with the Gaussian-W2 prior (north_star: "Gaussian W2 with empirical covariance"), batch 1024,
captured step: ms per step and the share of the eigendecomposition (latent 128)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ot_vae_lightning_amd as A  # noqa: E402
from ot_vae_lightning_amd.utils.synthetic import mnist_like  # noqa: E402


def main(steps=100, warmup=20, batch=1024):
    torch.manual_seed(0)
    enc = A.CNN(1, 128, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianW2Prior(loss_coeff=0.1)).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(batch, 1, 32, 32), data_parallel=False)
    xs = [mnist_like(batch, seed=5 + i).cuda() for i in range(4)]
    for i in range(warmup):
        tr.step(xs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = tr.step(xs[i % 4])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"MNIST-32 CNN VAE + GaussianW2Prior, batch {batch}: {ms:.3f} ms/step, {batch / ms * 1e3:.0f} img/s, loss {out.tolist()}")
    tr.close()


if __name__ == "__main__":
    main()
