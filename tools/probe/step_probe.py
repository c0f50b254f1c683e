"""timing probe of the captured step (wrong values allowed): python tools/probe/step_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ot_vae_lightning_amd as A

def main():
    torch.manual_seed(0)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(1024, 1, 32, 32), step_guard=None)
    x = torch.randn(1024, 1, 32, 32, device="cuda")
    for _ in range(20):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        tr.step(x)
    torch.cuda.synchronize()
    print("ms/step %.4f" % ((time.perf_counter() - t0) / 200 * 1e3), flush=True)
    tr.close()

main()
