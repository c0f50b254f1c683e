"""BASELINE configs[3] shape on one GPU: CIFAR-sized CNN VAE (3 channels, capacity 16, latent 256) + Sinkhorn prior, batch 256."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ot_vae_lightning_amd as A
torch.manual_seed(0)
B = 256
enc = A.CNN(3, 256, 32, 1, capacity=16, down_sample=True, residual="add")
dec = A.CNN(256, 3, 1, 32, capacity=16, up_sample=True, residual="add")
model = A.VAE(encoder=enc, decoder=dec, prior=A.SinkhornPrior(reg=0.05, max_iter=50)).cuda().train()
tr = A.HipTrainer(model, batch_shape=(B, 3, 32, 32), use_graph=True)
x = torch.randn(B, 3, 32, 32, device="cuda")
for _ in range(5): tr.step(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): out = tr.step(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("CIFAR cfg (3ch, capacity 16, latent 256, Sinkhorn prior) batch %d: %.3f ms/step, %.0f img/s, loss %s" % (B, dt * 1e3, B / dt, out.tolist()))
