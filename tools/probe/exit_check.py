"""Exit-path probe: one captured engine, three steps, normal interpreter exit (run under rocgdb to see where a teardown fault comes from)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import torch
import ot_vae_lightning_amd as A
torch.manual_seed(0)
B = 64
enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=True)
x = torch.rand(B, 1, 32, 32, device="cuda")
for _ in range(3):
    out = tr.step(x)
torch.cuda.synchronize()
print("steps ok", out.tolist(), flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "close":
    tr.close()
print("exiting", flush=True)
