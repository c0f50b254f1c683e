"""Conditional ViT VAE of the reference's tests/test_conditional_vit_vae.py (CIFAR-sized 32x32x3, patch 8, dim 128, depth 3,
4 heads, 10 classes, dropout DROPOUT=0 by default) on one GPU through HipTrainer (flat parameters, HIP Adam, hipGraph replay), batch 256.
A sanity number for the (f-4) path, not the headline metric."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ot_vae_lightning_amd as A

torch.manual_seed(0)
B, D = int(os.environ.get("B", "256")), 128
DROP = float(os.environ.get("DROPOUT", "0"))
cfg = dict(image_size=32, patch_size=8, dim=D, depth=3, heads=4, mlp_dim=4 * D, channels=3, dropout=DROP, emb_dropout=0., num_classes=10)
enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
prior = A.ConditionalGaussianPrior(dim=(1, D), num_classes=10, loss_coeff=0.1, annealing_steps=0)
model = A.VAE(encoder=enc, decoder=dec, prior=prior, conditional=True).cuda().train()
x = torch.randn(B, 3, 32, 32, device="cuda")
y = torch.randint(0, 10, (B,), device="cuda")
tr = A.HipTrainer(model, batch_shape=(B, 3, 32, 32), use_graph=os.environ.get("GRAPH", "1") == "1", batch_kwargs={"labels": y})
for _ in range(5):
    out = tr.step(x, labels=y)
torch.cuda.synchronize()
first = out.tolist()
t0 = time.perf_counter()
n = 50
for _ in range(n):
    out = tr.step(x, labels=y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("conditional ViT VAE (dim 128, depth 3, 19/18 tokens, dropout %g) batch %d: %.3f ms/step, %.0f img/s, loss %s -> %s"
      % (DROP, B, dt * 1e3, B / dt, [round(v, 4) for v in first], [round(v, 4) for v in out.tolist()]))
