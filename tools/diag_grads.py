import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ot_vae_lightning_amd as A
from ot_vae_lightning_amd.utils.synthetic import mnist_like
torch.manual_seed(0)
enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lat = A.GaussianTransport(128, source_cfg=dict(dtype=torch.double, reduce_on_update=False),
                          target_cfg=dict(dtype=torch.double, reduce_on_update=False), transport_cfg=dict(make_pd=True)).cuda()
tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=(sys.argv[1] == "graph"), latent_stats=lat if len(sys.argv) > 3 else None)
for _ in range(3):
    tr.step(mnist_like(B).cuda())
torch.cuda.synchronize()
stolen = cloned = 0
for (name, p), off in zip([(n, p) for n, p in list(model.encoder.named_parameters()) + list(model.decoder.named_parameters())], tr.offsets):
    slot = tr.gflat.data_ptr() + 4 * off
    if p.grad is None:
        print("no grad", name); continue
    if p.grad.data_ptr() == slot:
        stolen += 1
    else:
        cloned += 1
        if cloned <= 6:
            print("cloned:", name, tuple(p.shape), p.stride(), "grad stride", p.grad.stride(), "refcnt", sys.getrefcount(p.grad))
print("stolen", stolen, "cloned", cloned)
