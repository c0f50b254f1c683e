import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ot_vae_lightning_amd as A
from detfill import normal
from torch.profiler import ProfilerActivity, profile
z = normal((256, 128), 51).cuda().requires_grad_(True)
ps = normal((256, 128), 52).cuda()
prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, loss_coeff=0.5).cuda()
g = torch.full((256,), 1.0 / 256, device="cuda")
def run():
    z.grad = None
    zz, loss, _ = prior(z, step=0, prior_samples=ps)
    torch.autograd.backward(loss, grad_tensors=[g], inputs=[z])
run(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    run(); torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=80, max_src_column_width=120))
