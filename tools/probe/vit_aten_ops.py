"""which ATen operators launch kernels inside one eagerly issued conditional-ViT-VAE training step, and from which source line
(forward: the calling line; backward, run single-threaded so that the dispatch mode sees it: the autograd node's name)"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ot_vae_lightning_amd as A
from torch.utils._python_dispatch import TorchDispatchMode

GPU_OPS = ("copy_", "clone", "add", "fill_", "zero_", "zeros", "cat", "gather", "sum", "embedding", "index", "mul", "contiguous", "expand",
           "slice_backward", "select_backward", "new_zeros", "zeros_like", "empty_like")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        short = name.split(".")[0]
        if any(short == g or short.startswith(g) for g in ("copy_", "clone", "add", "fill_", "zero_", "cat", "gather", "sum", "embedding_dense_backward",
                                                            "index_select", "mul", "slice_backward", "select_backward", "zeros", "new_zeros")):
            big = [a for a in args if isinstance(a, torch.Tensor) and a.is_cuda]
            if big:
                frames = [f for f in traceback.extract_stack() if "/root/repo/" in f.filename or "GRAFT" in f.filename or "ot_vae_lightning_amd" in f.filename]
                frames = [f for f in frames if "vit_aten_ops" not in f.filename]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:]) or "(autograd engine)"
                self.by[(short, where)] += 1
        return func(*args, **(kwargs or {}))


torch.manual_seed(0)
B, D = 256, 128
cfg = dict(image_size=32, patch_size=8, dim=D, depth=3, heads=4, mlp_dim=4 * D, channels=3, dropout=0., emb_dropout=0., num_classes=10)
enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
prior = A.ConditionalGaussianPrior(dim=(1, D), num_classes=10, loss_coeff=0.1, annealing_steps=0)
model = A.VAE(encoder=enc, decoder=dec, prior=prior, conditional=True).cuda().train()
x = torch.randn(B, 3, 32, 32, device="cuda")
y = torch.randint(0, 10, (B,), device="cuda")
tr = A.HipTrainer(model, batch_shape=(B, 3, 32, 32), use_graph=False, batch_kwargs={"labels": y})
for _ in range(2):
    tr.step(x, labels=y)
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
spy = Spy()
with spy:
    tr.step(x, labels=y)
torch.cuda.synchronize()
for (name, where), n in sorted(spy.by.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:28s} {where}")
print("total:", sum(spy.by.values()))
