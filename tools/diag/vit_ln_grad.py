"""Where does the 4.5e-3 error of the decoder's positional-embedding LayerNorm parameter gradients come from (tests/test_gpu_configs.py::
test_config5_conditional_vit_vae_at_the_yaml_shape_vs_oracle, round 4)?  Captures g = dL/d(LayerNorm output) and xhat on the HIP path and
in the oracle (fp32 and fp64) and recombines dgamma = sum g * xhat from mixed factors in fp64."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import otvae_oracle as O  # noqa: E402
import ot_vae_lightning_amd as A  # noqa: E402
from detfill import fill_vit_state_dict, normal  # noqa: E402
from test_oracle_vs_golden import VIT_ROLES, vit_param_shapes  # noqa: E402

B, ncls = 64, 10
base = dict(image_size=64, patch_size=8, dim=256, heads=8, mlp_dim=1024, channels=3, num_classes=ncls)
cfgs = {"enc": dict(depth=3, **base), "dec": dict(depth=2, **base)}
x = normal((B, 3, 64, 64), 151)
eps = normal((B, 1, 256), 152)
labels = torch.arange(B) % ncls
g_ = torch.Generator().manual_seed(153)
mu_w, ls_w = torch.randn(ncls, 256, generator=g_) * 0.3, torch.randn(ncls, 256, generator=g_) * 0.1
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
vit = lambda role: dict(image_size=64, patch_size=8, dim=256, depth=cfgs[role]["depth"], heads=8, channels=3, labels=labels, **VIT_ROLES[role])  # noqa: E731


def oracle_run(dtype):
    p = {}
    for role in ("enc", "dec"):
        sd = {k: torch.zeros(sh) for k, sh in vit_param_shapes(cfgs[role], role).items()}
        fill_vit_state_dict(sd)
        p[role] = {k: v.to(dtype).requires_grad_(True) for k, v in sd.items()}
    mw, lw = mu_w.to(dtype).requires_grad_(True), ls_w.to(dtype).requires_grad_(True)
    calls = []
    real = F.layer_norm

    def spy(inp, shape, w, b, e):
        out = real(inp, shape, w, b, e)
        out.retain_grad()
        calls.append((inp, out, w))
        return out
    O.F.layer_norm = spy
    try:
        h = O.vit_forward(x.to(dtype), p["enc"], **vit("enc"))
        n_enc = len(calls)
        z, pl = O.cond_gaussian_prior_encode(h, eps.to(dtype), mw, lw, labels, 0.1, 0, 0)
        preds = O.vit_forward(z, p["dec"], **vit("dec"))
        (F.mse_loss(preds, x.to(dtype)) + pl.mean() / float(x[0].numel())).backward()
    finally:
        O.F.layer_norm = real
    inp, out, w = calls[n_enc]          # the decoder's first LayerNorm = positional_embed.LayerNorm
    assert w is p["dec"]["positional_embed.LayerNorm.weight"]
    xs = inp.detach().double()
    xh = (xs - xs.mean(-1, keepdim=True)) / torch.sqrt(xs.var(-1, unbiased=False, keepdim=True) + 1e-5)
    return dict(g=out.grad.detach().double(), xhat=xh, dgamma=w.grad.detach().double(),
                dbeta=p["dec"]["positional_embed.LayerNorm.bias"].grad.detach().double(), p=p, mw=mw, lw=lw)


r64, r32 = oracle_run(torch.float64), oracle_run(torch.float32)
from ot_vae_lightning_amd import functional as HF  # noqa: E402
import math  # noqa: E402

REAL = dict(mha=HF.mha_attention_tokens, ln=HF.layer_norm_tokens, lin=HF.linear_tokens)


def torch_mha(qkv, n_heads, dropout_p=0.0, dropout_key=None, stream_id=0, return_used=False, causal=False):
    n, t, w3 = qkv.shape
    c = w3 // (3 * n_heads)
    q, k, v = qkv.reshape(n, t, 3, n_heads, c).permute(2, 0, 3, 1, 4)   # [N, H, T, C]
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(c), dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n, t, n_heads * c)


def torch_ln(x, gamma, beta, eps=1e-5, residual=None, *a, **k):
    return F.layer_norm(x if residual is None else x + residual, (x.shape[-1],), gamma, beta, eps)


def torch_lin(x, weight, bias, relu_input=False):
    return F.linear(torch.relu(x) if relu_input else x, weight, bias)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def hip_run(swap):
    HF.mha_attention_tokens = torch_mha if "mha" in swap else REAL["mha"]
    HF.layer_norm_tokens = torch_ln if "ln" in swap else REAL["ln"]
    HF.linear_tokens = torch_lin if "lin" in swap else REAL["lin"]
    nets = {}
    for role in ("enc", "dec"):
        net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., **cfgs[role], **VIT_ROLES[role])
        net.load_state_dict({k: v.detach() for k, v in r32["p"][role].items()})
        nets[role] = net
    prior = A.ConditionalGaussianPrior(dim=(1, 256), num_classes=ncls, loss_coeff=0.1)
    with torch.no_grad():
        prior._mu.weight.copy_(mu_w)
        prior._log_std.weight.copy_(ls_w)
    model = A.VAE(encoder=nets["enc"], decoder=nets["dec"], prior=prior, conditional=True).cuda().train()
    store = {}
    ln = model.decoder.positional_embed.LayerNorm
    real_fwd = ln.forward

    def spy_fwd(inp, residual=None, *a, **k):
        out = real_fwd(inp, residual, *a, **k)
        store["xs"] = (inp + residual).detach() if residual is not None else inp.detach()
        if out.requires_grad:
            out.register_hook(lambda g: store.__setitem__("g", g.detach().clone()))
        return out

    ln.forward = spy_fwd
    tr = A.HipTrainer(model, batch_shape=(B, 3, 64, 64), use_graph=False, batch_kwargs={"labels": labels.cuda()})
    tr.step(x.cuda(), eps.cuda(), labels=labels.cuda())
    torch.cuda.synchronize()
    g_hip = store["g"].double().cpu().reshape(r64["g"].shape)
    gam = ln.weight.grad if ln.weight.grad is not None else ln.weight._otvae_grad_view()
    bet = ln.bias.grad if ln.bias.grad is not None else ln.bias._otvae_grad_view()
    res = dict(g=rel(g_hip, r64["g"]), dgamma=rel(gam.double().cpu(), r64["dgamma"]), dbeta=rel(bet.double().cpu(), r64["dbeta"]),
               dgamma_from_g=rel((g_hip * r64["xhat"]).sum((0, 1)), r64["dgamma"]))
    tr.close()
    return res


print("reference fp32 (oracle):  g %.3e  dgamma %.3e  dbeta %.3e" % (rel(r32["g"], r64["g"]), rel(r32["dgamma"], r64["dgamma"]), rel(r32["dbeta"], r64["dbeta"])))
for swap in ((), ("mha",), ("ln",), ("lin",), ("mha", "ln"), ("mha", "ln", "lin")):
    r = hip_run(swap)
    print("HIP path with torch ops for %-18s: g %.3e  dgamma %.3e  dbeta %.3e  (dgamma from g in fp64 %.3e)" %
          ("+".join(swap) or "nothing", r["g"], r["dgamma"], r["dbeta"], r["dgamma_from_g"]), flush=True)

# ---- the decoder's first attention call in isolation: which part of the HIP kernel's error is the systematic one?
HF.mha_attention_tokens, HF.layer_norm_tokens, HF.linear_tokens = REAL["mha"], REAL["ln"], REAL["lin"]
cap = {}


def spy_mha(qkv, n_heads, *a, **k):
    out = REAL["mha"](qkv, n_heads, *a, **k)
    if "qkv" not in cap and qkv.shape[1] == r64["g"].shape[1]:   # the decoder's token count
        cap["qkv"] = qkv.detach().clone()
        out.register_hook(lambda g: cap.__setitem__("gout", g.detach().clone()))
    return out


HF.mha_attention_tokens = spy_mha
nets = {}
for role in ("enc", "dec"):
    net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., **cfgs[role], **VIT_ROLES[role])
    net.load_state_dict({k: v.detach() for k, v in r32["p"][role].items()})
    nets[role] = net
prior = A.ConditionalGaussianPrior(dim=(1, 256), num_classes=ncls, loss_coeff=0.1)
with torch.no_grad():
    prior._mu.weight.copy_(mu_w)
    prior._log_std.weight.copy_(ls_w)
model = A.VAE(encoder=nets["enc"], decoder=nets["dec"], prior=prior, conditional=True).cuda().train()
tr = A.HipTrainer(model, batch_shape=(B, 3, 64, 64), use_graph=False, batch_kwargs={"labels": labels.cuda()})
tr.step(x.cuda(), eps.cuda(), labels=labels.cuda())
torch.cuda.synchronize()
HF.mha_attention_tokens = REAL["mha"]
qkv0, gout0 = cap["qkv"], cap["gout"]
w_in = model.decoder.transformer.layers[0].self_attn.in_proj_weight.detach().double().cpu()   # [3D, D]
print("captured qkv", tuple(qkv0.shape), "gout", tuple(gout0.shape), "|qkv| max %.2f" % float(qkv0.abs().max()))


def run(fn, dtype, dev):
    q = qkv0.to(device=dev, dtype=dtype).clone().requires_grad_(True)
    o = fn(q, 8)
    o.backward(gout0.to(device=dev, dtype=dtype))
    return o.detach().double().cpu(), q.grad.detach().double().cpu()


o64, d64 = run(torch_mha, torch.float64, "cpu")
o32, d32 = run(torch_mha, torch.float32, "cpu")
oh, dh = run(REAL["mha"], torch.float32, "cuda")
xh = r64["xhat"]
D = 256
for name, o_, d_ in (("torch fp32", o32, d32), ("HIP kernel", oh, dh)):
    e = d_ - d64
    parts = {"dq": e[..., :D], "dk": e[..., D:2 * D], "dv": e[..., 2 * D:]}
    msg = [f"{name}: out rel err {rel(o_, o64):.2e}; dqkv rel err {rel(d_, d64):.2e};"]
    for pn, pe in parts.items():
        full = torch.zeros_like(e)
        sl = {"dq": slice(0, D), "dk": slice(D, 2 * D), "dv": slice(2 * D, 3 * D)}[pn]
        full[..., sl] = pe
        ey = full @ w_in                                        # the in-projection's data gradient of the error
        contrib = (ey * xh).sum((0, 1))                          # what it adds to dgamma (before the LayerNorm's own backward)
        msg.append(f"{pn}: |e| {float(pe.norm() / d64[..., sl].norm()):.2e} -> dgamma shift {float(contrib.norm() / r64['dgamma'].norm()):.2e};")
    print(" ".join(msg))
tr.close()
