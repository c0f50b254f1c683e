"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

A CPU restatement (plain PyTorch CPU ops, fp32/fp64) of the arithmetic on the hot path of
theoad/ot-vae-lightning, written from the reference's behaviour, each function citing the
reference file:line it follows (paths relative to /root/reference/).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
file.  The product (``ot_vae_lightning_amd``) never imports it and has no CPU fallback: it fails
loudly when the HIP library is missing.

Pinning: every function here is checked against golden vectors recorded from the *real* reference
(imported in the build container by ``oracle/ref_import.py``; generator ``oracle/gen_golden.py``;
vectors in ``tests/golden/``) by ``tests/test_oracle_vs_golden.py``.

Everything is functional: network weights are passed as a flat ``dict`` keyed exactly like the
reference's ``state_dict`` (e.g. ``0.block.1._normalization.running_mean``), so a product
``state_dict`` can be fed straight in.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

STABILITY_CONST = 1e-8  # ot_vae_lightning/ot/matrix_utils.py:33
BN_EPS = 1e-5           # nn.BatchNorm2d default, networks/cnn.py:122
BN_MOMENTUM = 0.1


# ------------------------------------------------------------------------------------------------
# architecture inference  (networks/cnn.py:605-672, 416-455)
# ------------------------------------------------------------------------------------------------
def divisors(n: int) -> List[int]:
    return [d for d in range(1, n + 1) if n % d == 0]


def div_sqrt(n: int) -> int:
    """Smallest divisor of n that is >= sqrt(n) (networks/cnn.py:660-672: searchsorted on divisors)."""
    r = math.sqrt(n)
    for d in divisors(n):
        if d >= r:
            return d
    return n


def block_scaling(max_res: int, min_res: int, max_scaling: int) -> List[int]:
    """networks/cnn.py:605-621."""
    log_ratio = int(math.log2(max_res // min_res))
    log_scale = int(math.log2(max_scaling))
    out: List[int] = []
    while log_ratio > 0:
        out.extend([2 ** log_scale] * (log_ratio // log_scale))
        log_ratio %= log_scale
        log_scale -= 1
    return out


def channel_list(in_f: int, out_f: int, in_res: int, out_res: int, scaling: int, capacity: int):
    """networks/cnn.py:627-654."""
    sfs = block_scaling(in_res, out_res, scaling)
    feats = [max(min(2 ** i * capacity, out_f), in_f) for i in range(len(sfs))]
    res = [in_res]
    for sf in sfs:
        res.append(res[-1] // sf)
    feats[-1] = out_f
    return [in_f] + feats, res


def cnn_arch(in_features: int, out_features: int, in_resolution: int, out_resolution: int,
             capacity: int = 8, max_attn_res: int = 16, down_sample: bool = False, up_sample: bool = False,
             residual: Optional[str] = None, n_layers: int = 2) -> List[dict]:
    """Per-ConvBlock description of ``CNN(...)`` (networks/cnn.py:416-455, 306-329)."""
    assert bool(down_sample) != bool(up_sample)
    if down_sample:
        feats, res = channel_list(in_features, out_features, in_resolution, out_resolution, 2, capacity)
        attn_res = res[1:]
        in_res = res[:-1]
    else:
        feats, res = channel_list(out_features, in_features, out_resolution, in_resolution, 2, capacity)
        feats, res = feats[::-1], res[::-1]
        attn_res = res[:-1]
        in_res = res[:-1]
    blocks = []
    for ic, oc, ar, ir in zip(feats[:-1], feats[1:], attn_res, in_res):
        heads = div_sqrt(oc) if ar <= max_attn_res else 0
        embed = oc // 2 if residual == "cat" else oc
        if residual == "cat":
            heads = div_sqrt(oc) if ar <= max_attn_res else 0  # heads computed from oc (cnn.py:445,449)
        blocks.append(dict(cin=ic, cout=oc, embed=embed, heads=heads, down=bool(down_sample), up=bool(up_sample),
                           residual=residual, n_layers=n_layers, in_res=ir))
    return blocks


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
_ACTIVATIONS = {  # cnn.py:128-147 (the reference tests the names in this order: "leaky" before "relu")
    "leaky": lambda t: F.leaky_relu(t, 0.2), "relu": F.relu, "selu": F.selu, "gelu": F.gelu, "silu": F.silu,
}


def conv_layer(x: Tensor, p: Dict[str, Tensor], prefix: str, *, down: bool, up: bool, relu: bool,
               norm: bool, ksize: int = 3, training: bool = True, act: Optional[str] = None,
               equalized_lr: Optional[float] = None, other_norm: Optional[str] = None, embed: Optional[Tensor] = None,
               groups: int = 1, dilation: int = 1, padding: Optional[int] = None, up_module=None, down_module=None,
               stride: Optional[int] = None) -> Tensor:
    """``ConvLayer.forward`` (networks/cnn.py:183-192): BN -> act -> nearest x2 up -> conv (stride-2 4x4 when
    down-sampling, cnn.py:98-101; ``down`` may be the integer factor s: a max(2 s, ksize) kernel with stride s; ``stride``: a plain
    nn.Conv2d stride without down-sampling).  ``p[prefix+'_normalization.running_*']`` are updated in place like
    nn.BatchNorm2d does in training mode.  ``act``: one of leaky / relu / selu / gelu / silu (overrides ``relu``);
    ``equalized_lr``: weight * (1 / sqrt(fan_in)) * lr_mult, bias * lr_mult (cnn.py:114-118,186-188); ``groups`` / ``dilation`` /
    ``padding``: nn.Conv2d's (cnn.py:66-67,103-104), the weight is [out, in / groups, k, k]."""
    out = x
    if norm:
        out = F.batch_norm(out, p[prefix + "_normalization.running_mean"], p[prefix + "_normalization.running_var"],
                           p[prefix + "_normalization.weight"], p[prefix + "_normalization.bias"],
                           training=training, momentum=BN_MOMENTUM, eps=BN_EPS)
    if other_norm == "group":      # nn.GroupNorm(div_sqrt(C // groups), C) (cnn.py:123)
        c = out.shape[1]
        cg = c // groups
        g = next(d for d in range(1, cg + 1) if cg % d == 0 and d >= math.sqrt(cg))
        out = F.group_norm(out, g, p[prefix + "_normalization.weight"], p[prefix + "_normalization.bias"], eps=BN_EPS)
    elif other_norm == "instance":  # nn.InstanceNorm2d(C): no affine, no running statistics (cnn.py:124)
        out = F.instance_norm(out, eps=BN_EPS)
    fn = _ACTIVATIONS[act] if act is not None else (F.relu if relu else (lambda t: t))
    if embed is not None:  # FiLM (cnn.py:160-181): scale / bias = Linear(act(embed)) with the equalized_lr multipliers
        lr = equalized_lr or 1.0
        ls = (1.0 / math.sqrt(x.shape[1])) if equalized_lr else 1.0
        e = fn(embed)
        sc = F.linear(e, p[prefix + "_embed_proj_scale.weight"] * ls * lr, p[prefix + "_embed_proj_scale.bias"] * lr)
        bi = F.linear(e, p[prefix + "_embed_proj_bias.weight"] * ls * lr, p[prefix + "_embed_proj_bias.bias"] * lr)
        out = out * sc[..., None, None] + bi[..., None, None]
    out = fn(out)
    if up_module is not None:   # a user-supplied nn.Module instead of the built-in nearest up-sampling (cnn.py:106,187)
        out = up_module(out)
    if up:
        out = F.interpolate(out, scale_factor=2.0, mode="nearest")
    if down:
        f = 2 if down is True else int(down)
        k = max(2 * f, ksize)
        stride, pad = f, (k - 1) // 2
    else:
        k, stride, pad = ksize, (stride or 1), (1 if ksize == 3 else 0)
    w = p[prefix + "weight"]
    assert w.shape[-1] == k, (prefix, w.shape, k)
    bias = p.get(prefix + "bias")
    if equalized_lr:
        w = w * (1.0 / math.sqrt(w.shape[1] * w.shape[2] * w.shape[3])) * equalized_lr
        bias = bias * equalized_lr if bias is not None else None
    if padding is not None and not down:
        pad = padding
    out = F.conv2d(out, w, bias, stride=stride, padding=pad, dilation=dilation, groups=groups)
    return down_module(out) if down_module is not None else out   # a user-supplied down-sampling module (cnn.py:97,190)


def qkv_attention(qkv: Tensor, n_heads: int) -> Tensor:
    """``QKVAttention.forward`` (networks/nets_utils.py:63-82).  qkv [N, 3*H*C, T] -> [N, H*C, T].
    Both q and k are scaled by C**-0.5; softmax over keys in fp32; no mask / residual."""
    n, width, t = qkv.shape
    ch = width // (3 * n_heads)
    q, k, v = qkv.chunk(3, dim=1)
    scale = 1.0 / math.sqrt(ch)
    q = (q * scale).view(n, n_heads, ch, t)
    k = (k * scale).view(n, n_heads, ch, t)
    w = torch.einsum("nhct,nhcs->nhts", q, k)
    w = torch.softmax(w.float(), dim=-1).to(w.dtype)
    a = torch.einsum("nhts,nhcs->nhct", w, v.reshape(n, n_heads, ch, t))
    return a.reshape(n, -1, t)


def attention_block(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, training: bool = True) -> Tensor:
    """``AttentionBlock.forward`` (networks/cnn.py:235-240): proj_out(attn(qkv(BN(x)))), NOT residual."""
    spatial = x.shape[2:]
    qkv = conv_layer(x, p, prefix + "qkv.", down=False, up=False, relu=False, norm=True, ksize=1,
                     training=training).flatten(2)
    h = qkv_attention(qkv, heads).unflatten(2, spatial)
    return conv_layer(h, p, prefix + "proj_out.", down=False, up=False, relu=False, norm=False, ksize=1,
                      training=training)


def conv_block(x: Tensor, p: Dict[str, Tensor], prefix: str, blk: dict, training: bool = True) -> Tensor:
    """``ConvBlock.forward`` (networks/cnn.py:331-335)."""
    out = conv_layer(x, p, prefix + "block.0.", down=blk["down"], up=blk["up"], relu=True, norm=True,
                     training=training)
    for j in range(1, blk["n_layers"]):
        out = conv_layer(out, p, prefix + f"block.{j}.", down=False, up=False, relu=True, norm=True,
                         training=training)
    if blk["heads"] > 0:
        out = attention_block(out, p, prefix + f"block.{blk['n_layers']}.", blk["heads"], training=training)
    if blk["residual"] in ("add", "cat"):
        sk = conv_layer(x, p, prefix + "skip.", down=blk["down"], up=blk["up"], relu=False, norm=True, ksize=1,
                        training=training)
        out = out + sk if blk["residual"] == "add" else torch.cat([out, sk], dim=1)
    return out


def cnn_forward(x: Tensor, p: Dict[str, Tensor], arch: Sequence[dict], prefix: str = "",
                training: bool = True) -> Tensor:
    """``CNN.forward`` (networks/cnn.py:457-458) = the ConvBlocks in sequence."""
    for i, blk in enumerate(arch):
        x = conv_block(x, p, f"{prefix}{i}.", blk, training=training)
    return x


# ------------------------------------------------------------------------------------------------
# prior / loss  (prior/base.py:74-78, prior/gaussian.py:63-96, model/vae.py:158-189)
# ------------------------------------------------------------------------------------------------
def prior_annealing(step: int, annealing_steps: int) -> float:
    """prior/base.py:75."""
    if annealing_steps > step:
        return 0.5 * math.cos(math.pi * (step / annealing_steps + 1)) + 0.5
    return 1.0


def gaussian_prior_encode(x: Tensor, eps: Tensor, loss_coeff: float = 1.0, step: int = 0,
                          annealing_steps: int = 0) -> Tuple[Tensor, Tensor]:
    """``GaussianPrior.encode`` + ``Prior.forward`` with the N(0,1) draw made explicit (``eps``).
    mu, log_var = chunk(x, 2, dim=1); std = exp(log_var/2); z = mu + eps*std;
    KL(q||N(0,I)) = sum 0.5*(mu^2 + log(1) - log(std^2) + std^2 - 1) over all non-batch dims."""
    mu, log_var = torch.chunk(x, 2, dim=1)
    std = (log_var / 2).exp()
    z = mu + eps * std
    var = std ** 2
    dims = list(range(1, mu.dim()))
    # same term order as closed_form_reverse_kl with p = N(0, 1): (mu-0)^2/1 + log(1) - log(var) + var/1 - 1
    kl = torch.sum(0.5 * (mu ** 2 + 0.0 - var.log() + var - 1), dim=dims)
    return z, kl * (loss_coeff * prior_annealing(step, annealing_steps))


def gaussian_prior_encode_options(x: Tensor, eps: Tensor, loss_coeff: float = 1.0, empirical_kl: bool = False,
                                  fixed_var: bool = False, time: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """``GaussianPrior.encode`` with ``empirical_kl`` / ``fixed_var`` (prior/gaussian.py:63-96) and ``Prior.empirical_reverse_kl``
    (prior/base.py:65-68).  Note that the reference hands the STANDARD DEVIATION to ``Normal`` under the name ``var``."""
    if fixed_var:
        mu, sd = x, torch.ones_like(x)
        if time is not None:
            sd = sd * time.reshape(-1, *([1] * (x.dim() - 1))) + 1e-8
    else:
        mu, log_var = torch.chunk(x, 2, dim=1)
        sd = (log_var / 2).exp()
    z = mu + eps * sd
    dims = list(range(1, mu.dim()))
    if empirical_kl:
        q, p = torch.distributions.Normal(mu, sd), torch.distributions.Normal(torch.zeros_like(mu), torch.ones_like(mu))
        loss = (q.log_prob(z) - p.log_prob(z)).sum(dims)
    else:
        var = sd ** 2
        loss = torch.sum(0.5 * (mu ** 2 + 0.0 - var.log() + var - 1), dim=dims)
    return z, loss * loss_coeff


def prior_encode_general(x: Tensor, eps: Tensor, loss_coeff: float = 1.0, empirical_kl: bool = False, fixed_var: bool = False,
                         reparam_dim: int = 1, prior_mean: Optional[Tensor] = None, prior_log_std: Optional[Tensor] = None):
    """``GaussianPrior.encode`` / ``ConditionalGaussianPrior.encode`` in full generality (prior/gaussian.py:63-96,
    prior/conditional_gaussian.py:81-93, prior/base.py:65-68): q from ``reparametrization`` (fixed_var: N(x, 1); else chunk(x, 2,
    reparam_dim) -> N(mu, exp(log_var / 2))), p = N(0, 1) or N(prior_mean, exp(prior_log_std)) (the gathered class rows, shaped like
    z), z = mu + eps sd, loss = closed-form KL(q || p) or log q(z) - log p(z), summed over the non-batch dimensions."""
    if fixed_var:
        mu, sd = x, torch.ones_like(x)
    else:
        mu, log_var = torch.chunk(x, 2, reparam_dim)
        sd = (log_var / 2).exp()
    z = mu + eps * sd
    pm = torch.zeros_like(mu) if prior_mean is None else prior_mean.reshape(mu.shape)
    ps = torch.ones_like(mu) if prior_log_std is None else prior_log_std.reshape(mu.shape).exp()
    dims = list(range(1, mu.dim()))
    if empirical_kl:
        loss = (torch.distributions.Normal(mu, sd).log_prob(z) - torch.distributions.Normal(pm, ps).log_prob(z)).sum(dims)
    else:
        loss = torch.sum(0.5 * ((mu - pm) ** 2 / ps ** 2 + (ps ** 2).log() - (sd ** 2).log() + sd ** 2 / ps ** 2 - 1), dim=dims)
    return z, loss * loss_coeff


def vae_nelbo(x: Tensor, eps: Tensor, enc: Dict[str, Tensor], dec: Dict[str, Tensor], enc_arch, dec_arch,
              loss_coeff: float = 1.0, step: int = 0, annealing_steps: int = 0, training: bool = True, expansion: int = 1):
    """``VAE.nelbo`` (model/vae.py:165-189): loss = mse(mean over replicas of decode(z), x) + mean(prior)/(C*H*W).  ``expansion`` = n
    > 1: the encoder output is replicated n times, replica-major (utils.replicate_batch, utils/__init__.py:141-164: expand on a new
    leading axis, then fold it into the batch), ``eps`` holds n*B draws, the prior term is the mean over all n*B entries, the
    reconstruction loss sees the mean over the n replicas; `preds` / `latents` are the first replica's."""
    b = x.shape[0]
    h = cnn_forward(x, enc, enc_arch, training=training)
    hx = h if expansion <= 1 else h.unsqueeze(0).expand(expansion, *h.shape).reshape(expansion * b, *h.shape[1:])
    z, prior = gaussian_prior_encode(hx, eps, loss_coeff, step, annealing_steps)
    preds_all = cnn_forward(z, dec, dec_arch, training=training)
    preds_mean = preds_all if expansion <= 1 else preds_all.reshape(expansion, b, *preds_all.shape[1:]).mean(0)
    prior_loss = prior.mean() / float(x[0].numel())
    recon_loss = F.mse_loss(preds_mean, x)
    return dict(loss=recon_loss + prior_loss, recon=recon_loss, prior=prior_loss, preds=preds_all[:b], latents=z[:b],
                preds_mean=preds_mean, enc_out=h)


def adam_step(params: Sequence[Tensor], grads: Sequence[Tensor], exp_avg: Sequence[Tensor],
              exp_avg_sq: Sequence[Tensor], step: int, lr: float = 1e-3, b1: float = 0.9, b2: float = 0.999,
              eps: float = 1e-8) -> None:
    """torch.optim.Adam as configured in model/vae.py:148-151 (no weight decay, no amsgrad); ``step`` is the
    1-based step count *after* increment.  In place."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


# ------------------------------------------------------------------------------------------------
# optimal transport arithmetic
# ------------------------------------------------------------------------------------------------
def sinkhorn_log(a: Tensor, b: Tensor, C: Tensor, reg: float = 1e-5, max_iter: int = 1000,
                 threshold: float = STABILITY_CONST, return_potentials: bool = False):
    """``sinkhorn_log`` (ot/w2_utils.py:276-319).  a [*,N], b [*,M], C [*,N,M] -> pi [*,N,M].
    Stops as soon as the *minimum over the batch* of the L1 change of (u, v) drops below threshold."""
    u = torch.zeros_like(a)
    v = torch.zeros_like(b)
    log_a = torch.log(a + STABILITY_CONST)
    log_b = torch.log(b + STABILITY_CONST)
    Cr = -C / reg
    n_done = 0
    for _ in range(max_iter):
        u0, v0 = u, v
        v = log_b - torch.logsumexp(Cr + u.unsqueeze(-1), dim=-2)
        u = log_a - torch.logsumexp(Cr + v.unsqueeze(-2), dim=-1)
        n_done += 1
        diff = (u - u0).abs().sum(-1) + (v - v0).abs().sum(-1)
        if diff.min().item() < threshold:
            break
    pi = torch.exp(u.unsqueeze(-1) + v.unsqueeze(-2) + Cr)
    if return_potentials:
        return pi, u, v, n_done
    return pi


def sq_euclidean_cost(x: Tensor, y: Tensor) -> Tensor:
    """C_ij = ||x_i - y_j||^2, the cost the Sinkhorn prior of BASELINE configs 3-4 composes with
    ``sinkhorn_log`` (same expansion as ot/w2_utils.py:121-125)."""
    return (x ** 2).sum(-1, keepdim=True) + (y ** 2).sum(-1).unsqueeze(-2) - 2 * (x @ y.transpose(-2, -1))


def sinkhorn_ot_loss(z: Tensor, prior_samples: Tensor, reg: float = 0.05, max_iter: int = 50,
                     threshold: float = 0.0, normalize_cost: bool = True) -> Tensor:
    """Minibatch entropic OT cost sum(C * pi) with uniform marginals, cost optionally divided by its max
    before the solve exactly like ``batch_ot_gmm`` does (ot/w2_utils.py:265-269)."""
    C = sq_euclidean_cost(z, prior_samples)
    n, m = C.shape[-2:]
    a = torch.full(C.shape[:-1], 1.0 / n, dtype=C.dtype)
    b = torch.full((*C.shape[:-2], m), 1.0 / m, dtype=C.dtype)
    Cn = C / C.amax(dim=(-2, -1), keepdim=True) if normalize_cost else C
    pi = sinkhorn_log(a, b, Cn, reg=reg, max_iter=max_iter, threshold=threshold)
    return (C * pi).sum(dim=(-2, -1))


def gaussian_stats(samples: Tensor, diag: bool = False):
    """``GaussianModel._stats`` (ot/distribution_models/gaussian_model.py:144-151), fp64:
    n = B, sum_x = sum_b x, sum_xxT = sum_b x x^T (or x^2 if diag)."""
    s = samples.double()
    n = torch.as_tensor(float(s.size(-2)), dtype=s.dtype)
    sx = s.sum(-2)
    sxx = (s ** 2).sum(-2) if diag else torch.einsum("...bi,...bj->...ij", s, s)
    return n, sx, sxx


def ema(avg: Tensor, new: Tensor, decay: Optional[float]) -> Tensor:
    """utils/__init__.py:204-206."""
    if decay is None:
        return avg + new
    return avg * decay + new * (1 - decay)


class ParamEMARef:
    """The parameter moving average of the reference's ``ema_decay`` option (model/base.py:99,146-190).  The arithmetic lives in the
    third-party ``torch_ema`` package (requirements.txt:10 ``torch-ema``, UNPINNED, absent from /root/reference and from this image):
    this restates its published update rule (torch_ema 0.3 ``ExponentialMovingAverage``, ``use_num_updates=True``) operation by
    operation -- ``tmp = shadow - param; tmp.mul_(1 - d); shadow.sub_(tmp)`` with ``d = min(decay, (1 + n) / (10 + n))`` -- and
    ``store / copy_to / restore`` as copies.  PARITY UNPINNED for this one class: there is no reference-side vector to hold it to."""

    def __init__(self, params, decay: float):
        self.decay, self.num_updates = decay, 0
        self.shadow = [p.clone().detach() for p in params]
        self.collected = None

    def update(self, params):
        self.num_updates += 1
        d = min(self.decay, (1 + self.num_updates) / (10 + self.num_updates))
        one_minus_decay = 1.0 - d
        with torch.no_grad():
            for s_, p in zip(self.shadow, params):
                tmp = s_ - p
                tmp.mul_(one_minus_decay)
                s_.sub_(tmp)

    def store(self, params):
        self.collected = [p.clone() for p in params]

    def copy_to(self, params):
        for s_, p in zip(self.shadow, params):
            p.data.copy_(s_)

    def restore(self, params):
        for c_, p in zip(self.collected, params):
            p.data.copy_(c_)


def mean_cov(sum_x: Tensor, sum_xx: Tensor, n, diag: bool = False):
    """``mean_cov`` (ot/matrix_utils.py:145-158)."""
    n = torch.as_tensor(n, dtype=sum_x.dtype)
    mean = sum_x / n[(...,) + (None,) * (sum_x.dim() - n.dim())]
    cov = sum_xx / n[(...,) + (None,) * (sum_xx.dim() - n.dim())]
    cov = cov - (mean ** 2 if diag else mean.unsqueeze(-1) @ mean.unsqueeze(-2))
    return mean, cov


def _eig_fn(m: Tensor, fn) -> Tensor:
    """``_matrix_operator`` (ot/matrix_utils.py:37-46): V f(lambda) V^T from eigh(UPLO='L')."""
    lam, vec = torch.linalg.eigh(m, UPLO="L")
    return vec @ torch.diag_embed(fn(lam)) @ vec.transpose(-2, -1)


def sqrtm(m: Tensor) -> Tensor:
    return _eig_fn(m, torch.sqrt)


def invsqrtm(m: Tensor) -> Tensor:
    return _eig_fn(m, lambda x: 1.0 / torch.sqrt(x))


def min_eig(m: Tensor) -> Tensor:
    return torch.linalg.eigh(m)[0].min(dim=-1)[0]


def make_psd(m: Tensor, strict: bool = False) -> Tensor:
    """``make_psd`` (ot/matrix_utils.py:123-142): add |min(lambda_min, 0)| (+1e-8 if strict) to the diagonal."""
    shift = min_eig(m).clamp(max=0).abs()
    if strict:
        shift = shift + STABILITY_CONST
    eye = torch.eye(m.shape[-1], dtype=m.dtype)
    return m + eye * shift[..., None, None]


def symmetrize_triu(m: Tensor) -> Tensor:
    """``Symmetric`` parametrisation (gaussian_model.py:220-226): upper triangle mirrored."""
    return m.triu() + m.triu(1).transpose(-1, -2)


def w2_gaussian(mean_s: Tensor, mean_t: Tensor, cov_s: Tensor, cov_t: Tensor, make_pd: bool = False) -> Tensor:
    """``w2_gaussian`` (ot/w2_utils.py:40-80), fp64:
    ||mu_s-mu_t||^2 + tr(S_s + S_t - 2 (S_t^1/2 S_s S_t^1/2)^1/2); with make_pd the two
    covariances are shifted by make_psd(strict) only when one of the batch fails the eigenvalue test
    (w2_utils.py:667-669).  The inner product matrix is validated as 'spsd', for which the reference's
    substring test ``'pd' in 'spsd'`` is False, i.e. it is only checked for symmetry and never shifted."""
    mean_s, mean_t, cov_s, cov_t = (t.double() for t in (mean_s, mean_t, cov_s, cov_t))
    if make_pd:
        if not bool((min_eig(cov_s) > 0).all()):
            cov_s = make_psd(cov_s, strict=True)
        if not bool((min_eig(cov_t) > 0).all()):
            cov_t = make_psd(cov_t, strict=True)
    rt = sqrtm(cov_t)
    mix = rt @ cov_s @ rt
    shift = ((mean_s - mean_t) ** 2).sum(-1)
    tr = torch.diagonal(cov_s + cov_t - 2 * sqrtm(mix), dim1=-2, dim2=-1).sum(-1)
    return shift + tr


def batch_w2_dissimilarity_gaussian(mean_s: Tensor, mean_t: Tensor, cov_s: Tensor, cov_t: Tensor) -> Tensor:
    """``batch_w2_dissimilarity_gaussian`` (ot/w2_utils.py:138-189): D[*, i, j] = W2^2 between full-covariance components."""
    n, m = mean_s.size(-2), mean_t.size(-2)
    ones = [1] * (mean_s.dim() - 2)
    dis = w2_gaussian(mean_s.repeat_interleave(m, -2), mean_t.repeat(*ones, n, 1), cov_s.repeat_interleave(m, -3),
                      cov_t.repeat(*ones, n, 1, 1), make_pd=True)
    return dis.view(*mean_s.shape[:-2], n, m)


def batch_ot_gmm_full(mean_s: Tensor, mean_t: Tensor, cov_s: Tensor, cov_t: Tensor, w_s: Tensor, w_t: Tensor, **sinkhorn_kwargs):
    """``batch_ot_gmm(diag=False)`` (ot/w2_utils.py:197-270)."""
    cost = batch_w2_dissimilarity_gaussian(mean_s, mean_t, cov_s, cov_t)
    cmax = cost.max(-2, keepdim=True)[0].max(-1, keepdim=True)[0]
    plan = sinkhorn_log(w_s, w_t, cost / cmax, **sinkhorn_kwargs)
    return torch.sum(cost * plan, dim=(-2, -1)), plan


def gaussian_barycenter(mean: Tensor, cov: Tensor, weights: Tensor, diag: bool, n_iter: int = 100, init_index: int = 0):
    """``gaussian_barycenter`` (ot/w2_utils.py:325-385) with the start of the fixed point given (the reference draws it)."""
    w_row = weights.unsqueeze(-2)
    mean_b = (w_row @ mean).squeeze(-2)
    if diag:
        return mean_b, ((w_row @ torch.sqrt(cov)) ** 2).squeeze(-2)
    w4 = weights.unsqueeze(-1).unsqueeze(-1)
    cov_b = cov.select(dim=-3, index=init_index).unsqueeze(-3)
    for _ in range(n_iter):
        root = sqrtm(cov_b)
        cov_b = (w4 * sqrtm(root @ cov @ root)).sum(-3, keepdim=True)
    return mean_b, cov_b.squeeze(-3)


def gmm_full_energy(x: Tensor, mean: Tensor, cov: Tensor, weights: Tensor) -> Tensor:
    """``GaussianMixtureModel.energy`` with full covariances (gassian_mixture_model.py:91-99): log N(x; mu_k, C_k) + log w_k."""
    comp = torch.distributions.MultivariateNormal(mean.unsqueeze(-3), covariance_matrix=cov.unsqueeze(-4))
    return comp.log_prob(x.unsqueeze(-2)) + torch.log_softmax(torch.log(weights), -1).unsqueeze(-2)


def gmm_log_prob(x: Tensor, mean: Tensor, cov: Tensor, weights: Tensor, diag: bool) -> Tensor:
    """``GaussianMixtureModel.forward`` / ``predict`` of a fitted model: the class inherits GaussianModel.predict first
    (gaussian_model.py:129-132), i.e. ``batched_distribution.log_prob`` of the MixtureSameFamily (gassian_mixture_model.py:74-80):
    logsumexp_k(log N(x; mu_k, C_k) + log w_k), [*, B]."""
    energy = gmm_diag_energy(x, mean, cov, weights) if diag else gmm_full_energy(x, mean, cov, weights)
    return torch.logsumexp(energy, dim=-1)


def gmm_autograd_log_prob(x: Tensor, mean: Tensor, raw_cov: Tensor, raw_weights: Tensor, diag: bool) -> Tensor:
    """The same log-density for ``update_with_autograd=True`` (gassian_mixture_model.py:53-58; gaussian_model.py:52-55,76-93,186-201):
    the raw `cov` parameter passes through ExpScaleTril (diag: variances = exp(raw), scale = variances ** 0.5; full: Cholesky
    factor = strict lower triangle + exp(diagonal)), the raw weights through a soft-max; differentiable with torch.autograd."""
    w = torch.softmax(raw_weights, -1)
    if diag:
        comp = torch.distributions.Independent(torch.distributions.Normal(mean.unsqueeze(-3), raw_cov.exp().unsqueeze(-3) ** 0.5), 1)
    else:
        tril = raw_cov.tril(-1) + torch.diag_embed(raw_cov.diagonal(dim1=-1, dim2=-2).exp())
        comp = torch.distributions.MultivariateNormal(mean.unsqueeze(-3), scale_tril=tril.unsqueeze(-4))
    return torch.logsumexp(comp.log_prob(x.unsqueeze(-2)) + torch.log_softmax(torch.log(w), -1).unsqueeze(-2), dim=-1)


def w2_prior_loss(z: Tensor, target_mean: Optional[Tensor] = None, target_cov: Optional[Tensor] = None) -> Tensor:
    """Gaussian W2 with empirical covariance as a loss term (BASELINE north_star): the batch statistics of
    ``GaussianModel._stats`` (gaussian_model.py:144-151) -> ``mean_cov`` (matrix_utils.py:145-158) -> ``w2_gaussian``
    (w2_utils.py:40-80, make_pd=True) against N(target_mean, target_cov) (default N(0, I)); differentiable in z with
    torch.autograd through eigh, like the reference's update_with_autograd path (distribution_models/base.py:82-89)."""
    d = z.shape[-1]
    n, sx, sxx = gaussian_stats(z)
    mean, cov = mean_cov(sx, sxx, n)
    tm = torch.zeros(d, dtype=torch.double) if target_mean is None else target_mean.double()
    tc = torch.eye(d, dtype=torch.double) if target_cov is None else target_cov.double()
    return w2_gaussian(mean, tm, cov, tc, make_pd=True)


def transport_operator_full(cov_s: Tensor, cov_t: Tensor, pg_star: float = 0.0) -> Tensor:
    """``_compute_transport_full_mat`` (ot/w2_utils.py:756-769), eq. 17:
    T = (1-pg) S_s^-1/2 (S_s^1/2 S_t S_s^1/2)^1/2 S_s^-1/2 + pg I, with invsqrtm of S_s + 1e-8 I."""
    cov_s, cov_t = cov_s.double(), cov_t.double()
    eye = torch.eye(cov_s.shape[-1], dtype=cov_s.dtype).expand_as(cov_s)
    rs = sqrtm(cov_s)
    irs = invsqrtm(cov_s + STABILITY_CONST * eye)
    return (1 - pg_star) * (irs @ sqrtm(rs @ cov_t @ rs) @ irs) + pg_star * eye


def transport_operator_stochastic(cov_s: Tensor, cov_t: Tensor, pg_star: float = 0.0, diag: bool = False):
    """``_compute_transport_diag_stochastic`` / ``_compute_transport_full_mat_stochastic`` (ot/w2_utils.py:732-751,771-786),
    eq. 19: (T, Cw) for a possibly degenerate source."""
    cov_s, cov_t = cov_s.double(), cov_t.double()
    if diag:
        cov_s = torch.where(cov_s < STABILITY_CONST, torch.zeros_like(cov_s), cov_s)
        t_star = torch.sqrt(cov_s / cov_t + STABILITY_CONST)
        pinv = torch.where(cov_s > STABILITY_CONST, 1 / cov_s.clamp(min=1e-300), torch.zeros_like(cov_s))
        T = (1 - pg_star) * torch.sqrt(cov_t * cov_s) * pinv + pg_star
        return T, math.sqrt(1 - pg_star) * cov_t * (1 - cov_t * pinv * t_star ** 2)
    eye = torch.eye(cov_s.shape[-1], dtype=cov_s.dtype).expand_as(cov_s)
    pinv = torch.linalg.pinv(cov_s)
    rt, irt = sqrtm(cov_t), invsqrtm(cov_t + STABILITY_CONST * eye)
    t_star = transport_operator_full(cov_t, cov_s, 0.0)
    T = (1 - pg_star) * (rt @ sqrtm(rt @ cov_s @ rt) @ irt @ pinv) + pg_star * eye
    return T, math.sqrt(1 - pg_star) * rt @ (eye - rt @ t_star @ pinv @ t_star @ rt) @ rt


def apply_transport(x: Tensor, mean_s: Tensor, mean_t: Tensor, T: Tensor) -> Tensor:
    """``apply_transport`` without noise (ot/w2_utils.py:517-520), fp64: T (x - mu_s) + mu_t."""
    x, mean_s, mean_t, T = (t.double() for t in (x, mean_s, mean_t, T))
    return (T @ (x - mean_s).unsqueeze(-1)).squeeze(-1) + mean_t


def gaussian_fit(n: Tensor, sum_x: Tensor, sum_xx: Tensor):
    """``GaussianModel.fit`` tail (gaussian_model.py:123-126,204-226): mean_cov then the cov parametrisations
    Symmetric -> MakePositiveDefinite(strict=True) as read back through ``model.cov``."""
    mean, cov = mean_cov(sum_x, sum_xx, n)
    return mean, make_psd(symmetrize_triu(cov), strict=True)


def codebook_assign(x: Tensor, codebook: Tensor, temperature: float = 1.0, p: float = 2.0):
    """``CodebookModel.energy`` + ``MixtureMixin.assign`` in 'argmax' mode + ``predict``
    (ot/distribution_models/codebook_model.py:150-160, base.py:216-233).
    x [*,B,d], codebook [*,K,d] -> (one-hot @ codebook [*,B,d], argmax indices [*,B] int64)."""
    energy = 1.0 / (torch.cdist(x, codebook, p) + 1e-8)
    weights = torch.softmax(energy / temperature, dim=-1)
    idx = weights.argmax(-1)
    onehot = F.one_hot(idx, energy.size(-1)).to(weights.dtype)
    return onehot @ codebook, idx


# ------------------------------------------------------------------------------------------------ codebook k-means
def codebook_probs(x: Tensor, codebook: Tensor, temperature: float = 1.0) -> Tensor:
    """``MixtureMixin.assign`` soft weights (base.py:216-224) with ``CodebookModel.energy`` (codebook_model.py:150-156):
    softmax_k((1 / (cdist(x, c) + 1e-8)) / T)."""
    energy = 1 / (torch.cdist(x, codebook, 2.0) + 1e-8)
    return torch.softmax(energy / temperature, dim=-1)


def codebook_forward(x: Tensor, codebook: Tensor, temperature: float = 1.0, mode: str = "mean"):
    """``CodebookModel.predict`` (codebook_model.py:145-148) on ``MixtureMixin.assign`` (base.py:206-239) with everything kept in
    the autograd graph -- what ``update_with_autograd=True`` trains through (codebook_model.py:89): (weights @ codebook, probs,
    entropy of Categorical(probs)); 'mean': weights = probs, 'argmax': their one-hot arg-max."""
    probs = codebook_probs(x, codebook, temperature)
    weights = probs if mode == "mean" else F.one_hot(probs.argmax(-1), probs.size(-1)).type_as(probs)
    return weights @ codebook, probs, torch.distributions.Categorical(probs).entropy()


def codebook_prior_encode_soft_kl(z: Tensor, codebook: Tensor, temperature: float):
    """``CodebookPrior.encode`` (prior/codebook.py:86-105) for embed_dims=(1,), loss='kl', the soft 'mean' mode (commitment cost
    0.1), a trained codebook [1, K, C]: z [B, C, H, W] -> vectors [H*W, B, C]; (encodings [B, C, H, W], loss [B])."""
    b, c, h, w = z.shape
    x = z.permute(2, 3, 0, 1).reshape(h * w, b, c)
    preds, probs, ent = codebook_forward(x, codebook, temperature, "mean")
    loss = (math.log(codebook.shape[-2]) - ent).sum(0)
    loss = loss + 0.1 * ((preds - x.detach()) ** 2).mean(-1).sum(0)
    return preds.reshape(h, w, b, c).permute(2, 3, 0, 1), loss


def codebook_energy_general(x: Tensor, codebook: Tensor, metric: str = "euclidean", p: float = 2.0) -> Tensor:
    """``CodebookModel.energy`` (codebook_model.py:150-168): 'euclidean': 1 / (cdist_p + 1e-8); 'cosine':
    |x . c| / ((sum |x|^p)(sum |c|^p) + 1e-8)^(1/p).  [*, B, d] x [*, K, d] -> [*, B, K]."""
    if metric == "euclidean":
        return 1 / (torch.cdist(x, codebook, p) + 1e-8)
    norm_x = x.abs().pow(p).sum(-1, keepdim=True)
    norm_c = codebook.abs().pow(p).sum(-1).unsqueeze(-2)
    dot = (x @ codebook.transpose(-2, -1)).abs()
    return dot / (norm_x * norm_c + 1e-8) ** (1 / p)


def mixture_assign(energy: Tensor, topk: Optional[int], temperature: float, mode: str):
    """``MixtureMixin.assign`` (base.py:206-239) without the random draw: top-k restriction, soft-max, and the weights of the
    deterministic modes ('mean', 'argmax'; with topk == 1 the soft weights whatever the mode).  -> (weights, probs)"""
    if topk is not None and topk > 0:
        val, idx = torch.topk(energy, topk, dim=-1)
        energy = torch.full_like(energy, float("-inf")).scatter(-1, idx, val)
    probs = torch.softmax(energy / temperature, dim=-1)
    if mode == "mean" or topk == 1:
        return probs, probs
    assert mode == "argmax", mode
    return F.one_hot(probs.argmax(-1), probs.size(-1)).type_as(probs), probs


def codebook_kmeans_stats(x: Tensor, codebook: Tensor, temperature: float = 1.0, mode: str = "argmax"):
    """``kmean_iteration`` (base.py:241-252) with one-hot ('argmax') or soft ('mean') weights:
    (counts [*, K], sums [*, K, d])."""
    w = codebook_probs(x, codebook, temperature)
    if mode == "argmax":
        w = torch.nn.functional.one_hot(w.argmax(-1), codebook.shape[-2]).type_as(x)
    return w.sum(-2), w.transpose(-1, -2) @ x


def _laplace(x: Tensor, n_categories: int, eps: Optional[float] = 1e-5) -> Tensor:
    """utils/__init__.py:209-218"""
    if eps is None:
        return x
    return (x + eps) / (x.sum(-1, keepdim=True) + n_categories * eps) * x.sum(-1, keepdim=True)


def codebook_update(state: Dict[str, Tensor], x: Tensor, decay: Optional[float], rand_indices: Optional[Tensor] = None,
                    temperature: float = 1.0, laplace_eps: Optional[float] = 1e-5, mode: str = "argmax") -> Dict[str, Tensor]:
    """``CodebookModel.update`` (codebook_model.py:121-130, 189-214) on a state {codebook, vec_init, n_obs, running_sum}:
    first-call initialisation from ``rand_indices`` (the reference draws them with torch.randperm), one-hot k-means
    statistics, (EMA) accumulation into the buffers of the observed atoms, codebook = running_sum / smoothed counts."""
    st = {k: v.clone() for k, v in state.items()}
    K = st["codebook"].shape[-2]
    if torch.allclose(st["codebook"], st["vec_init"]):
        st["codebook"] = x[..., rand_indices, :].clone()
        st["n_obs"] = st["n_obs"] + 1
    counts, sums = codebook_kmeans_stats(x, st["codebook"], temperature, mode)
    hit = counts > 1e-8
    st["n_obs"][hit] = ema(st["n_obs"][hit], counts[hit], decay)
    st["running_sum"][hit] = ema(st["running_sum"][hit], sums[hit], decay)
    hit2 = st["n_obs"] > 1e-8
    st["codebook"][hit2] = st["running_sum"][hit2] / _laplace(st["n_obs"][hit2], K, laplace_eps).unsqueeze(-1)
    return st


def codebook_fit(state: Dict[str, Tensor], laplace_eps: Optional[float] = 1e-5) -> Dict[str, Tensor]:
    """``CodebookModel.fit()`` without samples (codebook_model.py:132-143): the codebook recomputed from the buffers."""
    st = {k: v.clone() for k, v in state.items()}
    K = st["codebook"].shape[-2]
    hit = st["n_obs"] > 1e-8
    st["codebook"][hit] = st["running_sum"][hit] / _laplace(st["n_obs"][hit], K, laplace_eps).unsqueeze(-1)
    return st


def codebook_w2(codebook: Tensor, weights: Tensor, other_atoms: Tensor, other_probs: Tensor) -> Tensor:
    """``CodebookModel.w2`` (codebook_model.py:175-182): entropic OT (reg 1e-5, 100 iterations, threshold 1e-3) between
    the atom sets with cost 1 / (energy + 1e-8)."""
    cost = 1 / (1 / (torch.cdist(other_atoms, codebook, 2.0) + 1e-8) + 1e-8)
    plan = sinkhorn_log(weights, other_probs, cost, reg=1e-5, max_iter=100, threshold=1e-3)
    return torch.sum(cost * plan, dim=(-2, -1))


# ------------------------------------------------------------------------------------------------ discrete transport
def codebook_weights(n_obs: Tensor) -> Tensor:
    """``CodebookModel.weights`` (codebook_model.py:96-100): uniform before the first observation, else n_obs / sum."""
    if torch.allclose(n_obs, torch.zeros_like(n_obs)):
        return torch.ones_like(n_obs) / n_obs.shape[-1]
    return n_obs / n_obs.sum(-1, keepdim=True)


def discrete_transport_compute(source_codebook: Tensor, source_probs: Tensor, target_codebook: Tensor, target_probs: Tensor,
                               reg: float = 1e-5, max_iter: int = 1000, threshold: float = 1e-6):
    """``DiscreteTransport.compute`` after the models are fitted (ot/transport/discrete_transport.py:56-69):
    cost = source energy of the target atoms = 1 / (cdist(target, source) + 1e-8)  [K_t x K_s as the reference lays it
    out: rows are the atoms handed to ``energy``], plan = sinkhorn_log(source probs, target probs, cost), total = <cost, plan>."""
    cost = 1 / (torch.cdist(target_codebook, source_codebook, 2.0) + 1e-8)
    plan = sinkhorn_log(source_probs, target_probs, cost, reg=reg, max_iter=max_iter, threshold=threshold)
    return torch.sum(cost * plan, dim=(-2, -1)), plan


def discrete_transport_apply(x: Tensor, source_codebook: Tensor, plan: Tensor, target_codebook: Tensor,
                             temperature: float = 1.0, inference_mode: str = "argmax", transport_type: str = "mean") -> Tensor:
    """``DiscreteTransport.transport`` (discrete_transport.py:71-95) for the deterministic choices: inputs -> assignment
    to source atoms (inference mode 'argmax' or 'mean') -> @ plan -> ('argmax': one-hot of the best coupled target atom)
    -> @ target atoms."""
    w = codebook_probs(x, source_codebook, temperature)
    if inference_mode == "argmax":
        w = F.one_hot(w.argmax(-1), w.size(-1)).type_as(w)
    moved = w @ plan
    if transport_type == "argmax":
        moved = F.one_hot(moved.argmax(-1), moved.size(-1)).type_as(moved)
    return moved @ target_codebook


def codebook_prior_encode(x: Tensor, codebook: Tensor, temperature: float = 1.0, loss: Optional[str] = None,
                          coeff: float = 1.0):
    """``CodebookPrior.encode`` + ``Prior.forward`` scaling (prior/codebook.py:75-105, prior/base.py:74-78) for a latent
    embedded as a whole (x [B, dim], one shared codebook [1, K, dim]) in a one-hot mode: (z = straight-through
    encodings [1, B, dim], loss [B] * coeff, assignment probabilities [1, B, K])."""
    probs = codebook_probs(x.detach(), codebook, temperature)                    # [1, B, K]
    enc = F.one_hot(probs.argmax(-1), probs.size(-1)).type_as(x) @ codebook
    if loss is None:
        val = torch.zeros(x.size(-2)).type_as(x)
    elif loss == "l2":
        val = F.mse_loss(x.expand_as(enc), enc.detach(), reduction="none").mean(-1).sum(0)
    else:
        gap = math.log(codebook.shape[-2]) - torch.distributions.Categorical(probs).entropy()
        val = gap.sum(0) if loss == "kl" else gap[0]
    z = x + (enc - x).detach()
    return z, val * coeff, probs


def codebook_prior_encode_soft(x: Tensor, codebook: Tensor, temperature: float = 1.0, loss: Optional[str] = "kl",
                               coeff: float = 1.0, commitment: float = 0.1):
    """``CodebookPrior.encode`` in the soft 'mean' training mode (prior/codebook.py:75-105): encodings = softmax weights @
    codebook (differentiable in x), entropy loss, + commitment * mse(encodings, sg[x])."""
    probs = codebook_probs(x, codebook, temperature)                             # [1, B, K]
    enc = probs @ codebook
    if loss is None:
        val = torch.zeros(x.size(-2)).type_as(x)
    elif loss == "l2":
        val = F.mse_loss(x.expand_as(enc), enc.detach(), reduction="none").mean(-1).sum(0)
    else:
        gap = math.log(codebook.shape[-2]) - torch.distributions.Categorical(probs).entropy()
        val = gap.sum(0) if loss == "kl" else gap[0]
    val = val + commitment * F.mse_loss(enc, x.detach().expand_as(enc), reduction="none").mean(-1).sum(0)
    return enc, val * coeff, probs


def gumbel_assign(energy: Tensor, gumbel: Tensor, temperature: float, hard: bool) -> Tensor:
    """``MixtureMixin.assign`` in the Gumbel modes (base.py:234-235, F.gumbel_softmax) for given Gumbel draws."""
    soft = torch.softmax((energy + gumbel) / temperature, dim=-1)
    if not hard:
        return soft
    idx = soft.max(-1, keepdim=True)[1]
    return torch.zeros_like(soft).scatter_(-1, idx, 1.0) - soft.detach() + soft


# ------------------------------------------------------------------------------------------------ Gaussian mixtures (diag)
def gmm_diag_energy(x: Tensor, mean: Tensor, var: Tensor, weights: Tensor) -> Tensor:
    """``GaussianMixtureModel.energy`` (ot/distribution_models/gassian_mixture_model.py:91-99) for diagonal covariances:
    log N(x_b; mean_k, diag var_k) + log w_k.  x [*, B, d], mean / var [*, K, d], weights [*, K] -> [*, B, K]."""
    comp = torch.distributions.Independent(torch.distributions.Normal(mean.unsqueeze(-3), var.unsqueeze(-3) ** 0.5), 1)
    return comp.log_prob(x.unsqueeze(-2)) + torch.log_softmax(torch.log(weights.unsqueeze(-2)), dim=-1)


def gmm_assign(x: Tensor, mean: Tensor, var: Tensor, weights: Tensor, temperature: float = 1.0, mode: str = "argmax") -> Tensor:
    """``MixtureMixin.assign`` weights (base.py:216-235) on the mixture energy: soft-max, one-hot of its arg-max in 'argmax'."""
    w = torch.softmax(gmm_diag_energy(x, mean, var, weights) / temperature, dim=-1)
    if mode == "argmax":
        w = F.one_hot(w.argmax(-1), w.size(-1)).type_as(w)
    return w


def _cov_read(raw: Tensor) -> Tensor:
    """the ``MakePositiveDefinite(diag=True, strict=True)`` parametrisation every read of ``cov`` goes through
    (gaussian_model.py:204-214, matrix_utils.py:123-142): + |min(var, 0)| + 1e-8 per component"""
    return raw + (raw.min(-1)[0].clamp(max=0).abs() + STABILITY_CONST)[..., None]


def _norm_sum(raw: Tensor) -> Tensor:
    """the ``NormSum`` parametrisation of the mixture weights (gassian_mixture_model.py:180-189)"""
    return raw / raw.sum(-1, keepdim=True)


def _gmm_update_parameters(st: Dict[str, Tensor], n_obs: Tensor, s1: Tensor, s2: Tensor, laplace_eps: Optional[float]):
    """``_update_parameters`` (gassian_mixture_model.py:146-151): mean / variance / weight of the observed components from
    Laplace-smoothed counts; parameters are read back through their parametrisations and written raw, as the reference."""
    K = n_obs.shape[-1]
    n = _laplace(n_obs, K, laplace_eps)
    if bool((n == 0).all()):
        return
    seen = n > 1e-8
    mean, cov = mean_cov(s1[seen], s2[seen], n[seen], diag=True)
    st["mean"][seen] = mean
    tmp = _cov_read(st["cov_raw"])
    tmp[seen] = cov
    st["cov_raw"] = tmp
    tmp = _norm_sum(st["w_raw"])
    tmp[seen] = n_obs[seen]
    st["w_raw"] = tmp


def gmm_update(state: Dict[str, Tensor], x: Tensor, decay: Optional[float], rand_indices: Optional[Tensor] = None,
               temperature: float = 1.0, laplace_eps: Optional[float] = 1e-5, mode: str = "argmax") -> Dict[str, Tensor]:
    """``GaussianMixtureModel.update`` (= ``CodebookModel.update``, codebook_model.py:121-130, with the mixture's
    kmean_iteration / _update_buffers / _update_parameters, gassian_mixture_model.py:108-176) on a state
    {mean, vec_init, cov_raw, w_raw, n_obs, s1, s2}."""
    st = {k: v.clone() for k, v in state.items()}
    if torch.allclose(st["mean"], st["vec_init"]):
        st["mean"] = x[..., rand_indices, :].clone()
        st["n_obs"] = st["n_obs"] + 1
    w = gmm_assign(x, st["mean"], _cov_read(st["cov_raw"]), _norm_sum(st["w_raw"]), temperature, mode)
    ws, s1, s2 = w.sum(-2), w.transpose(-1, -2) @ x, w.transpose(-1, -2) @ (x ** 2)
    hit = ws > 1e-8
    st["n_obs"][hit] = ema(st["n_obs"][hit], ws[hit], decay)
    st["s1"][hit] = ema(st["s1"][hit], s1[hit], decay)
    st["s2"][hit] = ema(st["s2"][hit], s2[hit], decay)
    _gmm_update_parameters(st, st["n_obs"], st["s1"], st["s2"], laplace_eps)
    return st


def gmm_fit(state: Dict[str, Tensor], laplace_eps: Optional[float] = 1e-5, iters: int = 100) -> Dict[str, Tensor]:
    """``fit()`` without samples (codebook_model.py:132-143): the parameters recomputed from the buffers -- once per
    k-means iteration in the reference, and every recomputation reads the parameters back through their parametrisations
    (so the unobserved components' variances creep by 1e-8 per iteration: ``iters`` matters)."""
    st = {k: v.clone() for k, v in state.items()}
    for _ in range(iters):
        _gmm_update_parameters(st, st["n_obs"], st["s1"], st["s2"], laplace_eps)
    return st


def batch_ot_gmm_diag(mean_s: Tensor, mean_t: Tensor, var_s: Tensor, var_t: Tensor, w_s: Tensor, w_t: Tensor, **sinkhorn_kwargs):
    """``batch_ot_gmm`` with diag=True (ot/w2_utils.py:197-270): ground cost = componentwise Gaussian W2^2
    (batch_w2_dissimilarity_gaussian_diag, :164-191), coupling = sinkhorn_log on the cost over its maximum."""
    cost = sq_euclidean_cost(mean_s, mean_t) + sq_euclidean_cost(var_s.sqrt(), var_t.sqrt())
    cmax = cost.max(-2, keepdim=True)[0].max(-1, keepdim=True)[0]
    coupling = sinkhorn_log(w_s, w_t, cost / cmax, **sinkhorn_kwargs)
    return torch.sum(cost * coupling, dim=(-2, -1)), coupling


def gmm_transport_apply(x: Tensor, src: Dict[str, Tensor], tgt: Dict[str, Tensor], coupling: Tensor) -> Tensor:
    """``GMMTransport.transport`` with transport_type='argmax' (ot/transport/gmm_transport.py:82-118): per input the
    likeliest source component, the target component it is most coupled with, and the diagonal Gaussian map between
    the two, T = sqrt(var_t / var_s + 1e-8) (w2_utils.py:737-741)."""
    xs = x.to(coupling.dtype)
    a = gmm_assign(xs, src["mean"], src["cov"], src["weights"])
    ms, vs = a @ src["mean"], a @ src["cov"]
    moved = a @ coupling
    b = F.one_hot(moved.argmax(-1), moved.size(-1)).type_as(moved)
    mt, vt = b @ tgt["mean"], b @ tgt["cov"]
    T = torch.sqrt(vt / vs + STABILITY_CONST)
    return (T * (xs - ms) + mt).type_as(x)


# ------------------------------------------------------------------------------------------------ ViT
def transformer_encoder_layer(x: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, causal: bool = False) -> Tensor:
    """One post-norm ``nn.TransformerEncoderLayer(dim, heads, mlp_dim, dropout=0, batch_first=True)`` as the reference's ViT
    builds them (networks/vit.py:169-172): x = LN1(x + out_proj(MHA(x))); x = LN2(x + linear2(relu(linear1(x)))), with
    ``nn.MultiheadAttention``'s packed in-projection (q | k | v, head-major channels) and 1/sqrt(head width) scores."""
    d = x.shape[-1]
    x = F.layer_norm(x + _mha(x, x, p, prefix + "self_attn.", heads, causal), (d,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], 1e-5)
    ff = F.linear(torch.relu(F.linear(x, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"])),
                  p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])
    return F.layer_norm(x + ff, (d,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], 1e-5)


def _mha(xq: Tensor, xkv: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, causal: bool = False) -> Tensor:
    """``nn.MultiheadAttention(xq, xkv, xkv)`` with the packed in-projection: q from the first third of in_proj on xq, k | v from
    the other two thirds on xkv; ``causal``: the -inf upper triangle of nn.Transformer.generate_square_subsequent_mask."""
    n, tq, d = xq.shape
    tk, hd = xkv.shape[1], d // heads
    w, b = p[prefix + "in_proj_weight"], p[prefix + "in_proj_bias"]
    q = F.linear(xq, w[:d], b[:d]).reshape(n, tq, heads, hd).transpose(1, 2)
    k = F.linear(xkv, w[d:2 * d], b[d:2 * d]).reshape(n, tk, heads, hd).transpose(1, 2)
    v = F.linear(xkv, w[2 * d:], b[2 * d:]).reshape(n, tk, heads, hd).transpose(1, 2)
    sc = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if causal:
        sc = sc + torch.full((tq, tk), float("-inf"), dtype=sc.dtype).triu(1)
    att = torch.softmax(sc, dim=-1) @ v
    return F.linear(att.transpose(1, 2).reshape(n, tq, d), p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"])


def transformer_decoder_layer(x: Tensor, memory: Tensor, p: Dict[str, Tensor], prefix: str, heads: int, causal: bool = False) -> Tensor:
    """One post-norm ``nn.TransformerDecoderLayer(dim, heads, mlp_dim, dropout=0, batch_first=True)`` of the reference's
    cross-attention ViT (networks/vit.py:176-181): x = LN1(x + self_attn(x)); x = LN2(x + multihead_attn(x, memory, memory));
    x = LN3(x + linear2(relu(linear1(x))))."""
    d = x.shape[-1]
    x = F.layer_norm(x + _mha(x, x, p, prefix + "self_attn.", heads, causal), (d,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], 1e-5)
    x = F.layer_norm(x + _mha(x, memory, p, prefix + "multihead_attn.", heads), (d,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], 1e-5)
    ff = F.linear(torch.relu(F.linear(x, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"])),
                  p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])
    return F.layer_norm(x + ff, (d,), p[prefix + "norm3.weight"], p[prefix + "norm3.bias"], 1e-5)


def vit_forward(x: Tensor, p: Dict[str, Tensor], *, image_size: int, patch_size: int, dim: int, depth: int, heads: int,
                channels: int, n_embed_tokens: Optional[int], n_input_tokens: Optional[int], patch_to_embed: bool,
                embed_to_patch: bool, labels: Optional[Tensor] = None, preprocess_depth: Optional[int] = None,
                causal_mask: bool = False, output_tokens: str = "embed") -> Tensor:
    """``ViT.forward`` (networks/vit.py:225-246) for output_tokens='embed', no time token, dropout 0:
    [patchify + Linear] -> append the learned embed tokens (and the class token) -> + positions, LayerNorm
    (PositionalEmbedding :41-58) -> the encoder layers -> the embed tokens -> [Linear + un-patchify].
    ``preprocess_depth`` (:171-181, 240-244): the embed tokens are the target of decoder layers whose memory is the other tokens
    after ``preprocess_depth`` encoder layers (state_dict prefix ``prepocess.``, the reference's spelling)."""
    ps = patch_size
    nh = image_size // ps
    num_patches = nh * nh
    if patch_to_embed:
        b, c = x.shape[:2]
        x = x.reshape(b, c, nh, ps, nh, ps).permute(0, 2, 4, 3, 5, 1).reshape(b, num_patches, ps * ps * c)
        x = F.linear(x, p["patch_to_embed.1.weight"], p["patch_to_embed.1.bias"])
    n_in = num_patches if n_input_tokens is None else n_input_tokens
    n_emb = num_patches if n_embed_tokens is None else n_embed_tokens
    if n_emb > 0:
        x = torch.cat((x, p["embed_token"].expand(x.size(0), -1, -1)), dim=1)
    if labels is not None:
        x = torch.cat((x, p["class_token.weight"][labels].unsqueeze(1)), dim=1)
    x = x + p["positional_embed.position_embeddings.weight"][:x.size(1)].unsqueeze(0)
    x = F.layer_norm(x, (dim,), p["positional_embed.LayerNorm.weight"], p["positional_embed.LayerNorm.bias"], 1e-5)
    lo, hi = (n_in, n_in + n_emb) if output_tokens == "embed" else (0, n_in)   # token order: input, embed, class
    if preprocess_depth is None:
        for i in range(depth):
            x = transformer_encoder_layer(x, p, f"transformer.layers.{i}.", heads, causal_mask)
        out = x[:, lo:hi]
    else:
        out = x[:, lo:hi]
        memory = torch.cat((x[:, :lo], x[:, hi:]), dim=1)
        for i in range(preprocess_depth):
            memory = transformer_encoder_layer(memory, p, f"prepocess.layers.{i}.", heads)
        for i in range(depth):
            out = transformer_decoder_layer(out, memory, p, f"transformer.layers.{i}.", heads, causal_mask)
    if embed_to_patch:
        out = out[:, -num_patches:]
        out = F.linear(out, p["embed_to_patch.0.weight"], p["embed_to_patch.0.bias"])
        b = out.size(0)
        out = out.reshape(b, nh, nh, ps, ps, channels).permute(0, 5, 1, 3, 2, 4).reshape(b, channels, nh * ps, nh * ps)
    return out


def autoregressive_forward(tokens: Tensor, p: Dict[str, Tensor], **vit_kwargs) -> Tensor:
    """``AutoRegressive.forward`` (networks/vit.py:249-260): vocabulary embedding -> the ViT on the embedded tokens -> Linear head"""
    hs = vit_forward(p["vocab_embed.weight"][tokens], p, **vit_kwargs)
    return F.linear(hs, p["head.weight"], p["head.bias"])


# ------------------------------------------------------------------------------------------------ conditional prior, ViT VAE
def cond_gaussian_prior_encode(x: Tensor, eps: Tensor, mu_weight: Tensor, log_std_weight: Tensor, labels: Tensor,
                               loss_coeff: float = 1.0, step: int = 0, annealing_steps: int = 0) -> Tuple[Tensor, Tensor]:
    """``ConditionalGaussianPrior.encode`` + ``Prior.forward`` (prior/conditional_gaussian.py:81-93, prior/base.py:74-78)
    with the N(0,1) draw explicit: q = N(mu, exp(log_var/2)) from chunk(x, 2, dim=1), p = N(mu_y, exp(log_std_y)) from
    the class embeddings, z = mu + eps std, loss = sum KL(q || p) over the non-batch dims (torch's Normal-Normal KL)."""
    mu, log_var = torch.chunk(x, 2, dim=1)
    std = (log_var / 2).exp()
    z = mu + eps * std
    pm = mu_weight[labels].reshape(mu.shape)
    ps = log_std_weight[labels].reshape(mu.shape).exp()
    var_ratio = (std / ps) ** 2
    t1 = ((mu - pm) / ps) ** 2
    kl = torch.sum(0.5 * (var_ratio + t1 - 1 - var_ratio.log()), dim=list(range(1, mu.dim())))
    return z, kl * (loss_coeff * prior_annealing(step, annealing_steps))


def cond_prior_ema_update(state: Dict[str, Tensor], x: Tensor, labels: Tensor, decay: float, eps_smooth: float = 1e-5):
    """``ConditionalGaussianPrior.ema_update`` (prior/conditional_gaussian.py:98-113) on {size, mu_avg, log_std_avg}:
    per-class sums of the posterior means / log standard deviations, EMA into the buffers, embeddings = averages over
    Laplace-smoothed class counts.  Returns (new state, mu embedding, log_std embedding)."""
    st = {k: v.clone() for k, v in state.items()}
    C = st["size"].shape[0]
    mu, log_var = torch.chunk(x, 2, dim=1)
    one_hot = F.one_hot(labels, num_classes=C).type(mu.dtype)
    st["size"] = st["size"] * decay + one_hot.sum(0) * (1 - decay)
    st["mu_avg"] = st["mu_avg"] * decay + (one_hot.t() @ mu.flatten(1)) * (1 - decay)
    st["log_std_avg"] = st["log_std_avg"] * decay + (one_hot.t() @ (log_var / 2).exp().log().flatten(1)) * (1 - decay)
    sizes = _laplace(st["size"], C, eps_smooth)
    return st, st["mu_avg"] / sizes.unsqueeze(-1), st["log_std_avg"] / sizes.unsqueeze(-1)


def vit_vae_nelbo(x: Tensor, eps: Tensor, labels: Tensor, enc: Dict[str, Tensor], dec: Dict[str, Tensor], mu_weight: Tensor,
                  log_std_weight: Tensor, cfg: dict, loss_coeff: float, step: int, annealing_steps: int):
    """``VAE.nelbo`` (model/vae.py:165-189) of the conditional ViT VAE of tests/test_conditional_vit_vae.py: the encoder
    ViT's two embed tokens are (mu | log_var), the decoder ViT turns the latent token back into the image; both receive
    the class token, the prior is conditioned on the label."""
    vit = dict(image_size=cfg["image_size"], patch_size=cfg["patch_size"], dim=cfg["dim"], depth=cfg["depth"], heads=cfg["heads"],
               channels=cfg["channels"], labels=labels)
    h = vit_forward(x, enc, n_embed_tokens=2, n_input_tokens=None, patch_to_embed=True, embed_to_patch=False, **vit)
    z, prior = cond_gaussian_prior_encode(h, eps, mu_weight, log_std_weight, labels, loss_coeff, step, annealing_steps)
    preds = vit_forward(z, dec, n_embed_tokens=None, n_input_tokens=1, patch_to_embed=False, embed_to_patch=True, **vit)
    prior_loss = prior.mean() / float(x[0].numel())
    recon_loss = F.mse_loss(preds, x)
    return dict(loss=recon_loss + prior_loss, recon=recon_loss, prior=prior_loss, preds=preds, latents=z)
