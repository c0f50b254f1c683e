"""PyTorch custom operators (``torch.library``) over the C ABI: the fused ops of the hot path as dispatcher-visible
``torch.ops.otvae.*`` with ``register_fake`` shape functions and ``register_autograd`` formulas, so
that an unmodified ``training_step`` / ``loss.backward()`` runs them, the profiler names them, and tracing sees opaque ops
with known output shapes.  Every forward op has a ``*_backward`` op of its own; nothing here computes outside the HIP library.

    otvae::qkv_attention            QKVAttention.forward                         networks/nets_utils.py:63-82
    otvae::bn_batch_stats +         ConvLayer.forward (BN -> act -> up -> conv)  networks/cnn.py:183-192
    otvae::conv_bn_act
    otvae::gaussian_prior           GaussianPrior.encode (+ loss coefficient)    prior/gaussian.py:63-96, prior/base.py:74-78
    otvae::nelbo_loss               VAE.nelbo's reduction                        model/vae.py:158-176
    otvae::sinkhorn_prior           sq. euclidean cost -> sinkhorn_log -> <C,pi> ot/w2_utils.py:265-269,276-319
    otvae::gaussian_w2_prior        _stats -> mean_cov -> w2_gaussian            gaussian_model.py:144-157, matrix_utils.py:145-158,
                                                                                 w2_utils.py:40-80

The modules call these through ``functional`` (``qkv_attention``, ``gaussian_prior``, ``nelbo_loss``, the two OT priors).
``ConvBlock`` runs its two branches and the training engine's in-place gradient slots through the packed variant of the
same kernels (``functional.conv_layers``); ``otvae::conv_bn_act`` is the single-layer functional form.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, ptr, stream

__all__ = ["OPS"]

OPS = ("qkv_attention", "bn_batch_stats", "conv_bn_act", "gaussian_prior", "nelbo_loss", "sinkhorn_prior", "gaussian_w2_prior")
_lib_def = torch.library.Library("otvae", "DEF")


def _nhwc(n, c, h, w, like, dtype=None):
    return torch.empty((n, h, w, c), device=like.device, dtype=dtype or like.dtype).permute(0, 3, 1, 2)


def _define(name: str, schema: str, impl, fake, tags=()):
    _lib_def.define(f"{name}{schema}", tags=tags)
    torch.library.impl(f"otvae::{name}", "CUDA")(impl)
    torch.library.register_fake(f"otvae::{name}")(fake)


# ------------------------------------------------------------------------------------------------ attention
def _attn_fwd(qkv: Tensor, heads: int, scale: float, need_aux: bool):
    from .functional import as_nhwc
    lib = _lib.load()
    qkv = as_nhwc(qkv)
    n, width, h, w = qkv.shape
    t, c = h * w, width // (3 * heads)
    out = _nhwc(n, heads * c, h, w, qkv)
    lse = torch.empty((n, heads, t), device=qkv.device, dtype=torch.float32)
    aux = torch.empty((n, heads, t, c * c) if (need_aux and c <= 2) else (0,), device=qkv.device, dtype=torch.float32)
    check(lib.otvae_attn_fwd_scaled(ptr(qkv), n, t, heads, c, float(scale), ptr(out), ptr(lse), ptr(aux) if aux.numel() else None,
                                    stream()), "otvae_attn_fwd")
    return out, lse, aux


def _attn_fwd_fake(qkv, heads, scale, need_aux):
    n, width, h, w = qkv.shape
    c = width // (3 * heads)
    return (_nhwc(n, heads * c, h, w, qkv), qkv.new_empty((n, heads, h * w)),
            qkv.new_empty((n, heads, h * w, c * c) if (need_aux and c <= 2) else (0,)))


def _attn_bwd(gout: Tensor, qkv: Tensor, out: Tensor, lse: Tensor, aux: Tensor, heads: int, scale: float):
    from .functional import as_nhwc
    lib = _lib.load()
    qkv = as_nhwc(qkv)
    n, width, h, w = qkv.shape
    t, c = h * w, width // (3 * heads)
    gout = as_nhwc(gout)
    gqkv = torch.empty_strided(qkv.shape, qkv.stride(), device=qkv.device, dtype=qkv.dtype)
    check(lib.otvae_attn_bwd_scaled(ptr(qkv), ptr(out), ptr(lse), ptr(gout), ptr(aux) if aux.numel() else None, n, t, heads, c,
                                    float(scale), ptr(gqkv), stream()), "otvae_attn_bwd")
    return gqkv


_define("qkv_attention", "(Tensor qkv, int heads, float scale, bool need_aux) -> (Tensor, Tensor, Tensor)", _attn_fwd, _attn_fwd_fake)
_define("qkv_attention_backward", "(Tensor gout, Tensor qkv, Tensor out, Tensor lse, Tensor aux, int heads, float scale) -> Tensor",
        _attn_bwd, lambda gout, qkv, out, lse, aux, heads, scale: torch.empty_strided(qkv.shape, qkv.stride(), device=qkv.device,
                                                                                      dtype=qkv.dtype))


def _attn_setup(ctx, inputs, output):
    qkv, heads, scale, _ = inputs
    out, lse, aux = output
    ctx.save_for_backward(qkv, out, lse, aux)
    ctx.cfg = (heads, scale)
    ctx.set_materialize_grads(False)  # lse / aux never carry a gradient: no zero-fill launches for them in every backward pass


def _attn_backward(ctx, gout, _glse, _gaux):
    if gout is None:
        return None, None, None, None
    qkv, out, lse, aux = ctx.saved_tensors
    return torch.ops.otvae.qkv_attention_backward(gout, qkv, out, lse, aux, *ctx.cfg), None, None, None


torch.library.register_autograd("otvae::qkv_attention", _attn_backward, setup_context=_attn_setup)


# ------------------------------------------------------------------------------------------------ ConvLayer
# Two ops, like aten's native_batch_norm split: ``bn_batch_stats`` (mutates the running buffers, not differentiable on its own)
# hands (mean, invstd, scale, shift) to the functional, differentiable ``conv_bn_act``, whose backward is the whole
# BatchNorm + activation + up-sampling + convolution adjoint.
def _bn_stats(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
              num_batches_tracked: Optional[Tensor], training: bool):
    from . import functional as HF
    _lib.require_cuda(x, "BatchNorm input")
    br = HF.BNBranch(gamma, beta, running_mean, running_var, num_batches_tracked)
    if training:
        mean, invstd, (scale,), (shift,) = HF.bn_batch_stats(HF.as_nhwc(x), [br])
        return mean, invstd, scale, shift
    return HF.bn_eval_affine(br)


_define("bn_batch_stats", "(Tensor x, Tensor gamma, Tensor beta, Tensor(a!)? running_mean, Tensor(b!)? running_var, "
        "Tensor(c!)? num_batches_tracked, bool training) -> (Tensor, Tensor, Tensor, Tensor)", _bn_stats,
        lambda x, gamma, beta, rm, rv, nbt, training: tuple(x.new_empty(x.shape[1]) for _ in range(4)))


def _conv_fwd(x: Tensor, weight: Tensor, bias: Optional[Tensor], gamma: Optional[Tensor], beta: Optional[Tensor],
              mean: Optional[Tensor], invstd: Optional[Tensor], scale: Optional[Tensor], shift: Optional[Tensor],
              residual: Optional[Tensor], stride: int, pad: int, up: int, relu: bool, training: bool):
    from . import functional as HF
    _lib.require_cuda(x, "conv input")
    if x.dtype != torch.float32:
        raise TypeError("the MI355X conv path computes in fp32")
    x = HF.as_nhwc(x)
    wt = weight if HF.is_hwio(weight) else HF.hwio_weight(weight)
    has_norm = gamma is not None
    if has_norm and (mean is None or invstd is None or scale is None or shift is None):
        raise ValueError("conv_bn_act with a BatchNorm needs the (mean, invstd, scale, shift) of otvae::bn_batch_stats")
    spec = HF.ConvSpec(stride, pad, up, relu, has_norm, bias is not None, residual is not None, False)
    res = HF.as_nhwc(residual) if residual is not None else None
    (y,), _, _ = HF.conv_forward_launch(x, (spec,), (mean, invstd, [scale], [shift], training), (wt, bias, gamma, beta, res))
    return y


def _conv_fwd_fake(x, weight, bias, gamma, beta, mean, invstd, scale, shift, residual, stride, pad, up, relu, training):
    kh, kw = weight.shape[2], weight.shape[3]
    ho, wo = (x.shape[2] * up + 2 * pad - kh) // stride + 1, (x.shape[3] * up + 2 * pad - kw) // stride + 1
    return _nhwc(x.shape[0], weight.shape[0], ho, wo, x)


def _conv_bwd(gy: Tensor, x: Tensor, weight: Tensor, gamma: Optional[Tensor], mean: Optional[Tensor], invstd: Optional[Tensor],
              scale: Optional[Tensor], shift: Optional[Tensor], stride: int, pad: int, up: int, relu: bool, has_bias: bool,
              has_residual: bool, training: bool, need_dx: bool):
    from . import functional as HF
    x = HF.as_nhwc(x)
    wt = weight if HF.is_hwio(weight) else HF.hwio_weight(weight)
    has_norm = gamma is not None
    spec = HF.ConvSpec(stride, pad, up, relu, has_norm, has_bias, has_residual, False)
    g, _, _ = HF._geom(x, wt, stride, pad, up)
    bias_like = x.new_empty(weight.shape[0]) if has_bias else None
    dx, ((gw, gb, dgam, dbet, _),) = HF.conv_backward_launch(x, (wt, bias_like, gamma, gamma, None), (spec,), [g],
                                                            (mean, invstd, [scale], [shift], training), ((None, None, None, None),),
                                                            (gy,), need_dx)
    e = x.new_empty(0)
    return (dx if dx is not None else e), gw, (gb if gb is not None else e), (dgam if dgam is not None else e), \
        (dbet if dbet is not None else e)


def _conv_bwd_fake(gy, x, weight, gamma, mean, invstd, scale, shift, stride, pad, up, relu, has_bias, has_residual, training,
                   need_dx):
    e = x.new_empty(0)
    return (torch.empty_strided(x.shape, x.stride(), device=x.device, dtype=x.dtype) if need_dx else e, torch.empty_like(weight),
            x.new_empty(weight.shape[0]) if has_bias else e, x.new_empty(x.shape[1]) if gamma is not None else e,
            x.new_empty(x.shape[1]) if gamma is not None else e)


_define("conv_bn_act",
        "(Tensor x, Tensor weight, Tensor? bias, Tensor? gamma, Tensor? beta, Tensor? mean, Tensor? invstd, Tensor? scale, "
        "Tensor? shift, Tensor? residual, int stride, int pad, int up, bool relu, bool training) -> Tensor", _conv_fwd, _conv_fwd_fake)
_define("conv_bn_act_backward",
        "(Tensor gy, Tensor x, Tensor weight, Tensor? gamma, Tensor? mean, Tensor? invstd, Tensor? scale, Tensor? shift, int stride, "
        "int pad, int up, bool relu, bool has_bias, bool has_residual, bool training, bool need_dx) -> "
        "(Tensor, Tensor, Tensor, Tensor, Tensor)", _conv_bwd, _conv_bwd_fake)


def _conv_setup(ctx, inputs, output):
    x, weight, bias, gamma, beta, mean, invstd, scale, shift, residual, stride, pad, up, relu, training = inputs
    ctx.save_for_backward(x, weight, gamma, mean, invstd, scale, shift)
    ctx.cfg = (stride, pad, up, relu, bias is not None, residual is not None, training)


def _conv_backward(ctx, gy):
    x, weight, gamma, mean, invstd, scale, shift = ctx.saved_tensors
    stride, pad, up, relu, has_bias, has_res, training = ctx.cfg
    need_dx = ctx.needs_input_grad[0]
    dx, gw, gb, dgam, dbet = torch.ops.otvae.conv_bn_act_backward(gy, x, weight, gamma, mean, invstd, scale, shift, stride, pad, up,
                                                                  relu, has_bias, has_res, training, need_dx)
    none = lambda t: t if t.numel() else None  # noqa: E731
    return (none(dx) if need_dx else None, gw, none(gb), none(dgam), none(dbet), None, None, None, None, gy if has_res else None,
            None, None, None, None, None)


torch.library.register_autograd("otvae::conv_bn_act", _conv_backward, setup_context=_conv_setup)


# ------------------------------------------------------------------------------------------------ GaussianPrior
def _gp_fwd(h: Tensor, eps: Tensor, coeff: float):
    lib = _lib.load()
    b, c2, hh, ww = h.shape
    d, s = c2 // 2, hh * ww
    z = _nhwc(b, d, hh, ww, h)
    loss = torch.empty(b, device=h.device, dtype=torch.float32)
    check(lib.otvae_gaussian_prior_fwd(ptr(h), ptr(eps), b, s, d, float(coeff), ptr(z), ptr(loss), stream()),
          "otvae_gaussian_prior_fwd")
    return z, loss


def _gp_bwd(h: Tensor, eps: Tensor, gz: Optional[Tensor], gloss: Optional[Tensor], coeff: float):
    from .functional import as_nhwc
    lib = _lib.load()
    b, c2, hh, ww = h.shape
    d, s = c2 // 2, hh * ww
    gz = as_nhwc(gz) if gz is not None else None
    gloss = gloss.contiguous() if gloss is not None else None
    gh = torch.empty_strided(h.shape, h.stride(), device=h.device, dtype=h.dtype)
    check(lib.otvae_gaussian_prior_bwd(ptr(h), ptr(eps), ptr(gz), ptr(gloss), b, s, d, float(coeff), ptr(gh), stream()),
          "otvae_gaussian_prior_bwd")
    return gh


_define("gaussian_prior", "(Tensor h, Tensor eps, float coeff) -> (Tensor, Tensor)", _gp_fwd,
        lambda h, eps, coeff: (_nhwc(h.shape[0], h.shape[1] // 2, h.shape[2], h.shape[3], h), h.new_empty(h.shape[0])))
_define("gaussian_prior_backward", "(Tensor h, Tensor eps, Tensor? gz, Tensor? gloss, float coeff) -> Tensor", _gp_bwd,
        lambda h, eps, gz, gloss, coeff: torch.empty_strided(h.shape, h.stride(), device=h.device, dtype=h.dtype))


def _gp_setup(ctx, inputs, output):
    h, eps, coeff = inputs
    ctx.save_for_backward(h, eps)
    ctx.coeff = coeff
    ctx.set_materialize_grads(False)  # gz / gloss are optional arguments of the backward kernel


def _gp_backward(ctx, gz, gloss):
    if gz is None and gloss is None:
        return None, None, None
    h, eps = ctx.saved_tensors
    return torch.ops.otvae.gaussian_prior_backward(h, eps, gz, gloss, ctx.coeff), None, None


torch.library.register_autograd("otvae::gaussian_prior", _gp_backward, setup_context=_gp_setup)


# ------------------------------------------------------------------------------------------------ nelbo reduction
def _nelbo_fwd(pred: Tensor, target: Tensor, prior_loss: Optional[Tensor], chw: float):
    from .functional import PriorLane
    if PriorLane.is_open(pred.device) and PriorLane.active(pred.device):
        # the prior term is still being computed on the prior lane (functional.PriorLane): the loss vector is formed there too,
        # behind it -- the backward pass needs pred and target, not this value
        PriorLane.hold(pred.device, pred, target, prior_loss)
        with PriorLane.section(pred.device):
            return _nelbo_fwd_launch(pred, target, prior_loss, chw, allow_defer=False)  # (ordered behind the lane's work, not the side stream's)
    return _nelbo_fwd_launch(pred, target, prior_loss, chw)


def _nelbo_fwd_launch(pred: Tensor, target: Tensor, prior_loss: Optional[Tensor], chw: float, allow_defer: bool = True):
    from .functional import _PendingReduce
    lib = _lib.load()
    numel = pred.numel()
    # the prior term is a mean over ITS entries: one per latent the prior saw = expansion * batch (model/vae.py:165-169), which is
    # the batch of `pred` (the mean over the replicas) only when expansion = 1
    n_prior = prior_loss.numel() if prior_loss is not None else pred.shape[0]
    if prior_loss is not None:
        prior_loss = prior_loss.contiguous()
    ws = torch.empty(lib.otvae_nelbo_ws(), device=pred.device, dtype=torch.float64)
    out = torch.empty(3, device=pred.device, dtype=torch.float32)

    def launch():
        check(lib.otvae_nelbo_fwd(ptr(pred), ptr(target), numel, ptr(prior_loss), n_prior, float(chw), ptr(ws), ptr(out), stream()),
              "otvae_nelbo_fwd")

    # inside a training engine's captured step the loss VALUE goes to the side stream with the backward pass's first weight-gradient
    # fork: the backward pass needs pred and target, not this value (functional._PendingReduce.defer_to_side)
    if not (allow_defer and _PendingReduce.defer_to_side(pred.device, launch, pred, target, prior_loss, ws, out)):
        launch()
    return out


def _nelbo_bwd(gout: Tensor, pred: Tensor, target: Tensor, n_prior: int, chw: float):
    lib = _lib.load()
    numel = pred.numel()
    gout = gout.contiguous()
    gpred = torch.empty_strided(pred.shape, pred.stride(), device=pred.device, dtype=pred.dtype)
    gprior = torch.empty(n_prior, device=pred.device, dtype=torch.float32)
    check(lib.otvae_nelbo_bwd(ptr(pred), ptr(target), numel, max(n_prior, 1), float(chw), ptr(gout), ptr(gpred),
                              ptr(gprior) if n_prior else None, stream()), "otvae_nelbo_bwd")
    return gpred, gprior


_define("nelbo_loss", "(Tensor pred, Tensor target, Tensor? prior_loss, float chw) -> Tensor", _nelbo_fwd,
        lambda pred, target, prior_loss, chw: pred.new_empty(3))
_define("nelbo_loss_backward", "(Tensor gout, Tensor pred, Tensor target, int n_prior, float chw) -> (Tensor, Tensor)", _nelbo_bwd,
        lambda gout, pred, target, n_prior, chw: (torch.empty_strided(pred.shape, pred.stride(), device=pred.device, dtype=pred.dtype),
                                                  pred.new_empty(n_prior)))


def _nelbo_setup(ctx, inputs, output):
    pred, target, prior_loss, chw = inputs
    ctx.save_for_backward(pred, target)
    ctx.cfg = (0 if prior_loss is None else prior_loss.numel(), tuple(prior_loss.shape) if prior_loss is not None else None, chw)


def _nelbo_backward(ctx, gout):
    pred, target = ctx.saved_tensors
    n_prior, pshape, chw = ctx.cfg
    gpred, gprior = torch.ops.otvae.nelbo_loss_backward(gout, pred, target, n_prior, chw)
    return gpred, None, (gprior.reshape(pshape) if n_prior else None), None


torch.library.register_autograd("otvae::nelbo_loss", _nelbo_backward, setup_context=_nelbo_setup)


# ------------------------------------------------------------------------------------------------ minibatch-OT prior
def _sk_fwd(z: Tensor, y: Tensor, reg: float, max_iter: int, threshold: float, scale: float):
    lib = _lib.load()
    _lib.require_cuda(z, "latents")
    if z.dtype not in (torch.float32, torch.float64):
        raise TypeError("SinkhornPrior computes in float32 or float64")
    z, y = z.contiguous(), y.to(z.dtype).contiguous()
    (n, d), m = z.shape, y.shape[0]
    if y.shape[1] != d:
        raise ValueError(f"prior samples have {y.shape[1]} dimensions, latents {d}")
    dt = 0 if z.dtype == torch.float32 else 1
    new = lambda *shape: torch.empty(shape, device=z.device, dtype=z.dtype)  # noqa: E731
    C, pi, u, v, cost, cmax = new(n, m), new(n, m), new(n), new(m), new(n), new(1)   # cost: one entry per sample
    ws = torch.empty(lib.otvae_sinkhorn_prior_ws(dt, n, m), device=z.device, dtype=torch.uint8)
    iters = torch.empty(1, device=z.device, dtype=torch.int32)
    check(lib.otvae_sinkhorn_prior_fwd(dt, ptr(z), ptr(y), n, m, d, float(reg), int(max_iter), float(threshold), float(scale), n,
                                       ptr(ws), ptr(C), ptr(pi), ptr(u), ptr(v), ptr(cost), ptr(cmax), ptr(iters), stream()),
          "otvae_sinkhorn_prior_fwd")
    return cost, pi, iters


def _sk_bwd(g: Tensor, gadd: Optional[Tensor], z: Tensor, y: Tensor, pi: Tensor, scale: float):
    z, y = z.contiguous(), y.to(z.dtype).contiguous()
    n, d = z.shape
    gz = torch.empty_like(z)
    gadd = gadd.contiguous() if gadd is not None else None
    check(_lib.load().otvae_ot_cost_grad(0 if z.dtype == torch.float32 else 1, ptr(z), ptr(y), ptr(pi), ptr(g.contiguous()), g.numel(),
                                         float(scale), ptr(gadd), n, y.shape[0], d, ptr(gz), stream()), "otvae_ot_cost_grad")
    return gz


_define("sinkhorn_prior", "(Tensor z, Tensor y, float reg, int max_iter, float threshold, float scale) -> (Tensor, Tensor, Tensor)",
        _sk_fwd, lambda z, y, reg, max_iter, threshold, scale: (z.new_empty(z.shape[0]), z.new_empty(z.shape[0], y.shape[0]),
                                                               z.new_empty(1, dtype=torch.int32)))
_define("sinkhorn_prior_backward", "(Tensor g, Tensor? gadd, Tensor z, Tensor y, Tensor pi, float scale) -> Tensor", _sk_bwd,
        lambda g, gadd, z, y, pi, scale: torch.empty_like(z))


def _sk_setup(ctx, inputs, output):
    z, y, _, _, _, scale = inputs
    ctx.save_for_backward(z, y, output[1])
    ctx.scale = scale
    ctx.set_materialize_grads(False)


def _sk_backward(ctx, g, _gpi, _giters):
    z, y, pi = ctx.saved_tensors
    if g is None:
        return None, None, None, None, None, None
    return torch.ops.otvae.sinkhorn_prior_backward(g, None, z, y, pi, ctx.scale), None, None, None, None, None


torch.library.register_autograd("otvae::sinkhorn_prior", _sk_backward, setup_context=_sk_setup)


# ------------------------------------------------------------------------------------------------ Gaussian W2 prior
def _w2_fwd(z: Tensor, mut: Optional[Tensor], covt: Optional[Tensor], rt: Optional[Tensor], v_init: Optional[Tensor],
            warm: Optional[Tensor], scale: float):
    from .ot import matrix_utils as MU
    lib = _lib.load()
    _lib.require_cuda(z, "latents")
    if z.dtype not in (torch.float32, torch.float64):
        raise TypeError("GaussianW2Prior takes float32 or float64 latents")
    z = z.contiguous()
    b, d = z.shape
    dev = z.device
    f64 = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float64)  # noqa: E731
    n, sx, sxx = f64(1), f64(1, d), f64(1, d, d)
    ws = torch.empty(max(8, lib.otvae_gauss_stats_ws(1, b, d, 0)), device=dev, dtype=torch.uint8)
    check(lib.otvae_gauss_stats(0 if z.dtype == torch.float32 else 1, ptr(z), 1, b, d, 0, 0, -1.0, ptr(ws), ptr(n), ptr(sx), ptr(sxx),
                                stream()), "otvae_gauss_stats")
    mu, cov = f64(1, d), f64(1, d, d)
    check(lib.otvae_mean_cov(ptr(n), ptr(sx), ptr(sxx), 1, d, 0, ptr(mu), ptr(cov), stream()), "otvae_mean_cov")
    m = cov if rt is None else MU.matmul64(MU.matmul64(rt, cov), rt)
    lam, vt = f64(1, d), f64(1, d, d)
    ews = torch.empty(lib.otvae_eigh_ws(1, d), device=dev, dtype=torch.uint8)
    if v_init is not None:  # the previous step's eigenvectors as the start basis (2-4 sweeps instead of ~9)
        g0 = f64(1, d, d)
        check(lib.otvae_eigh_fn_warm(ptr(m), ptr(v_init), ptr(warm), 1, d, 3, ptr(vt), ptr(lam), ptr(ews), ptr(g0), stream()),
              "otvae_eigh_fn_warm")
    else:
        check(lib.otvae_eigh_fn(ptr(m), 1, d, 3, ptr(vt), ptr(lam), ptr(ews), stream()), "otvae_eigh_fn")
    loss = torch.empty(b, device=dev, dtype=torch.float32)
    q = f64(d, d)
    check(lib.otvae_w2_prior_tail(ptr(mu), ptr(mut), ptr(cov), ptr(covt), ptr(lam), ptr(vt), d, float(scale), b, ptr(loss), ptr(q),
                                  stream()), "otvae_w2_prior_tail")
    return loss, mu, q, vt


def _w2_bwd(g: Tensor, gadd: Optional[Tensor], z: Tensor, mu: Tensor, q: Tensor, mut: Optional[Tensor], rt: Optional[Tensor],
            scale: float):
    from .ot import matrix_utils as MU
    lib = _lib.load()
    z = z.contiguous()
    b, d = z.shape
    w = MU.matmul64(q, q, trans_a=True)                          # M^-1/2
    if rt is not None:
        w = MU.matmul64(MU.matmul64(rt, w), rt)                  # covt^1/2 M^-1/2 covt^1/2
    gz = torch.empty_like(z)
    gadd = gadd.contiguous() if gadd is not None else None
    check(lib.otvae_w2_prior_bwd(0 if z.dtype == torch.float32 else 1, ptr(z), b, d, ptr(mu), ptr(mut), ptr(w),
                                 ptr(g.float().contiguous()), g.numel(), float(scale), ptr(gadd), ptr(gz), stream()),
          "otvae_w2_prior_bwd")
    return gz


_define("gaussian_w2_prior", "(Tensor z, Tensor? target_mean, Tensor? target_cov, Tensor? target_root, Tensor? v_init, Tensor? warm, "
        "float scale) -> (Tensor, Tensor, Tensor, Tensor)", _w2_fwd,
        lambda z, mut, covt, rt, v_init, warm, scale: (z.new_empty(z.shape[0], dtype=torch.float32),
                                                       z.new_empty((1, z.shape[1]), dtype=torch.float64),
                                                       z.new_empty((z.shape[1], z.shape[1]), dtype=torch.float64),
                                                       z.new_empty((1, z.shape[1], z.shape[1]), dtype=torch.float64)))
_define("gaussian_w2_prior_backward", "(Tensor g, Tensor? gadd, Tensor z, Tensor mu, Tensor q, Tensor? target_mean, "
        "Tensor? target_root, float scale) -> Tensor", _w2_bwd, lambda g, gadd, z, mu, q, mut, rt, scale: torch.empty_like(z))


def _w2_setup(ctx, inputs, output):
    z, mut, _, rt, _, _, scale = inputs
    ctx.save_for_backward(z, output[1], output[2], mut, rt)
    ctx.scale = scale
    ctx.set_materialize_grads(False)


def _w2_backward(ctx, g, _gmu, _gq, _gvt):
    z, mu, q, mut, rt = ctx.saved_tensors
    if g is None:
        return None, None, None, None, None, None, None
    return torch.ops.otvae.gaussian_w2_prior_backward(g, None, z, mu, q, mut, rt, ctx.scale), None, None, None, None, None, None


torch.library.register_autograd("otvae::gaussian_w2_prior", _w2_backward, setup_context=_w2_setup)
