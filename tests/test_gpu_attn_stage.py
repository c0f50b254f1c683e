"""GPU parity of the one-launch AttentionBlock forward and backward (``otvae_attn_stage_fwd`` / ``_bwd``, reference networks/cnn.py:212-240) -- run with
``-m gpu``: against the CPU oracle's ``attention_block`` (values, BatchNorm running buffers, every gradient), and against the same
block issued as its three launches (``OTVAE_ATTN_STAGE=0``'s route), including the statistics partials the next BatchNorm picks up."""
import ctypes as C

import pytest
import torch

import otvae_oracle as O
from conftest import rel_err
from detfill import normal

pytestmark = pytest.mark.gpu

# (batch, width, side, heads): the fused stages of the MNIST / CIFAR autoencoders + ragged batches (a last workgroup with fewer images)
SHAPES = [(8, 8, 16, 4), (6, 16, 8, 4), (9, 32, 4, 8), (13, 32, 2, 8), (3, 1, 32, 1), (5, 4, 4, 2), (4, 16, 16, 4), (7, 32, 4, 4), (5, 2, 8, 2)]
TOL = 1e-4


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available()
    import ot_vae_lightning_amd as pkg
    return pkg


def _block(A, width, heads, seed, norm="batchnorm"):
    from ot_vae_lightning_amd.networks.cnn import AttentionBlock
    blk = AttentionBlock(width, heads=heads, normalization=norm).cuda()
    with torch.no_grad():
        for i, p in enumerate(blk.parameters()):
            p.copy_(normal(tuple(p.shape), seed + i).mul_(0.4 if p.dim() > 1 else 0.2))
        if norm is not None:
            blk.qkv._normalization.weight.add_(1.0)
    return blk


def _run(blk, x, res, gy, fused, fused_bwd=True):
    from ot_vae_lightning_amd import functional as HF
    old = HF.ATTN_STAGE, HF.ATTN_STAGE_BWD
    HF.ATTN_STAGE, HF.ATTN_STAGE_BWD = fused, fused_bwd
    try:
        for p in blk.parameters():
            p.grad = None
        x = x.clone().requires_grad_(True)
        r = res.clone().requires_grad_(True) if res is not None else None
        y = blk(x, residual=r)
        st = getattr(y, "_otvae_stats", None)
        y.backward(gy)
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in blk.named_parameters()}
        return y.detach(), st, x.grad, (r.grad if r is not None else None), grads
    finally:
        HF.ATTN_STAGE, HF.ATTN_STAGE_BWD = old


@pytest.mark.parametrize("n,width,side,heads", SHAPES)
@pytest.mark.parametrize("with_res", [False, True])
def test_stage_vs_oracle_and_three_launches(A, n, width, side, heads, with_res):
    from ot_vae_lightning_amd import functional as HF
    lib = A._lib.load()
    rows = C.c_int(0)
    assert lib.otvae_attn_stage_plan(n, side * side, heads, width // heads, 1, C.byref(rows)) == 0, "shape expected to fuse"
    blk = _block(A, width, heads, 11)
    x = HF.as_nhwc(normal((n, width, side, side), 3).cuda())
    res = HF.as_nhwc(normal((n, width, side, side), 5).cuda()) if with_res else None
    gy = HF.as_nhwc(normal((n, width, side, side), 7).cuda())
    sd0 = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    y1, st1, dx1, dr1, g1 = _run(blk, x, res, gy, True)
    sd1 = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd0)
    y0, st0, dx0, dr0, g0 = _run(blk, x, res, gy, False)
    sd2 = blk.state_dict()
    # -- fused vs three launches
    assert rel_err(y1, y0) < TOL and rel_err(dx1, dx0) < TOL
    for k in g0:
        assert rel_err(g1[k], g0[k]) < TOL, k
    # -- the fused forward with the three-launch backward (what shapes take whose backward kernel does not fit)
    blk.load_state_dict(sd0)
    y2, _, dx2, _, g2 = _run(blk, x, res, gy, True, fused_bwd=False)
    assert torch.equal(y2, y1) and rel_err(dx2, dx0) < TOL
    for k in g0:
        assert rel_err(g2[k], g0[k]) < TOL, k
    blk.load_state_dict(sd0)
    _run(blk, x, res, gy, False)
    if with_res:
        assert torch.equal(dr1, dr0)
    for k in sd1:
        assert rel_err(sd1[k].double(), sd2[k].double()) < 1e-6, k
    # the statistics the next BatchNorm would read: per-channel sum and sum of squares of y
    assert st1 is not None and st0 is not None
    part, p_, ld = st1
    assert (p_, ld) == (rows.value, width)
    sums = part.reshape(-1).view(2, width, p_).sum(-1)  # memory order [2][width][rows]
    yy = y1.double().permute(0, 2, 3, 1).reshape(-1, width)
    assert rel_err(sums[0], yy.sum(0)) < 1e-6 and rel_err(sums[1], (yy * yy).sum(0)) < 1e-6
    # -- vs the oracle (fp32 torch on the CPU)
    p = {"b." + k: v.detach().cpu().clone().contiguous() for k, v in sd0.items()}
    leaves = {k: v.requires_grad_(True) for k, v in p.items() if v.is_floating_point() and "running" not in k}
    xc = x.detach().cpu().contiguous().requires_grad_(True)
    yo = O.attention_block(xc, p, "b.", heads, training=True)
    if with_res:
        yo = yo + res.cpu()
    yo.backward(gy.cpu())
    assert rel_err(y1.cpu(), yo.detach()) < TOL
    assert rel_err(dx1.cpu(), xc.grad) < TOL
    for k, v in leaves.items():
        # (the BatchNorm parameter gradients are sums with heavy cancellation: dgamma = 0.008 out of terms of order 1 at width 1,
        #  where the fp32 oracle itself is good to ~1e-3; the backward launches are the three-launch route's, compared above)
        assert rel_err(g1[k[2:]].cpu(), v.grad) < (2e-3 if "_normalization" in k else TOL), k
    for k in ("qkv._normalization.running_mean", "qkv._normalization.running_var"):
        assert rel_err(sd1[k].cpu(), p["b." + k]) < 1e-5, k


def test_stage_eval_mode_and_no_norm(A):
    from ot_vae_lightning_amd import functional as HF
    for norm in ("batchnorm", None):
        blk = _block(A, 16, 4, 23, norm=norm)
        if norm is not None:
            with torch.no_grad():
                blk.qkv._normalization.running_mean.copy_(normal((16,), 1).mul_(0.1))
                blk.qkv._normalization.running_var.copy_(normal((16,), 2).abs_().add_(0.5))
        blk.eval()
        x = HF.as_nhwc(normal((5, 16, 8, 8), 3).cuda())
        gy = HF.as_nhwc(normal((5, 16, 8, 8), 7).cuda())
        y1, st1, dx1, _, g1 = _run(blk, x, None, gy, True)
        y0, st0, dx0, _, g0 = _run(blk, x, None, gy, False)
        y2, _, dx2, _, g2 = _run(blk, x, None, gy, True, fused_bwd=False)
        assert torch.equal(y2, y1) and rel_err(dx2, dx0) < TOL
        assert st1 is None and st0 is None
        assert rel_err(y1, y0) < TOL and rel_err(dx1, dx0) < TOL
        for k in g0:
            assert rel_err(g1[k], g0[k]) < TOL, (norm, k)
        with torch.no_grad():
            assert rel_err(blk(x), y0) < TOL


def test_stage_plan_rejects_what_it_cannot_take(A):
    lib = A._lib.load()
    rows = C.c_int(0)
    for n, t, h, c in [(4, 1, 16, 16), (4, 9, 2, 4), (4, 16, 3, 4), (4, 16, 8, 16), (4, 4, 8, 8), (4, 1024, 4, 2)]:
        assert lib.otvae_attn_stage_plan(n, t, h, c, 1, C.byref(rows)) != 0, (n, t, h, c)
    # ... and such blocks still run (three launches)
    from ot_vae_lightning_amd import functional as HF
    blk = _block(A, 12, 3, 5)
    x = HF.as_nhwc(normal((3, 12, 4, 4), 3).cuda())
    y = blk(x)
    assert y.shape == x.shape and torch.isfinite(y).all()


def test_stage_bwd_with_saved_qkv_equals_recomputed(A):
    """``otvae_attn_stage_bwd`` through the C ABI: given the qkv tensor the forward kernel can write it reads q / k / v from there; given
    NULL it forms them again from x -- the same arithmetic, so the same bits."""
    from ot_vae_lightning_amd import functional as HF
    from ot_vae_lightning_amd._lib import ptr, stream, check
    lib = A._lib.load()
    n, hc, side, heads = 6, 16, 8, 4
    t, c = side * side, hc // heads
    x = HF.as_nhwc(normal((n, hc, side, side), 3).cuda())
    gy = HF.as_nhwc(normal((n, hc, side, side), 4).cuda())
    wq = normal((hc, 3 * hc), 5).mul_(0.3).cuda()
    wp = normal((hc, hc), 6).mul_(0.3).cuda()
    scale, shift = normal((hc,), 7).mul_(0.2).add_(1.0).cuda(), normal((hc,), 8).mul_(0.1).cuda()
    mean, invstd = normal((hc,), 9).mul_(0.1).cuda(), normal((hc,), 10).abs_().add_(0.5).cuda()
    rows, rows_b = C.c_int(0), C.c_int(0)
    assert lib.otvae_attn_stage_plan(n, t, heads, c, 1, C.byref(rows)) == 0
    assert lib.otvae_attn_stage_bwd_plan(n, t, heads, c, C.byref(rows_b)) == 0
    e = lambda *shape, dt=torch.float32: torch.empty(shape, device="cuda", dtype=dt)  # noqa: E731
    qkv, out, y, lse = e(n, t, 3 * hc), e(n, t, hc), e(n, t, hc), e(n, heads, t)
    check(lib.otvae_attn_stage_fwd(ptr(x), ptr(scale), ptr(shift), ptr(wq), ptr(wp), None, n, t, heads, c, 1.0 / c, ptr(qkv), ptr(out),
                                   ptr(lse), None, ptr(y), None, stream()), "fwd")
    res = []
    for saved in (qkv, None):
        gqkv, gv, part = e(n, t, 3 * hc), e(n, t, hc), e(rows_b.value, 2, hc, dt=torch.float64)
        check(lib.otvae_attn_stage_bwd(ptr(gy), ptr(wp), ptr(wq), ptr(x), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), ptr(saved),
                                       ptr(out), ptr(lse), None, n, t, heads, c, 1.0 / c, ptr(gqkv), ptr(gv), ptr(part), stream()), "bwd")
        torch.cuda.synchronize()
        res.append((gqkv, gv, part))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # and the forward kernel without the qkv output writes the same out / y / lse
    out2, y2, lse2 = e(n, t, hc), e(n, t, hc), e(n, heads, t)
    check(lib.otvae_attn_stage_fwd(ptr(x), ptr(scale), ptr(shift), ptr(wq), ptr(wp), None, n, t, heads, c, 1.0 / c, None, ptr(out2),
                                   ptr(lse2), None, ptr(y2), None, stream()), "fwd")
    torch.cuda.synchronize()
    assert torch.equal(out2, out) and torch.equal(y2, y) and torch.equal(lse2, lse)
