"""GPU tests of the convolution entry points of the C ABI, called directly through ctypes (no autograd, no modules):

  * every kernel family that can serve a layer (image-tile MFMA fwd/dgrad/wgrad, implicit-GEMM MFMA, direct VALU) against a plain torch
    float64 restatement of ConvLayer.forward / its backward (reference networks/cnn.py:183-192) on the same inputs;
  * image-tile vs implicit-GEMM outputs of the same call, which must agree BIT FOR BIT (same k order, same fp32 MFMA);
  * otvae_conv_multi vs one call per job (bit for bit, including the BatchNorm partial sums);
  * ragged sizes (batch not a multiple of the images-per-block, one image, Cn not a multiple of 16) and error returns.

Tolerance for the fp64 comparison: fp32 accumulation over K = KH*KW*Cs <= 1152 terms of O(1) products -> 2e-5 relative
to the tensor's max magnitude.
"""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5


@pytest.fixture(scope="module")
def lib():
    from ot_vae_lightning_amd import _lib
    assert torch.cuda.is_available()
    return _lib.load()


def _L():
    from ot_vae_lightning_amd import _lib
    return _lib


def nhwc(t):  # logical NCHW tensor in NHWC memory
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def raw(t):  # the NHWC memory as a plain contiguous [N, H, W, C] tensor
    return t.permute(0, 2, 3, 1)


def make_case(n, cs, cn, hs, k, stride, pad, up, norm=True, relu=True, bias=True, res=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cs, hs, hs, generator=g)
    w = torch.randn(cn, cs, k, k, generator=g) * (1.0 / (k * k * cs) ** 0.5)
    scale = torch.rand(cs, generator=g) + 0.5 if norm else None
    shift = torch.randn(cs, generator=g) * 0.3 if norm else None
    b = torch.randn(cn, generator=g) if bias else None
    ho = (hs * up + 2 * pad - k) // stride + 1
    r = torch.randn(n, cn, ho, ho, generator=g) if res else None
    gy = torch.randn(n, cn, ho, ho, generator=g)
    return dict(x=x, w=w, scale=scale, shift=shift, bias=b, res=r, gy=gy, stride=stride, pad=pad, up=up, relu=relu, ho=ho)


def ref64(c):
    """float64 restatement: y, gv (= d loss / d (x*scale+shift)), gw, gb for loss = sum(y * gy)."""
    x = c["x"].double()
    a = x
    if c["scale"] is not None:
        a = a * c["scale"].double()[None, :, None, None] + c["shift"].double()[None, :, None, None]
    a = a.detach().requires_grad_(True)  # the BatchNorm output: gv is the gradient HERE
    h = torch.relu(a) if c["relu"] else a
    if c["up"] == 2:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    w = c["w"].double().requires_grad_(True)
    b = c["bias"].double().requires_grad_(True) if c["bias"] is not None else None
    y = F.conv2d(h, w, b, stride=c["stride"], padding=c["pad"])
    if c["res"] is not None:
        y = y + c["res"].double()
    (y * c["gy"].double()).sum().backward()
    return y.detach(), a.grad, w.grad, (b.grad if b is not None else None)


class Dev:
    """Device copies of one case + the geometry struct."""

    def __init__(self, c):
        L = _L()
        self.c = c
        d = "cuda"
        self.x = nhwc(c["x"].to(d))
        self.gy = nhwc(c["gy"].to(d))
        self.res = nhwc(c["res"].to(d)) if c["res"] is not None else None
        w = c["w"].to(d)
        self.w_hwio = w.permute(2, 3, 1, 0).contiguous()  # [KH][KW][Cs][Cn]
        self.scale = c["scale"].to(d) if c["scale"] is not None else None
        self.shift = c["shift"].to(d) if c["shift"] is not None else None
        self.bias = c["bias"].to(d) if c["bias"] is not None else None
        n, cs, hs, _ = c["x"].shape
        cn, _, k, _ = c["w"].shape
        self.geom = L.ConvGeom(n, hs, hs, cs, c["up"], c["ho"], c["ho"], cn, k, k, c["stride"], c["pad"])
        self.n, self.cs, self.cn, self.hs, self.k = n, cs, cn, hs, k
        self.wd = torch.empty(k * k * cn * cs, device=d)
        assert load().otvae_weight_transpose(L.ptr(self.w_hwio), L.ptr(self.wd), k * k, cs, cn, L.stream()) == 0


def load():
    return _L().load()


def run_fwd(dv, stats=True):
    L, lib = _L(), load()
    y = nhwc(torch.empty(dv.n, dv.cn, dv.c["ho"], dv.c["ho"], device="cuda"))
    part = None
    p, ld = C.c_int(0), C.c_int(0)
    if stats:
        L.check(lib.otvae_conv_fwd_stats_ws(C.byref(dv.geom), C.byref(p), C.byref(ld)), "ws")
        part = torch.full((2, ld.value, p.value), float("nan"), device="cuda", dtype=torch.float64)
    L.check(lib.otvae_conv_fwd(C.byref(dv.geom), L.ptr(dv.x), L.ptr(dv.scale), L.ptr(dv.shift), int(dv.c["relu"]),
                               L.ptr(dv.w_hwio), L.ptr(dv.bias), L.ptr(dv.res), L.ptr(y), L.ptr(part), L.stream()), "fwd")
    torch.cuda.synchronize()
    sums = part[:, :dv.cn, :].sum(-1) if stats else None  # [2][Cn]
    return y, sums


def run_dgrad(dv, sums=True):
    L, lib = _L(), load()
    gv = nhwc(torch.empty(dv.n, dv.cs, dv.hs, dv.hs, device="cuda"))
    p, cp = C.c_int(0), C.c_int(0)
    L.check(lib.otvae_conv_bwd_data_ws(C.byref(dv.geom), C.byref(p), C.byref(cp)), "ws")
    mean = torch.linspace(-0.2, 0.2, dv.cs, device="cuda") if sums else None
    invstd = torch.linspace(0.8, 1.2, dv.cs, device="cuda") if sums else None
    part = torch.full((2, cp.value, p.value), float("nan"), device="cuda", dtype=torch.float64) if sums else None
    L.check(lib.otvae_conv_bwd_data(C.byref(dv.geom), L.ptr(dv.gy), L.ptr(dv.wd), L.ptr(dv.x), L.ptr(dv.scale), L.ptr(dv.shift),
                                    int(dv.c["relu"]), L.ptr(mean), L.ptr(invstd), L.ptr(gv), L.ptr(part), L.stream()), "dgrad")
    torch.cuda.synchronize()
    return gv, (part[:, :dv.cs, :].sum(-1) if sums else None), mean, invstd


def run_wgrad(dv):
    L, lib = _L(), load()
    p = C.c_int(0)
    has_bias = dv.bias is not None
    L.check(lib.otvae_conv_bwd_weight_ws(C.byref(dv.geom), int(has_bias), C.byref(p)), "ws")
    kk = dv.k * dv.k * dv.cs + (1 if has_bias else 0)
    part = torch.empty((p.value, kk, dv.cn), device="cuda")
    gw = torch.empty_like(dv.w_hwio)
    gb = torch.empty(dv.cn, device="cuda") if has_bias else None
    L.check(lib.otvae_conv_bwd_weight(C.byref(dv.geom), L.ptr(dv.x), L.ptr(dv.scale), L.ptr(dv.shift), int(dv.c["relu"]),
                                      L.ptr(dv.gy), int(has_bias), L.ptr(part), L.ptr(gw), L.ptr(gb), 0, L.stream()), "wgrad")
    torch.cuda.synchronize()
    return gw, gb


def rel(got, want):
    want = want.to(got.device)
    return float((got.double() - want.double()).abs().max() / want.double().abs().max().clamp_min(1e-30))


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# (n, cs, cn, hs, k, stride, pad, up): the distinct layer kinds of the CNN builder, ragged batches included
CASES = [
    (5, 8, 8, 16, 3, 1, 1, 1),      # image-tile, one image per block
    (7, 8, 24, 16, 1, 1, 0, 1),     # qkv 1x1, Cn not a multiple of 16
    (6, 8, 16, 16, 4, 2, 1, 1),     # stride 2 (4 parity classes in the data gradient)
    (9, 16, 8, 8, 3, 1, 1, 2),      # nearest x2 up-sampling
    (3, 16, 16, 8, 3, 1, 1, 1),     # 8x8 maps
    (6, 16, 32, 16, 4, 2, 1, 1),    # K = 257 rows x 32 columns: tile wgrad splits the columns over blockIdx.y
    (21, 32, 32, 4, 3, 1, 1, 1),    # 4x4 maps: implicit GEMM by default, image-tile with OTVAE_TILE_ALL
    (37, 64, 64, 2, 3, 1, 1, 1),    # 2x2
    (70, 128, 64, 1, 3, 1, 1, 2),   # decoder entry: 1x1 -> 2x2
    (33, 64, 128, 2, 4, 2, 1, 1),   # 2x2 -> 1x1, only 4 of 16 taps can touch the image
    (4, 1, 8, 32, 4, 2, 1, 1),      # image side: direct VALU kernels
    (4, 8, 1, 16, 3, 1, 1, 2),
    (2, 3, 8, 32, 4, 2, 1, 1),      # RGB
    (3, 3, 16, 32, 4, 2, 1, 1),     # RGB image side of the capacity-16 network: 3 -> 16 ...
    (3, 16, 3, 16, 3, 1, 1, 2),     # ... 16 -> 3 with up-sampling, 3x3 and
    (3, 16, 3, 16, 1, 1, 0, 2),     # 1x1 (skip), and the
    (3, 3, 9, 32, 1, 1, 0, 1),      # 3 -> 9 qkv layer
    (3, 3, 3, 32, 3, 1, 1, 1),      # 3 -> 3 (3x3 and
    (3, 3, 3, 32, 1, 1, 0, 1),      # 1x1)
    (1, 12, 20, 8, 3, 1, 1, 1),     # channel counts % 4 == 0 but not powers of two, a single image
    (4100, 1, 1, 32, 3, 1, 1, 1),   # > 2^22 positions: the direct kernels' integer-division fallback
    (4100, 4, 4, 32, 3, 1, 1, 1),   # > 2^22 rows in the implicit GEMM (row decode falls back to udiv)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "n%d_%dto%d_%dx%d_k%ds%dp%du%d" % (c[0], c[1], c[2], c[3], c[3], c[4], c[5], c[6], c[7]))
def test_conv_entry_points_vs_float64(lib, case):
    n, cs, cn, hs, k, s, p, up = case
    c = make_case(n, cs, cn, hs, k, s, p, up, seed=sum(case))
    y64, gv64, gw64, gb64 = ref64(c)
    dv = Dev(c)
    variants = [dict(OTVAE_NO_TILE=None, OTVAE_TILE_ALL=None, OTVAE_NO_WTILE=None, OTVAE_WTILE_ALL=None),
                dict(OTVAE_NO_TILE="1", OTVAE_TILE_ALL=None, OTVAE_NO_WTILE="1", OTVAE_WTILE_ALL=None),  # implicit GEMM only
                dict(OTVAE_NO_TILE=None, OTVAE_TILE_ALL="1", OTVAE_NO_WTILE=None, OTVAE_WTILE_ALL="1")]  # tile kernels wherever they can run
    outs = []
    for v in variants:
        with env(**v):
            y, ysum = run_fwd(dv)
            gv, gsum, mean, invstd = run_dgrad(dv)
            gw, gb = run_wgrad(dv)
        assert rel(raw(y), raw(y64)) < TOL
        assert rel(raw(gv), raw(gv64)) < TOL
        assert rel(gw, gw64.permute(2, 3, 1, 0)) < TOL
        if gb is not None:
            assert rel(gb, gb64) < TOL
        # BatchNorm statistics of the output / BatchNorm-backward sums: fp64 sums of the fp32 tensors the kernel wrote
        yd = raw(y).double().reshape(-1, cn)
        assert rel(ysum[0], yd.sum(0)) < 1e-10 and rel(ysum[1], (yd * yd).sum(0)) < 1e-10
        gd = raw(gv).double().reshape(-1, cs)
        xhat = ((raw(dv.x).reshape(-1, cs) - mean) * invstd).double()
        assert rel(gsum[0], gd.sum(0)) < 1e-10
        assert float((gsum[1] - (gd * xhat).sum(0)).abs().max()) < 1e-9 * float((gd.abs() * xhat.abs()).sum(0).max())
        outs.append((y, gv))
    # different kernel families, same fp32 arithmetic in the same order: identical bits
    for y, gv in outs[1:]:
        assert torch.equal(y, outs[0][0])
        assert torch.equal(gv, outs[0][1])


def test_conv_multi_equals_single_calls(lib):
    L = _L()
    cases = [make_case(6, 16, 16, 8, 3, 1, 1, 1, seed=1), make_case(6, 16, 16, 8, 1, 1, 0, 1, bias=False, res=False, seed=2),
             make_case(19, 64, 64, 2, 3, 1, 1, 1, seed=3), make_case(19, 64, 64, 2, 1, 1, 0, 1, bias=False, relu=False, seed=4)]
    for a, b in ((0, 1), (2, 3)):
        dvs = [Dev(cases[a]), Dev(cases[b])]
        dvs[1].x = dvs[0].x  # the two branches of a ConvBlock read the same input
        singles = []
        for dv in dvs:
            y, ysum = run_fwd(dv)
            gv, gsum, mean, invstd = run_dgrad(dv)
            gw, gb = run_wgrad(dv)
            singles.append((y, gv, gw, gb, ysum, gsum))
        # ---- forward: both branches in one call
        jobs = (L.ConvJob * 2)()
        ys, parts = [], []
        for i, dv in enumerate(dvs):
            p, ld = C.c_int(0), C.c_int(0)
            L.check(lib.otvae_conv_fwd_stats_ws(C.byref(dv.geom), C.byref(p), C.byref(ld)), "ws")
            part = torch.zeros((2, ld.value, p.value), device="cuda", dtype=torch.float64)
            y = nhwc(torch.empty(dv.n, dv.cn, dv.c["ho"], dv.c["ho"], device="cuda"))
            jb = jobs[i]
            jb.kind, jb.relu, jb.geom = L.JOB_FWD, int(dv.c["relu"]), dv.geom
            jb.x, jb.scale, jb.shift, jb.w = L.ptr(dv.x), L.ptr(dv.scale), L.ptr(dv.shift), L.ptr(dv.w_hwio)
            jb.bias, jb.residual, jb.y, jb.stat_partial = L.ptr(dv.bias), L.ptr(dv.res), L.ptr(y), L.ptr(part)
            ys.append(y)
            parts.append(part)
        L.check(lib.otvae_conv_multi(2, jobs, L.stream()), "multi fwd")
        torch.cuda.synchronize()
        for i, dv in enumerate(dvs):
            assert torch.equal(ys[i], singles[i][0])
            assert torch.equal(parts[i][:, :dv.cn].sum(-1), singles[i][4])
        # ---- backward: weight- and data-gradient of both branches in one call
        jobs = (L.ConvJob * 4)()
        outs = []
        keep = []
        for i, dv in enumerate(dvs):
            has_bias = dv.bias is not None
            pw = C.c_int(0)
            L.check(lib.otvae_conv_bwd_weight_ws(C.byref(dv.geom), int(has_bias), C.byref(pw)), "ws")
            kk = dv.k * dv.k * dv.cs + (1 if has_bias else 0)
            wpart = torch.empty((pw.value, kk, dv.cn), device="cuda")
            gw = torch.empty_like(dv.w_hwio)
            gb = torch.empty(dv.cn, device="cuda") if has_bias else None
            jb = jobs[2 * i]
            jb.kind, jb.relu, jb.has_bias, jb.defer_reduce, jb.geom = L.JOB_BWD_WEIGHT, int(dv.c["relu"]), int(has_bias), 0, dv.geom
            jb.x, jb.gy, jb.scale, jb.shift = L.ptr(dv.x), L.ptr(dv.gy), L.ptr(dv.scale), L.ptr(dv.shift)
            jb.wpartial, jb.gw, jb.gb = L.ptr(wpart), L.ptr(gw), L.ptr(gb)
            pd, cp = C.c_int(0), C.c_int(0)
            L.check(lib.otvae_conv_bwd_data_ws(C.byref(dv.geom), C.byref(pd), C.byref(cp)), "ws")
            gv = nhwc(torch.empty(dv.n, dv.cs, dv.hs, dv.hs, device="cuda"))
            mean = torch.linspace(-0.2, 0.2, dv.cs, device="cuda")
            invstd = torch.linspace(0.8, 1.2, dv.cs, device="cuda")
            part = torch.zeros((2, cp.value, pd.value), device="cuda", dtype=torch.float64)
            jb = jobs[2 * i + 1]
            jb.kind, jb.relu, jb.geom = L.JOB_BWD_DATA, int(dv.c["relu"]), dv.geom
            jb.gy, jb.w, jb.x, jb.scale, jb.shift = L.ptr(dv.gy), L.ptr(dv.wd), L.ptr(dv.x), L.ptr(dv.scale), L.ptr(dv.shift)
            jb.mean, jb.invstd, jb.gv, jb.bn_partial = L.ptr(mean), L.ptr(invstd), L.ptr(gv), L.ptr(part)
            outs.append((gv, gw, gb, part))
            keep += [wpart, mean, invstd]
        L.check(lib.otvae_conv_multi(4, jobs, L.stream()), "multi bwd")
        torch.cuda.synchronize()
        for i, dv in enumerate(dvs):
            gv, gw, gb, part = outs[i]
            assert torch.equal(gv, singles[i][1])
            assert torch.equal(gw, singles[i][2])
            if gb is not None:
                assert torch.equal(gb, singles[i][3])
            assert torch.equal(part[:, :dv.cs].sum(-1), singles[i][5])


@pytest.mark.parametrize("shape", [(64, 256, 256, 1, 3, 1, 1, 1), (64, 64, 128, 2, 4, 2, 1, 1), (64, 32, 64, 2, 3, 1, 1, 1)],
                         ids=["3x3_on_1x1", "4x4s2_2x2_to_1x1", "3x3_on_2x2_no_dead_tap"])
def test_deferred_sparse_reduce_skips_dead_taps(lib, shape):
    """Layers whose map is smaller than the kernel: the taps that never touch the image have a zero gradient.  With
    defer_reduce = OTVAE_DEFER_SPARSE the kernels may leave those partial rows unwritten (the workspace is pre-filled with
    NaN here) and otvae_wgrad_reduce_batched, given the mask of otvae_conv_dead_taps, must produce exactly the gradient of
    the immediate (dense) path."""
    L = _L()
    n, cs, cn, hs, k, s, p, up = shape
    dv = Dev(make_case(n, cs, cn, hs, k, s, p, up, seed=11))
    gw_ref, gb_ref = run_wgrad(dv)
    dead = C.c_uint32(0)
    L.check(lib.otvae_conv_dead_taps(C.byref(dv.geom), C.byref(dead)), "dead taps")
    ho = dv.c["ho"]
    expect = 0
    for kh in range(k):
        for kw in range(k):
            touch = lambda d: (d + (ho - 1) * s >= 0) and (d < hs * up)  # noqa: E731
            if not (touch(kh - p) and touch(kw - p)):
                expect |= 1 << (kh * k + kw)
    assert dead.value == expect
    has_bias = dv.bias is not None
    pw = C.c_int(0)
    L.check(lib.otvae_conv_bwd_weight_ws(C.byref(dv.geom), int(has_bias), C.byref(pw)), "ws")
    kk = k * k * cs + (1 if has_bias else 0)
    wpart = torch.full((pw.value, kk, cn), float("nan"), device="cuda")
    gw = torch.full_like(dv.w_hwio, float("nan"))
    gb = torch.full((cn,), float("nan"), device="cuda") if has_bias else None
    jobs = (L.ConvJob * 1)()
    jb = jobs[0]
    jb.kind, jb.relu, jb.has_bias, jb.defer_reduce, jb.geom = L.JOB_BWD_WEIGHT, int(dv.c["relu"]), int(has_bias), L.DEFER_SPARSE, dv.geom
    jb.x, jb.gy, jb.scale, jb.shift = L.ptr(dv.x), L.ptr(dv.gy), L.ptr(dv.scale), L.ptr(dv.shift)
    jb.wpartial, jb.gw, jb.gb = L.ptr(wpart), L.ptr(gw), L.ptr(gb)
    L.check(lib.otvae_conv_multi(1, jobs, L.stream()), "multi wgrad, deferred")
    one = lambda v: (C.c_int * 1)(v)  # noqa: E731
    L.check(lib.otvae_wgrad_reduce_batched(1, L.ptr_array([wpart]), one(pw.value), one(kk - int(has_bias)), one(kk), one(cn),
                                           L.ptr_array([gw]), L.ptr_array([gb]), one(cs), (C.c_uint32 * 1)(dead.value),
                                           L.stream()), "reduce")
    torch.cuda.synchronize()
    assert torch.equal(gw, gw_ref)
    if has_bias:
        assert torch.equal(gb, gb_ref)
    if dead.value:
        g4 = gw.reshape(k * k, cs, cn)
        for t in range(k * k):
            if (dead.value >> t) & 1:
                assert not g4[t].any()
        assert torch.isnan(wpart).any()      # the kernels really skipped the dead rows
    # the dense deferred form (every row written and read) must agree too
    wpart2 = torch.full((pw.value, kk, cn), float("nan"), device="cuda")
    gw2 = torch.empty_like(gw)
    jb.defer_reduce, jb.wpartial = L.DEFER_DENSE, L.ptr(wpart2)
    L.check(lib.otvae_conv_multi(1, jobs, L.stream()), "multi wgrad, deferred dense")
    L.check(lib.otvae_wgrad_reduce_batched(1, L.ptr_array([wpart2]), one(pw.value), one(kk - int(has_bias)), one(kk), one(cn),
                                           L.ptr_array([gw2]), L.ptr_array([gb]), None, None, L.stream()), "reduce dense")
    torch.cuda.synchronize()
    assert torch.equal(gw2, gw_ref) and not torch.isnan(wpart2).any()


def test_conv_multi_rejects_bad_jobs(lib):
    L = _L()
    jobs = (L.ConvJob * 1)()
    jobs[0].kind = 7
    assert lib.otvae_conv_multi(1, jobs, None) != 0
    assert b"kind" in lib.otvae_last_error()
    jobs[0].kind = L.JOB_FWD
    jobs[0].geom = L.ConvGeom(2, 8, 8, 8, 1, 8, 8, 8, 3, 3, 1, 1)
    assert lib.otvae_conv_multi(1, jobs, None) != 0  # NULL tensors
    assert lib.otvae_conv_multi(0, jobs, None) != 0


# ---- BASELINE batch size (1024): size-independent properties of the three passes ----------------------------------------
FULL = [
    (1024, 8, 8, 16, 3, 1, 1, 1),     # image-tile kernels (forward, data gradient, weight gradient)
    (1024, 16, 32, 8, 4, 2, 1, 1),    # stride 2, weight-gradient columns over blockIdx.y
    (1024, 64, 64, 2, 3, 1, 1, 1),    # deep layer: uniform-tap implicit GEMM
    (1024, 128, 64, 1, 3, 1, 1, 2),   # decoder entry with up-sampling
    (1024, 1, 8, 32, 4, 2, 1, 1),     # image side: direct kernels + tile weight gradient
    (1024, 8, 1, 16, 3, 1, 1, 2),     # reconstruction side
]


@pytest.mark.parametrize("case", FULL, ids=lambda c: "n%d_%dto%d_%dx%d_k%ds%dp%du%d" % (c[0], c[1], c[2], c[3], c[3], c[4], c[5], c[6], c[7]))
def test_full_batch_adjoint_and_equivariance(lib, case):
    """At batch 1024 (too large for a float64 restatement in seconds): with the activation removed the layer is linear, so
    <gy, conv(x)> = <dgrad(gy), x> = <wgrad(x, gy), w> + <gy, bias>  (the three passes are adjoints of one another), and
    permuting the images of the batch permutes the outputs (bit for bit: images never mix)."""
    n, cs, cn, hs, k, s, p, up = case
    c = make_case(n, cs, cn, hs, k, s, p, up, norm=False, relu=False, bias=True, res=False, seed=7)
    dv = Dev(c)
    y, _ = run_fwd(dv, stats=False)
    gv, _, _, _ = run_dgrad(dv, sums=False)
    gw, gb = run_wgrad(dv)
    gyd, xd = raw(dv.gy).double(), raw(dv.x).double()
    lhs = float((gyd * (raw(y).double() - dv.bias.double())).sum())          # <gy, conv(x)> without the bias term
    via_x = float((raw(gv).double() * xd).sum())
    via_w = float((gw.double() * dv.w_hwio.double()).sum())
    scale = float(gyd.abs().sum()) * float(raw(y).abs().max()) / gyd.numel() ** 0.5 + 1e-30
    assert abs(lhs - via_x) < 2e-5 * max(abs(lhs), scale), (lhs, via_x)
    assert abs(lhs - via_w) < 2e-5 * max(abs(lhs), scale), (lhs, via_w)
    assert rel(gb, gyd.reshape(-1, cn).sum(0)) < 1e-5
    # batch-permutation equivariance
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1)).cuda()
    dv2 = Dev(c)
    dv2.x = nhwc(dv.x[perm])
    y2, _ = run_fwd(dv2, stats=False)
    assert torch.equal(y2, nhwc(y[perm]))
