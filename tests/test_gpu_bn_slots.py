"""BatchNorm statistic slots and the finalize folded into the consumer kernel (csrc/common.h ``bn_slot_add`` / ``bn_fold_prologue``,
functional.SlotArena / PendingFold).

What the reference computes here is nn.BatchNorm2d in training mode in front of every ConvLayer (networks/cnn.py:122, 183-192): batch
mean / biased variance for the normalisation, momentum update of the running buffers with the unbiased variance.  The slots route
must give those numbers (vs an fp64 restatement in the test), must not depend on the order in which blocks arrive (bit-equal
repeats), must propagate a non-finite input as NaN statistics without touching the running buffers, and a training step taken
with it must agree with the step taken through the per-block partials + finalize launch of rounds 1-3.
"""
import ctypes as C

import pytest
import torch

from detfill import mnist_like, normal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available()
    import ot_vae_lightning_amd as pkg
    return pkg


def _truth(x, gamma, beta, rmean, rvar, eps=1e-5, momentum=0.1):
    xd = x.double()
    m = xd.numel() // xd.shape[1]
    mu = xd.mean(dim=(0, 2, 3))
    var = xd.var(dim=(0, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var + eps)
    scale = gamma.double() * invstd
    shift = beta.double() - mu * scale
    unb = var * (m / (m - 1)) if m > 1 else var
    return mu, invstd, scale, shift, (1 - momentum) * rmean.double() + momentum * mu, (1 - momentum) * rvar.double() + momentum * unb


@pytest.mark.parametrize("nslots", [1, 4, 16, 64])
@pytest.mark.parametrize("shape", [(4, 3, 5, 7), (32, 32, 16, 16), (2, 257, 3, 3), (1, 1024, 1, 1), (64, 8, 32, 32)])
def test_slots_statistics_and_finalize_vs_fp64(A, shape, nslots):
    from ot_vae_lightning_amd import _lib, functional as HF
    from ot_vae_lightning_amd._lib import check, ptr, stream
    lib = _lib.load()
    n, c, h, w = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(shape, generator=g) * 3.0 + 1.5).cuda().contiguous(memory_format=torch.channels_last)
    gam = [(torch.rand(c, generator=g) + 0.5).cuda() for _ in range(2)]
    bet = [torch.randn(c, generator=g).cuda() for _ in range(2)]
    rm0 = [torch.randn(c, generator=g).cuda() for _ in range(2)]
    rv0 = [(torch.rand(c, generator=g) + 0.5).cuda() for _ in range(2)]
    m = n * h * w

    def run():
        rm, rv = [t.clone() for t in rm0], [t.clone() for t in rv0]
        nbt = [torch.tensor(3, device="cuda") for _ in range(2)]
        slots = torch.zeros(lib.otvae_bn_slots_words(c, nslots), device="cuda", dtype=torch.int64)
        check(lib.otvae_bn_stats_slots(ptr(x), m, c, ptr(slots), c, nslots, stream()), "stats_slots")
        br = [HF.BNBranch(gam[j], bet[j], rm[j], rv[j], nbt[j]) for j in range(2)]
        mean, invstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        sc, sh = [torch.empty(c, device="cuda") for _ in range(2)], [torch.empty(c, device="cuda") for _ in range(2)]
        fold = HF.PendingFold(HF.Slots(slots, c, nslots), m, br, True, mean, invstd, sc, sh)
        fold.materialize()
        torch.cuda.synchronize()
        return mean, invstd, sc, sh, rm, rv, nbt, slots

    mean, invstd, sc, sh, rm, rv, nbt, slots = run()
    assert int(slots[-2]) == 0, "no value was unrepresentable"
    for j in range(2):
        mu_t, is_t, sc_t, sh_t, rm_t, rv_t = _truth(x, gam[j], bet[j], rm0[j], rv0[j])
        # the sums are exact (integer limbs); what is left is one rounding to fp32 per number
        for got, want, name in ((mean, mu_t, "mean"), (invstd, is_t, "invstd"), (sc[j], sc_t, "scale"), (sh[j], sh_t, "shift"),
                                (rm[j], rm_t, "running_mean"), (rv[j], rv_t, "running_var")):
            err = float((got.double() - want).abs().max())
            tol = 4e-7 * float(want.abs().max()) + 1e-7
            assert err <= tol, (shape, j, name, err, tol)
        assert int(nbt[j]) == 4
    # order independence: the same bits every time, and the same numbers as the per-block fp64 partials of rounds 1-3 give
    again = run()
    for a, b in zip((mean, invstd, *sc, *sh, *rm, *rv), (again[0], again[1], *again[2], *again[3], *again[4], *again[5])):
        assert torch.equal(a, b)
    rm2, rv2 = [t.clone() for t in rm0], [t.clone() for t in rv0]
    br = [HF.BNBranch(gam[j], bet[j], rm2[j], rv2[j], torch.tensor(3, device="cuda")) for j in range(2)]
    mean_p, invstd_p, sc_p, sh_p = HF.bn_batch_stats(x, br)
    for a, b in zip((mean, invstd, *sc, *sh, *rm, *rv), (mean_p, invstd_p, *sc_p, *sh_p, *rm2, *rv2)):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-7), float((a - b).abs().max())


def test_slots_non_finite_input_reads_as_nan_and_keeps_the_running_buffers(A):
    from ot_vae_lightning_amd import _lib, functional as HF
    from ot_vae_lightning_amd._lib import check, ptr, stream
    lib = _lib.load()
    c = 16
    x = torch.randn(8, c, 4, 4).cuda().contiguous(memory_format=torch.channels_last)
    x[3, 5, 2, 1] = float("inf")
    slots = torch.zeros(lib.otvae_bn_slots_words(c, 8), device="cuda", dtype=torch.int64)
    check(lib.otvae_bn_stats_slots(ptr(x), 8 * 16, c, ptr(slots), c, 8, stream()), "stats_slots")
    rm, rv, nbt = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.tensor(0, device="cuda")
    br = [HF.BNBranch(torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), rm, rv, nbt)]
    mean, invstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    sc, sh = [torch.empty(c, device="cuda")], [torch.empty(c, device="cuda")]
    HF.PendingFold(HF.Slots(slots, c, 8), 128, br, True, mean, invstd, sc, sh).materialize()
    torch.cuda.synchronize()
    assert int(slots[-2]) > 0
    assert torch.isnan(mean).all() and torch.isnan(sc[0]).all(), "an unrepresentable sum poisons the tensor's statistics"
    assert torch.equal(rm, torch.zeros_like(rm)) and torch.equal(rv, torch.ones_like(rv)), "NaN statistics never enter the running buffers"
    with pytest.raises(ValueError, match="power of two"):
        check(lib.otvae_bn_stats_slots(ptr(x), 8 * 16, c, ptr(slots), c, 3, stream()), "stats_slots")


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("case", [(8, 8, 16, 16, 1), (4, 64, 2, 2, 2), (16, 6, 5, 5, 2), (2, 128, 1, 1, 1)])
def test_backward_pair_through_slots_vs_finalize_and_apply_launches(A, case, training):
    """A data-gradient job with its BatchNorm-backward sums into slots + ``otvae_bn_bwd_apply_slots`` against the same job with fp64
    partials + ``otvae_bn_bwd_finalize`` + ``otvae_bn_bwd_apply``, and both against the fp64 formula
    dx = sum_b k_b g_b - A x - B (csrc/bn.hip), dgamma = sum g xhat, dbeta = sum g."""
    from ot_vae_lightning_amd import _lib, functional as HF
    from ot_vae_lightning_amd._lib import check, ptr, ptr_array, stream
    lib = _lib.load()
    n, c, h, w, nb = case
    g_ = torch.Generator().manual_seed(sum(case))
    cl = torch.channels_last
    x = (torch.randn(n, c, h, w, generator=g_) * 2 + 0.5).cuda().contiguous(memory_format=cl)
    gv = [torch.randn(n, c, h, w, generator=g_).cuda().contiguous(memory_format=cl) for _ in range(nb)]
    gam = [(torch.rand(c, generator=g_) + 0.5).cuda() for _ in range(nb)]
    m = n * h * w
    xd = x.double()
    mu = xd.mean(dim=(0, 2, 3))
    var = xd.var(dim=(0, 2, 3), unbiased=False)
    mean, invstd = mu.float(), (1.0 / torch.sqrt(var + 1e-5)).float()
    xhat = (xd - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)
    # the sums, as a data-gradient kernel's epilogue leaves them: here from the stand-alone statistics kernels on (gv, gv * xhat)
    slots, parts = [], []
    for b in range(nb):
        s1 = gv[b].double().sum(dim=(0, 2, 3))
        s2 = (gv[b].double() * xhat).sum(dim=(0, 2, 3))
        sl = torch.zeros(lib.otvae_bn_slots_words(c, 4), device="cuda", dtype=torch.int64)
        # one "block" per slot carrying a share of the total: limbs hi (2^-10 units) / lo (2^-53 units), as bn_slot_add splits them
        S = 4
        for k in range(S):
            share = s1 / S if k < S - 1 else s1 - (s1 / S) * (S - 1)
            share2 = s2 / S if k < S - 1 else s2 - (s2 / S) * (S - 1)
            for stat, val in ((0, share), (1, share2)):
                hi = torch.floor(val * 1024.0)
                lo = torch.floor((val - hi / 1024.0) * 9007199254740992.0)
                view = sl[: S * 2 * c * 2].view(S, 2, c, 2)
                view[k, stat, :, 0] = hi.long()
                view[k, stat, :, 1] = lo.long()
        slots.append(HF.Slots(sl, c, S))
        part = torch.zeros((2, c, 3), device="cuda", dtype=torch.float64)   # [2][CsPad][P], P = 3
        part[0, :, 0], part[0, :, 1], part[0, :, 2] = s1 * 0.5, s1 * 0.25, s1 * 0.25
        part[1, :, 0], part[1, :, 1], part[1, :, 2] = s2 * 0.5, s2 * 0.25, s2 * 0.25
        parts.append(part)
    # slots route
    dga, dbe = [torch.empty(c, device="cuda") for _ in range(nb)], [torch.empty(c, device="cuda") for _ in range(nb)]
    dx = torch.empty_like(x)
    check(lib.otvae_bn_bwd_apply_slots(nb, ptr_array(gv), ptr(x), ptr_array([s_.buf for s_ in slots]), (C.c_int * nb)(*[s_.n for s_ in slots]),
                                       c, m, c, ptr(mean), ptr(invstd), ptr_array(gam), ptr_array(dga), ptr_array(dbe), int(training),
                                       ptr(dx), stream()), "apply_slots")
    # parameter gradients only
    dga0, dbe0 = [torch.empty(c, device="cuda") for _ in range(nb)], [torch.empty(c, device="cuda") for _ in range(nb)]
    check(lib.otvae_bn_bwd_apply_slots(nb, None, None, ptr_array([s_.buf for s_ in slots]), (C.c_int * nb)(*[s_.n for s_ in slots]),
                                       c, m, c, ptr(mean), ptr(invstd), ptr_array(gam), ptr_array(dga0), ptr_array(dbe0), int(training),
                                       None, stream()), "apply_slots(no dx)")
    # launches route
    dga2, dbe2 = [torch.empty(c, device="cuda") for _ in range(nb)], [torch.empty(c, device="cuda") for _ in range(nb)]
    coef = torch.empty((2 + nb, c), device="cuda")
    check(lib.otvae_bn_bwd_finalize(nb, ptr_array(parts), (C.c_int * nb)(*([3] * nb)), c, m, c, ptr(mean), ptr(invstd), ptr_array(gam),
                                    ptr_array(dga2), ptr_array(dbe2), ptr(coef), stream()), "finalize")
    if not training:
        coef[:2].zero_()
    dx2 = torch.empty_like(x)
    check(lib.otvae_bn_bwd_apply(nb, ptr_array(gv), ptr(x), ptr(coef), m, c, ptr(dx2), stream()), "apply")
    torch.cuda.synchronize()
    # fp64 truth
    Aa = torch.zeros(c, dtype=torch.float64, device="cuda")
    Bb = torch.zeros_like(Aa)
    dx_t = torch.zeros_like(xd)
    for b in range(nb):
        s1 = gv[b].double().sum(dim=(0, 2, 3))
        s2 = (gv[b].double() * xhat).sum(dim=(0, 2, 3))
        k = gam[b].double() * invstd.double()
        Aa += k * s2
        Bb += k * s1
        dx_t += k.view(1, -1, 1, 1) * gv[b].double()
        for got in (dga[b], dga0[b], dga2[b]):
            assert float((got.double() - s2).abs().max()) <= 1e-6 * float(s2.abs().max()) + 1e-6
        for got in (dbe[b], dbe0[b], dbe2[b]):
            assert float((got.double() - s1).abs().max()) <= 1e-6 * float(s1.abs().max()) + 1e-6
    Aa = Aa * invstd.double() / m
    Bb = Bb / m - Aa * mean.double()
    if training:
        dx_t = dx_t - Aa.view(1, -1, 1, 1) * xd - Bb.view(1, -1, 1, 1)
    # dx is a difference of terms of size k |g| (on a 1x1 map with two samples they cancel almost completely): fp32 rounding of the terms
    scale_ = max(float((gam[b].double() * invstd.double()).view(1, -1, 1, 1).mul(gv[b].double().abs()).max()) for b in range(nb))
    assert float((dx.double() - dx_t).abs().max()) <= 2e-6 * scale_ + 1e-6
    assert float((dx2.double() - dx_t).abs().max()) <= 2e-6 * scale_ + 1e-6


CONFIGS = {
    # BASELINE configs[1] / configs[3] shapes (MNIST 1x32x32 capacity 8 latent 128; CIFAR-10 3x32x32 capacity 16 latent 256); the
    # default max_attn_res = 16 puts AttentionBlocks at 16x16 and below, max_attn_res = 32 at every resolution
    "mnist": dict(cin=1, latent=128, capacity=8, max_attn_res=16),
    "mnist_attn32": dict(cin=1, latent=128, capacity=8, max_attn_res=32),
    "cifar": dict(cin=3, latent=256, capacity=16, max_attn_res=16),
}


def _model(A, seed, cin, latent, capacity, max_attn_res):
    torch.manual_seed(seed)
    enc = A.CNN(cin, 2 * latent, 32, 1, capacity=capacity, down_sample=True, residual="add", max_attn_res=max_attn_res)
    dec = A.CNN(latent, cin, 1, 32, capacity=capacity, up_sample=True, residual="add", max_attn_res=max_attn_res)
    return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_training_steps_through_slots_agree_with_the_finalize_launch_route(A, monkeypatch, graph, cfg):
    """Three optimizer steps with the forward BatchNorm finalize folded into the consumer kernels (statistic slots) against the same
    steps through per-block partials + ``otvae_bn_finalize``: the two routes round the sums differently (exact integer limbs vs
    a fixed-order fp64 sum), so the steps agree to fp32 rounding, not bit for bit; each route is bit-reproducible by itself."""
    from ot_vae_lightning_amd import functional as HF
    B, kw = 64, CONFIGS[cfg]
    x = [normal((B, kw["cin"], 32, 32), 140 + i).cuda() if kw["cin"] > 1 else mnist_like(B, 140 + i).cuda() for i in range(3)]
    eps = [normal((B, kw["latent"], 1, 1), 150 + i).cuda() for i in range(3)]
    taken = []

    def run(mode):
        monkeypatch.setattr(HF, "BN_SLOTS_MODE", mode)
        model = _model(A, 21, **kw)
        tr = A.HipTrainer(model, batch_shape=(B, kw["cin"], 32, 32), use_graph=graph, data_parallel=False)
        losses = [tr.step(x[i], eps[i]).clone() for i in range(3)]
        torch.cuda.synchronize()
        st = HF.SlotArena._state.get(tr.device)
        taken.append(0 if (st is None or mode == 0) else st[1])
        bufs = torch.cat([b.detach().float().flatten() for b in model.buffers()])
        out = tr.pflat.clone(), torch.stack(losses), bufs
        tr.close()
        return out

    a, b, a2, f = run(2), run(0), run(2), run(1)
    assert taken[0] > taken[3] > 0 and taken[1] == 0, "slots were taken: forward + backward (2), forward only (1), never (0)"
    for u, v in zip(a, a2):
        assert torch.equal(u, v), "the slots route is order independent: same bits on a second run"
    for got in (a, f):
        for u, v, name in zip(got, b, ("params", "losses", "buffers")):
            err = float((u.double() - v.double()).norm() / v.double().norm())
            assert err < 2e-5, (name, err)


def test_an_arena_that_runs_out_mid_pass_falls_back_to_partials_and_changes_nothing_else(A, monkeypatch):
    """The arena is a fixed 8 MiB; a pass that wants more takes the partial + finalize-launch route for the tensors that no longer fit
    (also between the two branches of one layer).  Same numbers as a pass that fits, to the rounding the two routes differ by."""
    from ot_vae_lightning_amd import functional as HF
    B, kw = 64, CONFIGS["mnist"]
    x = [mnist_like(B, 160 + i).cuda() for i in range(2)]
    eps = [normal((B, kw["latent"], 1, 1), 170 + i).cuda() for i in range(2)]

    def run(words, mode, graph):
        monkeypatch.setattr(HF, "BN_SLOTS_MODE", mode)
        monkeypatch.setattr(HF.SlotArena, "WORDS", words)
        HF.SlotArena._state.clear()   # a fresh arena of the patched size
        tr = A.HipTrainer(_model(A, 31, **kw), batch_shape=(B, 1, 32, 32), use_graph=graph, data_parallel=False)
        losses = [tr.step(x[i], eps[i]).clone() for i in range(2)]
        torch.cuda.synchronize()
        st = HF.SlotArena._state.get(tr.device)
        used = st[3] if st is not None else 0
        out = tr.pflat.clone(), torch.stack(losses)
        tr.close()
        return out, used

    try:
        full, used_full = run(1 << 20, 2, True)
        for words in (used_full // 2, used_full // 7, 2048):
            for graph in (False, True):
                small, used = run(words, 2, graph)
                assert 0 < used <= words < used_full, "the arena did run out"
                for u, v, name in zip(small, full, ("params", "losses")):
                    err = float((u.double() - v.double()).norm() / v.double().norm())
                    assert err < 2e-5, (words, graph, name, err)
    finally:
        HF.SlotArena._state.clear()


def test_two_engines_share_the_arena_without_seeing_each_other(A):
    """One arena per device, zeroed at the start of every step: two engines (one captured, one issued eagerly, different networks)
    stepping in turn give the bits each gives alone."""
    from ot_vae_lightning_amd import functional as HF
    B = 32
    xa = [mnist_like(B, 180 + i).cuda() for i in range(3)]
    ea = [normal((B, 128, 1, 1), 185 + i).cuda() for i in range(3)]
    xb = [normal((B, 3, 32, 32), 190 + i).cuda() for i in range(3)]
    eb = [normal((B, 256, 1, 1), 195 + i).cuda() for i in range(3)]

    def make():
        ta = A.HipTrainer(_model(A, 41, **CONFIGS["mnist"]), batch_shape=(B, 1, 32, 32), use_graph=True, data_parallel=False)
        tb = A.HipTrainer(_model(A, 42, **CONFIGS["cifar"]), batch_shape=(B, 3, 32, 32), use_graph=False, data_parallel=False)
        return ta, tb

    ta, tb = make()
    alone_a = [ta.step(xa[i], ea[i]).clone() for i in range(3)]
    alone_b = [tb.step(xb[i], eb[i]).clone() for i in range(3)]
    pa, pb = ta.pflat.clone(), tb.pflat.clone()
    ta.close(); tb.close()
    ta, tb = make()
    turn_a, turn_b = [], []
    for i in range(3):
        turn_b.append(tb.step(xb[i], eb[i]).clone())
        turn_a.append(ta.step(xa[i], ea[i]).clone())
    torch.cuda.synchronize()
    assert all(torch.equal(u, v) for u, v in zip(alone_a, turn_a)) and torch.equal(pa, ta.pflat)
    assert all(torch.equal(u, v) for u, v in zip(alone_b, turn_b)) and torch.equal(pb, tb.pflat)
    ta.close(); tb.close()
