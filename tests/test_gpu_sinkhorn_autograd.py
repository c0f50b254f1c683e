"""``ot.sinkhorn_log`` carries the reference's autograd: the reference function is plain torch arithmetic and is
differentiable through all of its iterations with respect to a, b and C (ot/w2_utils.py:301-319).  Golden vectors:
``tests/golden/sinkhorn_autograd.npz`` from the REAL function (``oracle/gen_golden.py::gen_sinkhorn_autograd``).
Bounds: 1e-4 relative (fp32), 1e-8 (fp64), as VERDICT r3 #4 sets them."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu

DIRECT = ["n7_f64", "n7x9_f32", "n64_f32", "n64_f64", "n48x80_f64_reg01", "batch23_f64_thr", "n256_f32", "n256_f64"]
PRIOR = ["n7_f64", "n7_f32", "n64_f64", "n64_f32", "n256_f64", "n256_f32"]


def problem(lead, n, m, dtype, seed):
    """the inputs of a recorded case, regenerated from its seed exactly as oracle/gen_golden.py::gen_sinkhorn_autograd drew them"""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*lead, n, 5, generator=g, dtype=torch.float64)
    y = torch.randn(*lead, m, 5, generator=g, dtype=torch.float64) * 1.2 + 0.3
    C = ((x.unsqueeze(-2) - y.unsqueeze(-3)) ** 2).sum(-1)
    C = C / C.amax(dim=(-2, -1), keepdim=True)
    a = torch.rand(*lead, n, generator=g, dtype=torch.float64) + 0.1
    b = torch.rand(*lead, m, generator=g, dtype=torch.float64) + 0.1
    a, b = a / a.sum(-1, keepdim=True), b / b.sum(-1, keepdim=True)
    W = torch.randn(*lead, n, m, generator=g, dtype=torch.float64)
    return a.to(dtype), b.to(dtype), C.to(dtype), W.to(dtype)


def _tol(dtype):
    return 1e-8 if dtype == torch.float64 else 1e-4


@pytest.mark.parametrize("case", DIRECT)
def test_sinkhorn_log_gradients_vs_reference_autograd(case):
    from ot_vae_lightning_amd.ot import sinkhorn_log
    gold = load_golden("sinkhorn_autograd.npz")
    k = f"direct/{case}"
    reg, it, thr = gold[f"{k}/cfg"]
    ga_w, gb_w, gC_w = (torch.from_numpy(gold[f"{k}/{n}"]) for n in ("ga", "gb", "gC"))
    dtype = gC_w.dtype
    seed, n, m = (int(v) for v in gold[f"{k}/seed"])
    a, b, C, W = problem(tuple(gC_w.shape[:-2]), n, m, dtype, seed)
    if f"{k}/C" in gold.files:   # the small cases also store their inputs: the regeneration itself is pinned
        assert np.array_equal(C.numpy(), gold[f"{k}/C"]) and np.array_equal(a.numpy(), gold[f"{k}/a"])
    a, b, C, W = (t.cuda() for t in (a, b, C, W))
    a.requires_grad_(True), b.requires_grad_(True), C.requires_grad_(True)
    pi = sinkhorn_log(a, b, C, reg=float(reg), max_iter=int(it), threshold=float(thr))
    assert pi.grad_fn is not None
    key = f"{k}/pi" if f"{k}/pi" in gold.files else None
    if key is not None:
        assert rel_err(pi, torch.from_numpy(gold[key])) < _tol(dtype)
    else:
        assert rel_err(pi[..., :8, :8], torch.from_numpy(gold[f"{k}/pi_corner"])) < _tol(dtype)
    (pi * W).sum().backward()
    tol = _tol(dtype)
    assert rel_err(C.grad, gC_w) < tol, ("gC", rel_err(C.grad, gC_w))
    assert rel_err(a.grad, ga_w) < tol, ("ga", rel_err(a.grad, ga_w))
    assert rel_err(b.grad, gb_w) < tol, ("gb", rel_err(b.grad, gb_w))
    # only C needs a gradient: the marginals' adjoints are not formed, the cost gradient is unchanged
    C2 = C.detach().clone().requires_grad_(True)
    (sinkhorn_log(a.detach(), b.detach(), C2, reg=float(reg), max_iter=int(it), threshold=float(thr)) * W).sum().backward()
    assert torch.equal(C2.grad, C.grad)
    # no input requires a gradient: the single-launch solver, no graph, the same plan to rounding
    plain = sinkhorn_log(a.detach(), b.detach(), C.detach(), reg=float(reg), max_iter=int(it), threshold=float(thr))
    assert plain.grad_fn is None and rel_err(plain, pi) < (1e-12 if dtype == torch.float64 else 2e-6)


@pytest.mark.parametrize("case", PRIOR)
def test_sinkhorn_prior_gradient_conventions_vs_reference_autograd(case):
    """``SinkhornPrior(differentiate_plan=True)`` = autograd through the reference's composition (cost -> / max -> sinkhorn_log ->
    sum(C * pi)); the default is the envelope form (plan detached).  Both recorded from the reference's function."""
    import ot_vae_lightning_amd as A
    from ot_vae_lightning_amd.ot import sinkhorn_log, w2_utils as Wm
    gold = load_golden("sinkhorn_autograd.npz")
    k = f"prior/{case}"
    z0, y = torch.from_numpy(gold[f"{k}/z"]).cuda(), torch.from_numpy(gold[f"{k}/y"]).cuda()
    tol = _tol(z0.dtype)
    want_loss = float(gold[f"{k}/loss"])
    for flag, name in ((True, "gz_full"), (False, "gz_envelope")):
        prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, differentiate_plan=flag).cuda().train()
        z = z0.clone().requires_grad_(True)
        z_out, loss, _ = prior(z, 0, prior_samples=y)
        assert loss.shape == (z.shape[0],)
        assert abs(float(loss[0].detach()) - want_loss) / abs(want_loss) < max(tol, 1e-7), (name, float(loss[0].detach()), want_loss)
        loss.mean().backward()
        err = rel_err(z.grad, torch.from_numpy(gold[f"{k}/{name}"]))
        assert err < tol, (name, err)
    # the other read-out VERDICT r3 names: sum(Cn * pi) with the normalised cost in both places
    z = z0.clone().requires_grad_(True)
    C = Wm.sq_euclidean_cost(z, y)
    Cn = C / C.amax()
    n, m = C.shape
    a = torch.full((n,), 1.0 / n, device="cuda", dtype=z.dtype)
    b = torch.full((m,), 1.0 / m, device="cuda", dtype=z.dtype)
    loss_n = (Cn * sinkhorn_log(a, b, Cn, 0.05, 50, 0.0)).sum()
    loss_n.backward()
    assert abs(float(loss_n) - float(gold[f"{k}/loss_n"])) / float(gold[f"{k}/loss_n"]) < max(tol, 1e-7)
    assert rel_err(z.grad, torch.from_numpy(gold[f"{k}/gz_n"])) < tol


def test_differentiated_sinkhorn_prior_trains_a_vae_step():
    """the opt-in prior inside ``VAE.nelbo`` + ``HipTrainer``, eagerly issued and captured (the prior draws its own samples from the
    device generator, whose counter the capture's warm-up advances: the two routes see different draws, so each is only held to be
    finite and to give a positive OT term)"""
    import ot_vae_lightning_amd as A
    from detfill import mnist_like
    B = 64
    xs = [mnist_like(B, 310 + i).cuda() for i in range(2)]
    outs = []
    for graph in (False, True):
        torch.manual_seed(12)
        enc = A.CNN(1, 32, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(32, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        prior = A.SinkhornPrior(reg=0.05, max_iter=20, threshold=0.0, loss_coeff=0.5, differentiate_plan=True, seed=3)
        model = A.VAE(encoder=enc, decoder=dec, prior=prior).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=graph)
        outs.append(torch.stack([tr.step(x).clone() for x in xs]))
        tr.close()
    for o in outs:
        assert torch.isfinite(o).all() and (o[:, 2] > 0).all(), o


def test_sinkhorn_log_autograd_at_the_benchmark_size_vs_oracle():
    """BASELINE configs[2]'s plan size (1024 x 1024, reg 0.05, 50 iterations, fp32): the reverse sweep against the oracle's autograd
    (CPU, the reference's arithmetic: pinned to the reference's own gradients by tests/test_oracle_vs_golden.py), plus two
    size-independent properties of the exact derivative: (i) the plan's row sums are a for ANY cost after a full iteration (the last
    half-iteration normalises the rows), so the gradient of sum_j pi_ij with respect to C vanishes; (ii) d/dC of sum(pi * W) is
    linear in W."""
    import time
    import otvae_oracle as O
    from ot_vae_lightning_amd.ot import sinkhorn_log
    n = 1024
    g = torch.Generator().manual_seed(77)
    z, y = torch.randn(n, 16, generator=g) * 1.2 + 0.1, torch.randn(n, 16, generator=g)
    C0 = ((z.unsqueeze(1) - y.unsqueeze(0)) ** 2).sum(-1)
    C0 = C0 / C0.max()
    W = torch.randn(n, n, generator=g)
    a = torch.full((n,), 1.0 / n)
    b = torch.full((n,), 1.0 / n)
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    Cc = C0.clone().requires_grad_(True)
    (O.sinkhorn_log(a, b, Cc, reg=0.05, max_iter=50, threshold=0.0) * W).sum().backward()
    Cg = C0.cuda().requires_grad_(True)
    ag, bg, Wg = a.cuda(), b.cuda(), W.cuda()
    pi = sinkhorn_log(ag, bg, Cg, reg=0.05, max_iter=50, threshold=0.0)
    (pi * Wg).sum().backward()
    err = rel_err(Cg.grad, Cc.grad)
    assert err < 1e-4, err
    # (i) row sums do not depend on C
    Cr = C0.cuda().requires_grad_(True)
    sinkhorn_log(ag, bg, Cr, reg=0.05, max_iter=50, threshold=0.0).sum(-1)[::7].sum().backward()
    assert float(Cr.grad.abs().max()) < 1e-4 * float(Cg.grad.abs().max()), float(Cr.grad.abs().max())
    # (ii) linearity in the upstream gradient
    W2 = torch.randn(n, n, generator=g).cuda()
    grads = []
    for w_ in (Wg, W2, 0.5 * Wg - 2.0 * W2):
        Ck = C0.cuda().requires_grad_(True)
        (sinkhorn_log(ag, bg, Ck, reg=0.05, max_iter=50, threshold=0.0) * w_).sum().backward()
        grads.append(Ck.grad)
    assert rel_err(grads[2], 0.5 * grads[0] - 2.0 * grads[1]) < 2e-5
    # timing of the differentiable route at this size (reported, not asserted)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        Ck = C0.cuda().requires_grad_(True)
        (sinkhorn_log(ag, bg, Ck, reg=0.05, max_iter=50, threshold=0.0) * Wg).sum().backward()
    torch.cuda.synchronize()
    print(f"\n[sinkhorn autograd 1024^2 fp32, 50 iterations] forward + backward {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")
