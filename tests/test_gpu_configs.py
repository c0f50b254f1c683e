"""GPU parity for the remaining BASELINE configurations and API surface (run with ``-m gpu``):
  * configs[0] (README shape: residual=None, batch 250) and configs[3] (CIFAR-10 3x32x32, capacity 16, latent 256):
    one training step vs the CPU oracle on identical weights / batch / eps;
  * configs[2]: the minibatch Sinkhorn OT prior (eps 0.05, 50 iterations) inside VAE.nelbo, loss and gradients;
  * inference mode (BatchNorm running statistics), ``VAE.forward/encode/decode/sample`` shapes as asserted by the
    reference's tests/test_mnist_cnn_vae.py:217-226, ``AutoEncoder``, ``load_state_dict`` round trip;
  * error behaviour the reference's validators have (ValueError on bad shapes, NotImplementedError on options outside
    the hot path).
"""
import os

import pytest
import torch

import otvae_oracle as O
from conftest import rel_err
from detfill import mnist_like, normal
from test_gpu_parity import Report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available()
    import ot_vae_lightning_amd as pkg
    return pkg


def _pnames(model):
    return [f"{net}.{k}" for net, m_ in (("encoder", model.encoder), ("decoder", model.decoder)) for k, _ in m_.named_parameters()]


def _oracle_step(model, enc_kw, dec_kw, x, eps, loss_coeff):
    ea, da = O.cnn_arch(**enc_kw), O.cnn_arch(**dec_kw)
    pe = {k: v.detach().cpu().clone().contiguous() for k, v in model.encoder.state_dict().items()}
    pd = {k: v.detach().cpu().clone().contiguous() for k, v in model.decoder.state_dict().items()}
    leaves = [v.requires_grad_(True) for d in (pe, pd) for k, v in d.items() if v.is_floating_point() and "running" not in k]
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    r = O.vae_nelbo(x, eps, pe, pd, ea, da, loss_coeff=loss_coeff)
    r["loss"].backward()
    return r, leaves, pe, pd


@pytest.mark.parametrize("cfg", ["readme_mnist_b250", "cifar_b32"])
def test_training_step_other_configs_vs_oracle(A, cfg):
    rep = Report(f"training step {cfg} vs CPU oracle")
    torch.manual_seed(1)
    if cfg == "readme_mnist_b250":
        B, cin, lat, cap, res = 250, 1, 128, 8, None
        x = mnist_like(B, 21)
    else:
        B, cin, lat, cap, res = 32, 3, 256, 16, "add"
        x = normal((B, 3, 32, 32), 22)
    enc_kw = dict(in_features=cin, out_features=2 * lat, in_resolution=32, out_resolution=1, capacity=cap,
                  down_sample=True, residual=res)
    dec_kw = dict(in_features=lat, out_features=cin, in_resolution=1, out_resolution=32, capacity=cap, up_sample=True,
                  residual=res)
    enc = A.CNN(cin, 2 * lat, 32, 1, capacity=cap, down_sample=True, residual=res)
    dec = A.CNN(lat, cin, 1, 32, capacity=cap, up_sample=True, residual=res)
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1))
    eps = normal((B, lat, 1, 1), 23)
    r, leaves, pe, pd = _oracle_step(model, enc_kw, dec_kw, x, eps, 0.1)
    model = model.cuda().train()
    tr = A.HipTrainer(model, batch_shape=tuple(x.shape), use_graph=False)
    out = tr.step(x.cuda(), eps.cuda())
    rep.check("loss[total,recon,prior]", out, torch.stack([r["loss"], r["recon"], r["prior"]]).detach())
    params = [p for net in (model.encoder, model.decoder) for p in net.parameters()]
    gl2 = torch.tensor([p.grad.double().norm().item() for p in params])
    rep.check("grad_l2 (all parameters)", gl2, torch.tensor([v.grad.double().norm().item() for v in leaves]), tol=5e-4)
    rep.check_grads(f"gradients [{cfg}]", [p.grad for p in params], [v.grad for v in leaves], _pnames(model))
    # BatchNorm running statistics after the step
    rs_gpu = torch.tensor([b.double().sum().item() for net in (model.encoder, model.decoder)
                           for k, b in net.named_buffers() if "running_" in k])
    rs_cpu = torch.tensor([v.double().sum().item() for d in (pe, pd) for k, v in d.items() if "running_" in k])
    rep.check("BatchNorm running statistics", rs_gpu, rs_cpu)
    with torch.no_grad():
        loss, logs, art = model.nelbo({"samples": x.cuda(), "target": x.cuda(), "kwargs": {"eps": eps.cuda()}}, 0)
    assert art["preds"].shape == x.shape and art["latents"].shape == (B, lat, 1, 1)
    rep.finish()


def test_benchmark_config_batch1024_gradients_elementwise_vs_oracle(A):
    """BASELINE configs[1] exactly as benched (MNIST-32 CNN VAE + GaussianPrior(0.1), batch 1024, residual="add"): the captured
    step's loss vector and EVERY parameter gradient element by element against the CPU oracle (norms only before)."""
    rep = Report("configs[1] (benchmark) B=1024: loss + element-wise gradients vs CPU oracle")
    B = 1024
    torch.manual_seed(0)
    enc_kw = dict(in_features=1, out_features=256, in_resolution=32, out_resolution=1, capacity=8, down_sample=True, residual="add")
    dec_kw = dict(in_features=128, out_features=1, in_resolution=1, out_resolution=32, capacity=8, up_sample=True, residual="add")
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1))
    x, eps = mnist_like(B, 41), normal((B, 128, 1, 1), 42)
    r, leaves, pe, pd = _oracle_step(model, enc_kw, dec_kw, x, eps, 0.1)
    model = model.cuda().train()
    tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=True)
    out = tr.step(x.cuda(), eps.cuda()).clone()
    torch.cuda.synchronize()
    rep.check("loss[total,recon,prior]", out, torch.stack([r["loss"], r["recon"], r["prior"]]).detach())
    grads = [p._otvae_grad_view() for net in (model.encoder, model.decoder) for p in net.parameters()]  # the flat buffer's slots
    rep.check_grads("gradients B=1024 (captured step)", grads, [v.grad for v in leaves], _pnames(model))
    tr.close()
    rep.finish()


@pytest.mark.parametrize("cfg", ["mnist_b128", "cifar_b256_config3", "mnist_b1024_coeff"])
def test_sinkhorn_prior_in_vae_step_vs_oracle(A, cfg):
    """configs[2] / configs[3]: deterministic encoder + entropic OT (eps 0.05, 50 iterations) between the minibatch of latents
    and N(0, I) draws.  ``cifar_b256_config3`` is BASELINE configs[3] as written, per GPU: CIFAR-10 3 x 32 x 32, capacity 16,
    latent 256 x 1 x 1, SinkhornPrior(0.05, 50), 256 images (2048 over 8 GPUs; the OT term is rank-local like any batch loss
    under DDP).  ``mnist_b1024_coeff``: the benchmark's 1024 x 1024 plan with a loss coefficient folded into the kernels."""
    rep = Report(f"VAE + SinkhornPrior (eps=0.05, 50 it) [{cfg}] vs CPU oracle")
    if cfg == "cifar_b256_config3":
        B, cin, lat, cap, coeff = 256, 3, 256, 16, 1.0
        x = normal((B, 3, 32, 32), 33)
    else:
        B, cin, lat, cap, coeff = (128, 1, 128, 8, 1.0) if cfg == "mnist_b128" else (1024, 1, 128, 8, 0.3)
        x = mnist_like(B, 31)
    torch.manual_seed(3)
    enc = A.CNN(cin, lat, 32, 1, capacity=cap, down_sample=True, residual="add")
    dec = A.CNN(lat, cin, 1, 32, capacity=cap, up_sample=True, residual="add")
    prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, loss_coeff=coeff)
    model = A.VAE(encoder=enc, decoder=dec, prior=prior)
    ps = normal((B, lat), 32)
    # oracle
    ea = O.cnn_arch(cin, lat, 32, 1, capacity=cap, down_sample=True, residual="add")
    da = O.cnn_arch(lat, cin, 1, 32, capacity=cap, up_sample=True, residual="add")
    pe = {k: v.detach().clone().contiguous() for k, v in model.encoder.state_dict().items()}
    pd = {k: v.detach().clone().contiguous() for k, v in model.decoder.state_dict().items()}
    leaves = [v.requires_grad_(True) for d in (pe, pd) for k, v in d.items() if v.is_floating_point() and "running" not in k]
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    h = O.cnn_forward(x, pe, ea)
    z = h
    ot = O.sinkhorn_ot_loss(z.flatten(1), ps, reg=0.05, max_iter=50, threshold=0.0)
    preds = O.cnn_forward(z, pd, da)
    recon = torch.nn.functional.mse_loss(preds, x)
    # envelope gradient: the plan is treated as a constant (SinkhornPrior's documented semantics)
    C = O.sq_euclidean_cost(z.flatten(1), ps)
    with torch.no_grad():
        a = torch.full((B,), 1.0 / B)
        pi = O.sinkhorn_log(a, a, C / C.max(), reg=0.05, max_iter=50, threshold=0.0)
    loss = recon + coeff * (C * pi).sum() / x[0].numel()
    loss.backward()
    # product
    model = model.cuda().train()
    lossg, logs, art = model.nelbo({"samples": x.cuda(), "target": x.cuda(), "kwargs": {"prior_samples": ps.cuda()}}, 0)
    lossg.backward()
    assert int(prior.last_iters) == 50
    rep.check("OT cost", logs["train/loss/prior"] * x[0].numel() / coeff, ot.detach())
    rep.check("total loss", lossg, loss.detach())
    params = [p for net in (model.encoder, model.decoder) for p in net.parameters()]
    rep.check("grad_l2 (all parameters)", torch.tensor([p.grad.double().norm().item() for p in params]),
              torch.tensor([v.grad.double().norm().item() for v in leaves]), tol=5e-4)
    rep.check_grads(f"gradients [{cfg}]", [p.grad for p in params], [v.grad for v in leaves], _pnames(model))
    rep.finish()


def test_sinkhorn_prior_step_launches_no_aten_kernels(A):
    """The Sinkhorn-prior path (configs[2]/[3]) is native end to end: a profiler trace of prior forward + backward shows only
    this library's kernels -- no library GEMM, no ATen max / div / fill / mul / copy."""
    from torch.profiler import ProfilerActivity, profile
    z = normal((256, 128), 51).cuda().requires_grad_(True)
    ps = normal((256, 128), 52).cuda()
    prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, loss_coeff=0.5).cuda()
    g = torch.full((256,), 1.0 / 256, device="cuda")

    def run():
        z.grad = None
        zz, loss, _ = prior(z, step=0, prior_samples=ps)
        torch.autograd.backward(loss, grad_tensors=[g], inputs=[z])

    run()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        run()
        torch.cuda.synchronize()
    names = sorted({e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA})
    assert names, "the profiler saw no device kernels"
    ours = ("sqdist", "sk_", "ot_cost")
    foreign = [n for n in names if not any(tag in n for tag in ours)]
    assert not foreign, foreign
    # the drawn-samples path: the device generator instead of an ATen philox kernel
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        prior(z, step=0)
        torch.cuda.synchronize()
    names = sorted({e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA})
    assert any("normal_fill" in n for n in names) and not [n for n in names if "at::" in n or "Cijk" in n], names


def test_sinkhorn_starved_solver_cannot_pass_for_a_result(A, monkeypatch):
    """The single-launch solver's workgroups wait for each other; when a wait runs out (forced here with a poll budget of
    zero) the plan, the potentials and the prior loss are NaN, iters reports -1 and the eager wrappers raise; the
    one-launch-per-half-iteration path is unaffected and agrees with an unstarved persistent solve."""
    from ot_vae_lightning_amd.ot import w2_utils as W
    z, ps = normal((1024, 128), 53).cuda(), normal((1024, 128), 54).cuda()
    a = torch.full((1024,), 1.0 / 1024, device="cuda")
    C = W.sq_euclidean_cost(z, ps)
    Cn = C / C.max()
    good = W.sinkhorn_log(a, a, Cn, reg=0.05, max_iter=50, threshold=0.0)
    monkeypatch.setenv("OTVAE_SK_SPIN_LIMIT", "0")
    with pytest.raises(W.SinkhornSolverStarved):
        W.sinkhorn_log(a, a, Cn, reg=0.05, max_iter=50, threshold=0.0)
    pi, u, v, iters = W.sinkhorn_log_potentials(a, a, Cn, reg=0.05, max_iter=50, threshold=0.0, check_starved=False)
    assert int(iters) == -1 and torch.isnan(pi).all() and torch.isnan(u).all() and torch.isnan(v).all()
    prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0)
    _, loss, _ = prior(z, step=0, prior_samples=ps)               # the step never synchronises: the loss is NaN ...
    assert torch.isnan(loss).all()
    with pytest.raises(W.SinkhornSolverStarved):                     # ... and the host check says why
        prior.raise_if_starved()
    # inside a capture nothing can be read back: the NaN loss is the signal
    static_z = z.clone()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
        _, loss, _ = prior(static_z, step=0, prior_samples=ps)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.isnan(loss).all() and int(prior.last_iters) == -1
    monkeypatch.setenv("OTVAE_SK_MULTILAUNCH", "1")
    multi = W.sinkhorn_log(a, a, Cn, reg=0.05, max_iter=50, threshold=0.0)
    assert torch.equal(multi, good)


def _w2_prior_case(G, D, B):
    import math
    kk = f"D{D}_B{B}"
    if B <= 256:
        return torch.from_numpy(G[f"{kk}/z"])
    g = torch.Generator().manual_seed(900 + D + B)   # the generator calls of oracle/gen_golden.py:gen_w2_prior
    mix = torch.randn(D, D, generator=g) / math.sqrt(D)
    return (torch.randn(B, D, generator=g) @ (0.6 * mix + 0.7 * torch.eye(D)) + 0.3 * torch.randn(D, generator=g)).float()


@pytest.mark.parametrize("shape", [(16, 64), (128, 256), (128, 1024)])
def test_gaussian_w2_prior_vs_reference_autograd(A, shape):
    """GaussianW2Prior (W2^2 between the batch's empirical Gaussian and a target Gaussian, HIP forward + closed-form HIP
    backward) against the reference's own ``GaussianModel._stats -> mean_cov -> w2_gaussian`` run under torch.autograd
    (tests/golden/w2_prior.npz): loss and dL/dz at 1e-4 (north_star), measured ~1e-7.  Standard-normal and general targets;
    loss coefficient and the decoder-side gradient folded into the kernels."""
    from conftest import load_golden
    G = load_golden("w2_prior.npz")
    D, B = shape
    rep = Report(f"GaussianW2Prior D={D} B={B} vs the reference under torch.autograd")
    z0 = _w2_prior_case(G, D, B)
    for tag in ("std", "gen"):
        k = f"D{D}_B{B}/{tag}"
        tm = None if tag == "std" else torch.from_numpy(G[f"D{D}_B{B}/target_mean"])
        tc = None if tag == "std" else torch.from_numpy(G[f"D{D}_B{B}/target_cov"])
        prior = A.GaussianW2Prior(loss_coeff=0.25, target_mean=tm, target_cov=tc).cuda()
        z = z0.clone().cuda().reshape(B, D, 1, 1).requires_grad_(True)
        zz, loss, _ = prior(z, step=0)
        assert zz.shape == z.shape and loss.shape == (B,)
        loss.mean().backward(retain_graph=True)
        gz = z.grad.flatten(1).cpu() / 0.25
        # a decoder-side gradient reaching z through the prior's identity output is added inside the backward kernel
        extra = 1e-3 * torch.sin(torch.arange(B * D, dtype=torch.float32).reshape(B, D, 1, 1)).cuda()
        pure = z.grad.clone()
        z.grad = None
        (loss.mean() + (zz * extra).sum()).backward()
        rep.check(f"{tag}: decoder-side gradient folded in", z.grad - pure, extra, tol=1e-5)
        rep.check(f"{tag}: loss", loss[0] / 0.25, torch.from_numpy(G[f"{k}/loss"]).float(), tol=1e-4)
        if B <= 256:
            rep.check(f"{tag}: dL/dz", gz, torch.from_numpy(G[f"{k}/gz"]), tol=1e-4)
        else:
            rep.check(f"{tag}: dL/dz (head)", gz[:8], torch.from_numpy(G[f"{k}/gz_head"]), tol=1e-4)
            rep.check(f"{tag}: dL/dz (row sums)", gz.double().sum(1), torch.from_numpy(G[f"{k}/gz_rowsum"]), tol=1e-4)
            rep.check(f"{tag}: dL/dz (column sums)", gz.double().sum(0), torch.from_numpy(G[f"{k}/gz_colsum"]), tol=1e-4)
        # the oracle on the same inputs (the chain the other parity tests use)
        zo = z0.clone().requires_grad_(True)
        lo = O.w2_prior_loss(zo, tm, tc)
        lo.backward()
        rep.check(f"{tag}: loss vs oracle", loss[0] / 0.25, lo.detach().float(), tol=1e-4)
        rep.check(f"{tag}: dL/dz vs oracle", gz, zo.grad, tol=1e-4)
        s = prior.sample((4, D, 1, 1), "cuda")
        assert s.shape == (4, D, 1, 1) and torch.isfinite(s).all()
    with pytest.raises(ValueError):
        A.GaussianW2Prior(target_mean=torch.zeros(3), target_cov=torch.eye(4))
    rep.finish()


def test_gaussian_w2_prior_in_vae_training_step(A):
    """The W2 prior inside VAE.nelbo through HipTrainer (eager and captured: same bits), against the oracle's step."""
    rep = Report("VAE + GaussianW2Prior training step vs CPU oracle")
    B, lat = 256, 128
    x = mnist_like(B, 35)

    def make():
        torch.manual_seed(9)
        enc = A.CNN(1, lat, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(lat, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianW2Prior(loss_coeff=0.5))

    model = make()
    ea = O.cnn_arch(1, lat, 32, 1, capacity=8, down_sample=True, residual="add")
    da = O.cnn_arch(lat, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    pe = {k: v.detach().clone().contiguous() for k, v in model.encoder.state_dict().items()}
    pd = {k: v.detach().clone().contiguous() for k, v in model.decoder.state_dict().items()}
    leaves = [v.requires_grad_(True) for d in (pe, pd) for k, v in d.items() if v.is_floating_point() and "running" not in k]
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    h = O.cnn_forward(x, pe, ea)
    w2 = O.w2_prior_loss(h.flatten(1))
    recon = torch.nn.functional.mse_loss(O.cnn_forward(h, pd, da), x)
    loss = recon + 0.5 * w2.float() / x[0].numel()
    loss.backward()
    outs = []
    for graph in (False, True):
        tr = A.HipTrainer(make().cuda().train(), batch_shape=(B, 1, 32, 32), use_graph=graph)
        out = tr.step(x.cuda()).clone()
        outs.append((out, tr.gflat.clone(), tr.pflat.clone()))
        if not graph:
            rep.check("loss[total,recon,prior]", out, torch.stack([loss, recon, 0.5 * w2.float() / x[0].numel()]).detach())
            params = [p for net in (tr.model.encoder, tr.model.decoder) for p in net.parameters()]
            rep.check("grad_l2 (all parameters)", torch.tensor([p.grad.double().norm().item() for p in params]),
                      torch.tensor([v.grad.double().norm().item() for v in leaves]), tol=5e-4)
            rep.check_grads("gradients [GaussianW2Prior step]", [p.grad for p in params], [v.grad for v in leaves], _pnames(tr.model))
        # steps 2-4 start the eigendecomposition from the previous step's eigenvectors (the first one cold, in both modes)
        later = torch.stack([tr.step(x.cuda()).clone() for _ in range(3)])
        assert int(tr.model.prior._warm) == 1 and torch.isfinite(later).all()
        outs[-1] = outs[-1] + (later, tr.pflat.clone())
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    rep.finish()


def test_gaussian_w2_prior_rank_deficient_batch_with_a_general_target_stays_finite(A):
    """B <= D makes the batch covariance rank deficient; with a general ``target_cov`` the eigenvalues of
    covt^1/2 S covt^1/2 then hold zeros, lambda^-1/4 was infinite and loss / gradient came out inf / NaN (the reference shifts S by
    1e-8 through make_pd and stays finite).  The tail kernel now clamps at that size."""
    D, B = 32, 16
    g = torch.Generator().manual_seed(41)
    a = torch.randn(D, D, generator=g, dtype=torch.double) / D ** 0.5
    prior = A.GaussianW2Prior(loss_coeff=1.0, target_mean=torch.zeros(D, dtype=torch.double),
                              target_cov=a @ a.T + 0.5 * torch.eye(D, dtype=torch.double)).cuda().train()
    z = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
    _, loss, _ = prior(z, step=0)
    loss.mean().backward()
    assert torch.isfinite(loss).all() and torch.isfinite(z.grad).all() and float(z.grad.abs().max()) > 0


def test_device_normal_generator(A):
    """``otvae_normal_fill``: standard-normal moments, a fresh draw per call (the counter advances on the device, also inside a
    replayed graph), the same values for the same (seed, counter) whatever the launch, independent streams."""
    from ot_vae_lightning_amd import functional as HF
    key = HF.new_rng_key("cuda", seed=1234)
    a = HF.normal_like(torch.empty(1 << 20, device="cuda"), key)
    b = HF.normal_like(torch.empty(1 << 20, device="cuda"), key)
    assert key.tolist() == [1234, 2, 0]
    for t in (a, b):
        d = t.double()
        assert abs(float(d.mean())) < 5e-3 and abs(float(d.var()) - 1) < 5e-3
        assert abs(float((d ** 3).mean())) < 2e-2 and abs(float((d ** 4).mean()) - 3) < 5e-2 and float(d.abs().max()) < 7
    assert abs(float((a.double() * b.double()).mean())) < 5e-3 and not torch.equal(a, b)
    assert abs(float((a[:-1].double() * a[1:].double()).mean())) < 5e-3          # neighbouring elements are uncorrelated
    key2 = HF.new_rng_key("cuda", seed=1234)
    short = HF.normal_like(torch.empty(1001, device="cuda"), key2)                # another launch shape, odd length
    assert torch.equal(short, a[:1001])
    other = HF.normal_fill_(torch.empty(1001, device="cuda"), HF.new_rng_key("cuda", seed=1234), stream_id=1)
    assert not torch.equal(other, short)
    out = torch.zeros(4096, device="cuda")
    k3 = HF.new_rng_key("cuda", seed=7)
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
        HF.normal_fill_(out, k3)
    draws = []
    for _ in range(3):
        graph.replay()
        draws.append(out.clone())
    torch.cuda.synchronize()
    assert not torch.equal(draws[0], draws[1]) and not torch.equal(draws[1], draws[2]) and int(k3[1]) == 3


def test_inference_mode_and_api_shapes(A):
    """Shapes asserted by the reference's inference test (tests/test_mnist_cnn_vae.py:217-226) + eval-mode parity."""
    rep = Report("inference (BatchNorm running statistics) vs CPU oracle; API shapes")
    torch.manual_seed(5)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1))
    assert model.latent_size == torch.Size((128, 1, 1))
    # give the running statistics non-trivial values
    with torch.no_grad():
        for k, b in model.state_dict().items():
            if k.endswith("running_mean"):
                b.copy_(0.1 * torch.sin(torch.arange(b.numel(), dtype=torch.float32)))
            if k.endswith("running_var"):
                b.copy_(1.0 + 0.3 * torch.cos(torch.arange(b.numel(), dtype=torch.float32)))
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = mnist_like(10, 41)
    ea = O.cnn_arch(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    da = O.cnn_arch(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    pe = {k[len("encoder."):]: v.contiguous() for k, v in sd.items() if k.startswith("encoder.")}
    pd = {k[len("decoder."):]: v.contiguous() for k, v in sd.items() if k.startswith("decoder.")}
    with torch.no_grad():
        h = O.cnn_forward(x, pe, ea, training=False)
        rec = O.cnn_forward(h[:, :128], pd, da, training=False)
    model = model.cuda().eval()
    with torch.no_grad():
        hg = model.encoder(x.cuda())
        recg = model.decoder(hg[:, :128].contiguous())
        rep.check("encoder output (eval)", hg, h)
        rep.check("decoder output (eval)", recg, rec)
        lat = model.encode(x.cuda())
        assert lat.shape == (10, 128, 1, 1)
        assert model.decode(lat).shape == (10, 1, 32, 32)
        assert model(x.cuda()).shape == (10, 1, 32, 32)
        assert model.sample(5).shape == (5, 1, 32, 32)
    for k, b in model.state_dict().items():  # eval mode must not touch the running statistics
        if "running_" in k or "num_batches" in k:
            assert torch.equal(b.cpu(), sd[k]), k
    # state_dict round trip into a fresh model (values land on HWIO memory)
    torch.manual_seed(6)
    m2 = A.VAE(encoder=A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add"),
               decoder=A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add"),
               prior=A.GaussianPrior(loss_coeff=0.1))
    m2.load_state_dict(model.state_dict())
    m2 = m2.cuda().eval()
    with torch.no_grad():
        rep.check("load_state_dict round trip", m2.encoder(x.cuda()), hg, tol=1e-7)
    # eval-mode backward (fixed affine BatchNorm) vs oracle
    xg = x.cuda().requires_grad_(True)
    hg2 = model.encoder(xg)
    hg2.sum().backward()
    xc = x.clone().requires_grad_(True)
    pe2 = {k: v.clone() for k, v in pe.items()}
    O.cnn_forward(xc, pe2, ea, training=False).sum().backward()
    rep.check("d encoder / d input (eval)", xg.grad, xc.grad, tol=3e-4)
    rep.finish()


def test_autoencoder_plugin(A):
    ae = A.AutoEncoder(1, 8, 16, 1, capacity=2, double_encoded_features=True, residual="add", down_up_sample=True)
    assert ae.latent_size == torch.Size((16, 1, 1))
    model = A.VAE(autoencoder=ae, prior=A.GaussianPrior(loss_coeff=0.5)).cuda().train()
    x = normal((4, 1, 16, 16), 9).cuda()
    loss, logs, art = model.nelbo({"samples": x, "target": x, "kwargs": {}}, 0)
    loss.backward()
    assert art["preds"].shape == x.shape and art["latents"].shape == (4, 8, 1, 1)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in ae.parameters())
    assert set(logs) == {"train/loss/total", "train/loss/recon", "train/loss/prior"}


def test_grouped_dilated_cnn_trains_through_the_captured_step(A):
    """A CNN whose layers are grouped (and one dilated ConvLayer in front of it) through ``HipTrainer``: the grouped parameters sit in
    the flat buffers with nn.Conv2d's shape, their gradients come back through the weight expansion (csrc/weight_expand.hip), and
    the captured step equals the eagerly issued one bit for bit over three Adam steps; the input gradient of the dilated layer
    matches torch's dilated convolution."""
    import copy
    import torch.nn.functional as F
    torch.manual_seed(3)
    enc = A.CNN(1, 16, 16, 1, capacity=4, down_sample=True, residual="add", groups=2)
    dec = A.CNN(8, 1, 1, 16, capacity=4, up_sample=True, residual="add", groups=2)
    assert any(m.groups == 2 and m.weight.shape[1] * 2 == m.in_channels for m in enc.modules() if isinstance(m, A.ConvLayer))
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    twin = copy.deepcopy(model)
    x, eps = normal((8, 1, 16, 16), 5).cuda(), normal((8, 8, 1, 1), 6).cuda()
    t_graph = A.HipTrainer(model, batch_shape=(8, 1, 16, 16), use_graph=True)
    t_eager = A.HipTrainer(twin, batch_shape=(8, 1, 16, 16), use_graph=False)
    for _ in range(3):
        a, b = t_graph.step(x, eps), t_eager.step(x, eps)
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    assert torch.equal(t_graph.pflat, t_eager.pflat)
    assert float((t_graph.pflat - t_graph.pflat.new_tensor(0)).abs().sum()) > 0
    t_graph.close(); t_eager.close()
    # a dilated, grouped layer against torch's own convolution
    layer = A.ConvLayer(6, 6, activation="relu", dilation=2, padding=2, groups=3).cuda()
    xi = normal((2, 6, 8, 8), 7).cuda().requires_grad_(True)
    y = layer(xi)
    y.square().sum().backward()
    w = layer.weight.detach().clone().requires_grad_(True)
    xr = xi.detach().clone().requires_grad_(True)
    yr = F.conv2d(F.relu(xr), w, layer.bias.detach(), padding=2, dilation=2, groups=3)
    yr.square().sum().backward()
    assert float((y.detach() - yr.detach()).abs().max() / yr.detach().abs().max()) < 1e-5
    assert float((xi.grad - xr.grad).abs().max() / xr.grad.abs().max()) < 1e-5
    assert float((layer.weight.grad - w.grad).abs().max() / w.grad.abs().max()) < 1e-5


def test_scaling_factor_4_cnn_trains_through_the_captured_step(A):
    """Round 4: `CNN(down_sample=4)` / `CNN(up_sample=4)` (8 x 8 kernels with stride 4, nn.Upsample(4) + 3 x 3: the reference's scaling
    factor 4, cnn.py:98-101,605-621) as a VAE through ``HipTrainer``: the direct-convolution fallback writes its weight gradients
    into the flat buffer, the captured step equals the eagerly issued one bit for bit over three Adam steps, and the loss falls."""
    import copy
    torch.manual_seed(4)
    enc = A.CNN(1, 32, 16, 1, capacity=4, down_sample=4, residual="add")
    dec = A.CNN(16, 1, 1, 16, capacity=4, up_sample=4, residual="add")
    assert any(m._generic and m.stride == (4, 4) and m.kernel_size == (8, 8) for m in enc.modules() if isinstance(m, A.ConvLayer))
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    twin = copy.deepcopy(model)
    x, eps = normal((16, 1, 16, 16), 15).cuda(), normal((16, 16, 1, 1), 16).cuda()
    t_graph = A.HipTrainer(model, batch_shape=(16, 1, 16, 16), use_graph=True)
    t_eager = A.HipTrainer(twin, batch_shape=(16, 1, 16, 16), use_graph=False)
    losses = []
    for _ in range(6):
        a, b = t_graph.step(x, eps), t_eager.step(x, eps)
        assert torch.equal(a, b) and bool(torch.isfinite(a).all())
        losses.append(float(a[0]))
    assert torch.equal(t_graph.pflat, t_eager.pflat) and losses[-1] < losses[0], losses
    t_graph.close(); t_eager.close()


def test_error_behaviour(A):
    assert A.ConvLayer(4, 4, kernel_size=5, dilation=2)._generic   # a 9 x 9 footprint: the direct-convolution fallback since round 4
    with pytest.raises(NotImplementedError):
        A.ConvLayer(4, 4, kernel_size=17, dilation=3)  # a 49 x 49 footprint: beyond the 32 x 32 taps of any kernel here
    # (round 4: `ema_decay` is built -- tests/test_gpu_lifetime.py::test_parameter_ema_... -- and no longer refused)
    m_ = A.VAE(encoder=A.CNN(1, 16, 16, 1, capacity=4, down_sample=True), decoder=A.CNN(8, 1, 1, 16, capacity=4, up_sample=True),
               prior=A.GaussianPrior(), ema_decay=0.999)
    assert m_.ema_decay == 0.999 and m_._ema is None   # created at on_fit_start / by the engine, as in the reference
    with pytest.raises(NotImplementedError):
        A.ConvLayer(4, 4, activation="tanh")           # not among the reference's activations either (cnn.py:147)
    with pytest.raises(NotImplementedError):
        A.ConvLayer(4, 4, normalization="whatever")
    with pytest.raises(ValueError):
        A.CNN(1, 8, 32, 32, down_sample=True)          # in_resolution <= out_resolution
    with pytest.raises(ValueError):
        A.CNN(1, 8, up_sample=True, down_sample=True, in_resolution=4, out_resolution=8)
    with pytest.raises(ValueError):
        A.AttentionBlock(6, heads=4)
    with pytest.raises(ValueError):
        A.VAE(prior=None)
    a = torch.ones(4, device="cuda") / 4
    with pytest.raises(ValueError):
        A.sinkhorn_log(a, a, torch.rand(3, 4, device="cuda"))
    with pytest.raises(NotImplementedError):
        A.w2_gaussian(torch.zeros(2100, device="cuda"), torch.zeros(2100, device="cuda"),
                      torch.eye(2100, device="cuda"), torch.eye(2100, device="cuda"))   # D > 2048: not implemented
    w = A.w2_gaussian(torch.zeros(200, device="cuda"), torch.ones(200, device="cuda"),
                      torch.eye(200, device="cuda"), torch.eye(200, device="cuda"))   # block-Jacobi path
    assert abs(float(w) - 200.0) < 1e-9
    # a module left on the host meeting GPU samples: refused at the pointer gate, never a GPU fault
    op = A.GaussianTransport(16, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double))
    with pytest.raises(RuntimeError, match="MI355X kernel"):
        op.update(source_samples=torch.randn(32, 16, device="cuda"), target_samples=torch.randn(32, 16, device="cuda"))
    # the reference's argument validation survives the one-call W2 + operator path: an indefinite "covariance" raises without
    # make_pd and is shifted with it; an asymmetric inner product raises either way
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    from ot_vae_lightning_amd.ot.w2_utils import w2_and_transport_operator
    g = torch.Generator().manual_seed(3)
    b = torch.randn(12, 12, generator=g, dtype=torch.double)
    spd = (b @ b.T / 12 + 0.1 * torch.eye(12, dtype=torch.double)).cuda()
    indef = spd - 0.5 * torch.eye(12, dtype=torch.double, device="cuda")
    zero = torch.zeros(12, dtype=torch.double, device="cuda")
    spec = lambda m: (m, *MU.eigh_vectors(m))  # noqa: E731
    w2, T, _ = w2_and_transport_operator(zero, zero + 1, spec(indef), spec(spd), make_pd=True)
    assert torch.isfinite(w2).all() and torch.isfinite(T).all()
    for bad_side in (0, 1):
        specs = [spec(spd), spec(spd)]
        specs[bad_side] = spec(indef)
        with pytest.raises(ValueError, match="cov_source" if bad_side == 0 else "cov_target"):
            w2_and_transport_operator(zero, zero, specs[0], specs[1], make_pd=False)
    conv = A.ConvLayer(4, 4, normalization="batchnorm", activation="relu")   # parameters on the host
    with pytest.raises(RuntimeError):
        conv(torch.zeros(2, 4, 8, 8, device="cuda"))


@pytest.mark.parametrize("D", [128, 256])
def test_config5_latent_transport_token_shape_vs_oracle(A, D):
    """BASELINE configs[4] / SURVEY 8(d) C5: ViT token latents [512, 1, D] (one token per image: leading shape (1,),
    samples laid out [tokens, batch, D] as `permute_and_flatten(batch_first=False)` gives them), a GaussianModel.update
    per training step for source and target, one GaussianTransport.compute at validation, transport of a batch."""
    rep = Report(f"config 5 latent transport, tokens [512, 1, {D}] vs CPU oracle")
    steps, B = 4, 512
    g = torch.Generator().manual_seed(45)
    src = [(torch.randn(1, B, D, generator=g) * 2 + 1) for _ in range(steps)]
    tgt = [torch.randn(1, B, D, generator=g) for _ in range(steps)]
    op = A.GaussianTransport(1, D, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double),
                             transport_cfg=dict(make_pd=True)).cuda()
    for s, t in zip(src, tgt):
        op.update(source_samples=s.cuda(), target_samples=t.cuda())
    w2 = op.compute()
    n = torch.zeros(1, dtype=torch.float64)
    acc = {k: [torch.zeros(1, dtype=torch.float64), torch.zeros(1, D, dtype=torch.float64),
               torch.zeros(1, D, D, dtype=torch.float64)] for k in ("s", "t")}
    for s, t in zip(src, tgt):
        for k, x in (("s", s), ("t", t)):
            ni, sx, sxx = O.gaussian_stats(x)
            acc[k][0] += ni
            acc[k][1] += sx
            acc[k][2] += sxx
    ms, cs = O.gaussian_fit(*acc["s"])
    mt, ct = O.gaussian_fit(*acc["t"])
    rep.check("W2^2", w2, O.w2_gaussian(ms, mt, cs, ct, make_pd=True), tol=1e-8)
    T = O.transport_operator_full(cs, ct)
    rep.check("transport operator", op.transport_operator, T, tol=1e-7)
    probe = src[0][:, :32]
    rep.check("transported tokens", op.transport(probe.cuda()), O.apply_transport(probe.double(), ms, mt, T).float(), tol=1e-5)
    rep.finish()


def test_latent_transport_callback_validation_epoch_vs_oracle(A):
    """The reference's tests/test_latent_transport.py:44-79 set-up for the Gaussian operator: AutoEncoder(1 -> 64x4x4,
    capacity 4, residual add), LatentTransport(GaussianTransport, transport_dims=(1, 2, 3)  =>  D = 1024, common operator,
    unpaired: even validation batches give the target, blurred odd ones the source).  The latents are stochastic as a
    VAE's are (z = encoder(x) + 0.5 eps), which also keeps the 1024x1024 covariances well conditioned.  One validation
    epoch driven through the callback hooks; W2^2, the operator and the transported latents against the CPU oracle on the
    same latents, plus the push-forward property (T Sigma_s T' = Sigma_t, transported mean = target mean) and the reset
    at the next epoch start."""
    import torch.nn.functional as F
    rep = Report("LatentTransport callback: validation epoch, D=1024 Gaussian operator vs CPU oracle")
    torch.manual_seed(3)
    ae = A.AutoEncoder(1, 64, 32, 4, capacity=4, double_encoded_features=False, down_up_sample=True, residual="add")
    vae = A.VAE(autoencoder=ae, prior=None).cuda().eval()

    class StochasticLatents:
        """module stand-in handed to the hooks: encode = VAE.encode + fixed-scale noise, every result recorded"""
        training, device = False, torch.device("cuda")

        def __init__(self):
            self.gen, self.record = torch.Generator(device="cuda").manual_seed(5), []

        def encode(self, x, **kw):
            z = vae.encode(x, **kw)
            z = z + 0.5 * torch.randn(z.shape, generator=self.gen, device=z.device)
            self.record.append(z.flatten(1).double().cpu())
            return z

        def decode(self, z, **kw):
            return vae.decode(z, **kw)

    model = StochasticLatents()
    kern = torch.tensor([1., 4., 6., 4., 1.])
    kern = (kern[:, None] * kern[None, :] / 256.0)[None, None].cuda()
    blur = lambda x: F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), kern)  # noqa: E731  (stands in for GaussianBlur(5))
    w2_cfg = dict(diag=False, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
    cb = A.LatentTransport(size=vae.latent_size, transport_operator=A.GaussianTransport, transport_dims=(1, 2, 3),
                           logging_prefix="gaussian", transport_cfg=w2_cfg, source_cfg=dict(dtype=torch.double),
                           target_cfg=dict(dtype=torch.double), transformations=blur, source_latents_from_train=False,
                           target_latents_from_train=False, unpaired=True, common_operator=True)
    assert tuple(vae.latent_size) == (64, 4, 4) and cb.dim == 1024
    cb.on_fit_start(None, model)
    cb.on_validation_epoch_start(None, model)
    with torch.no_grad():
        for i in range(12):
            cb.on_validation_batch_end(None, model, {"samples": mnist_like(512, seed=500 + i).cuda()}, None, i)
        cb.on_validation_epoch_end(None, model)
    lat = {"t": model.record[0::2], "s": model.record[1::2]}   # unpaired: even batches target, odd batches source
    cost = cb.logged["transport/gaussian/gaussian/avg_transport_cost"]
    fit = {}
    for k in ("s", "t"):
        acc = None
        for x in lat[k]:
            st = O.gaussian_stats(x)
            acc = list(st) if acc is None else [a + b for a, b in zip(acc, st)]
        fit[k] = O.gaussian_fit(*acc)
    (ms, cs), (mt, ct) = fit["s"], fit["t"]
    rep.check("avg_transport_cost = W2^2", cost, O.w2_gaussian(ms, mt, cs, ct, make_pd=True), tol=1e-8)
    op = cb.transport_operator
    T = O.transport_operator_full(cs, ct)
    rep.check("transport operator", op.transport_operator, T, tol=1e-7)
    x_src = torch.cat(lat["s"])[:256]
    z_src = x_src.float().reshape(-1, 64, 4, 4).cuda()
    with torch.no_grad():
        moved = cb.transport(z_src)
    assert moved.shape == z_src.shape
    rep.check("transported latents", moved.flatten(1), O.apply_transport(x_src, ms, mt, T).float(), tol=1e-5)
    Tg = op.transport_operator.double().cpu()
    # exact up to the 1e-8 I the operator adds to Sigma_s before its inverse square root (w2_utils.py:755): ~1e-8 / 0.25
    rep.check("push-forward: T Sigma_s T' = Sigma_t", Tg @ cs @ Tg.transpose(-1, -2), ct, tol=1e-6)
    all_src = torch.cat(lat["s"])
    with torch.no_grad():
        moved_all = torch.cat([cb.transport(all_src[j:j + 512].float().reshape(-1, 64, 4, 4).cuda()).flatten(1).double().cpu()
                               for j in range(0, all_src.shape[0], 512)])
    rep.check("mean of the transported source = target mean", moved_all.mean(0), mt, tol=1e-5)
    assert cb.sample(4, "target").shape == (4, 64, 4, 4)
    cb.on_validation_epoch_start(None, model)
    assert op.transport_operator is None and float(op.source_model._n_obs.sum()) == 0 and float(op.target_model._n_obs.sum()) == 0
    rep.finish()


def test_latent_transport_callback_with_gmm_and_discrete_operators(A):
    """The other two operators of the reference's tests/test_latent_transport.py:80-101 behind the same callback: a
    GMMTransport per latent needle (transport_dims=(1,), 10 components, diagonal) and a DiscreteTransport per channel map
    (transport_dims=(2, 3), soft 'mean' training mode at temperature 1e-2), one common operator each.  A validation epoch
    through the hooks, then the transported latents must be finite, have the latent shape, and land nearer (on average)
    to the target latents' mean than the source latents were."""
    import torch.nn.functional as F
    torch.manual_seed(4)
    ae = A.AutoEncoder(1, 64, 32, 4, capacity=4, double_encoded_features=False, down_up_sample=True, residual="add")
    vae = A.VAE(autoencoder=ae, prior=None).cuda().eval()
    kern = torch.tensor([1., 4., 6., 4., 1.])
    kern = (kern[:, None] * kern[None, :] / 256.0)[None, None].cuda()
    shift_blur = lambda x: F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), kern) * 0.5 - 0.7  # noqa: E731
    w2_cfg = dict(diag=True, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
    mix = dict(metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax", inference_mode="argmax")
    base = dict(update_decay=None, update_with_autograd=False, dtype=torch.double)
    common = dict(size=vae.latent_size, transformations=shift_blur, source_latents_from_train=False,
                  target_latents_from_train=False, unpaired=True, common_operator=True)
    cbs = [A.LatentTransport(transport_operator=A.GMMTransport, transport_dims=(1,), logging_prefix="gmm", transport_type="argmax",
                             transport_cfg=w2_cfg, source_cfg={**base, "mixture_cfg": {**mix, "n_components": 10}},
                             target_cfg={**base, "mixture_cfg": {**mix, "n_components": 10}}, **common),
           A.LatentTransport(transport_operator=A.DiscreteTransport, transport_dims=(2, 3), logging_prefix="discrete",
                             transport_type="mean",
                             source_cfg={**base, "mixture_cfg": {**mix, "training_mode": "mean", "n_components": 64, "temperature": 1e-2}},
                             target_cfg={**base, "mixture_cfg": {**mix, "training_mode": "mean", "n_components": 64, "temperature": 1e-2}},
                             **common)]
    assert cbs[0].dim == 64 and cbs[1].dim == 16
    with torch.no_grad():
        for cb in cbs:
            cb.on_fit_start(None, vae)
            cb.on_validation_epoch_start(None, vae)
        for i in range(6):
            x = mnist_like(256, seed=700 + i).cuda()
            for cb in cbs:
                cb.on_validation_batch_end(None, vae, {"samples": x}, None, i)
        x = mnist_like(128, seed=800).cuda()
        z_tgt, z_src = vae.encode(x), vae.encode(shift_blur(x))
        for cb in cbs:
            cb.on_validation_epoch_end(None, vae)
            cost = cb.logged[cb.logging_prefix + "avg_transport_cost"]
            assert torch.isfinite(cost) and float(cost) >= 0
            moved = cb.transport(z_src)
            assert moved.shape == z_src.shape and moved.dtype == z_src.dtype and torch.isfinite(moved).all()
            axes = (0, 2, 3) if cb.transport_dims == (1,) else (0, 1)     # what the common operator pools over
            before = (z_src.mean(axes) - z_tgt.mean(axes)).norm()
            after = (moved.mean(axes) - z_tgt.mean(axes)).norm()
            assert float(after) < 0.5 * float(before), (cb.logging_prefix, float(before), float(after))
            assert vae.decode(moved).shape == x.shape


def test_conditional_vit_vae_trains_through_hip_trainer(A):
    """The conditional ViT VAE (ViT encoder / decoder + ConditionalGaussianPrior, class labels as a resident batch keyword)
    through HipTrainer: gradients that arrive by plain autograd (learned tokens, embeddings, LayerNorm weights) are
    collected into the flat buffer next to the ones the kernels write there; the captured (hipGraph) step must give the
    bits of the eager one, and the loss must go down."""
    cfg = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.0, emb_dropout=0.,
               num_classes=10)
    x = normal((64, 3, 16, 16), 71).cuda()
    eps = [normal((64, 1, 32), 72 + i).cuda() for i in range(6)]
    y = (torch.arange(64) % 10).cuda()

    def run(graph):
        torch.manual_seed(5)
        enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
        dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
        prior = A.ConditionalGaussianPrior(dim=(1, 32), num_classes=10, loss_coeff=0.1)
        model = A.VAE(encoder=enc, decoder=dec, prior=prior, conditional=True).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(64, 3, 16, 16), use_graph=graph, batch_kwargs={"labels": y})
        losses = [tr.step(x, eps[i], labels=y).clone() for i in range(6)]
        torch.cuda.synchronize()
        return tr.pflat.clone(), torch.stack(losses)

    p_eager, l_eager = run(False)
    p_graph, l_graph = run(True)
    assert torch.equal(l_eager, l_graph) and torch.equal(p_eager, p_graph)
    assert float(l_eager[-1, 0]) < float(l_eager[0, 0]) and torch.isfinite(p_eager).all()
    with pytest.raises(KeyError):
        A.HipTrainer(A.VAE(encoder=A.CNN(1, 16, 16, 1, capacity=2, down_sample=True), decoder=A.CNN(8, 1, 1, 16, capacity=2, up_sample=True),
                           prior=A.GaussianPrior()).cuda(), batch_shape=(4, 1, 16, 16), use_graph=False).step(labels=y)


def test_config5_conditional_vit_vae_at_the_yaml_shape_vs_oracle(A):
    """BASELINE configs[4]'s NETWORK on the HIP path at the size the reference's configs/vae/vit.yaml:12-48 gives it
    (64 x 64 x 3 images, 8 x 8 patches -> 64 patch tokens, dim 256, 8 heads of width 32, encoder depth 3, decoder depth 2,
    mlp 4 x dim) with the class conditioning and ConditionalGaussianPrior of tests/test_conditional_vit_vae.py, 64 images per
    GPU (512 over 8): one captured training step's losses, reconstructions, latents and EVERY parameter gradient element-wise
    against the CPU oracle on the same weights / batch / eps / labels.  Dropout 0 (the oracle has no mask to share)."""
    from detfill import fill_vit_state_dict
    from test_oracle_vs_golden import VIT_ROLES, vit_param_shapes
    rep = Report("conditional ViT VAE at the vit.yaml shape (64x64, dim 256, heads 8, depth 3/2), B=64, vs CPU oracle")
    B, ncls = 64, 10
    base = dict(image_size=64, patch_size=8, dim=256, heads=8, mlp_dim=1024, channels=3, num_classes=ncls)
    cfgs = {"enc": dict(depth=3, **base), "dec": dict(depth=2, **base)}
    x = normal((B, 3, 64, 64), 151)
    eps = normal((B, 1, 256), 152)
    labels = torch.arange(B) % ncls
    g = torch.Generator().manual_seed(153)
    mu_w, ls_w = torch.randn(ncls, 256, generator=g) * 0.3, torch.randn(ncls, 256, generator=g) * 0.1
    # oracle
    pcpu = {}
    for role in ("enc", "dec"):
        sd = {k: torch.zeros(sh) for k, sh in vit_param_shapes(cfgs[role], role).items()}
        fill_vit_state_dict(sd)
        pcpu[role] = {k: v.requires_grad_(True) for k, v in sd.items()}
    mw, lw = mu_w.clone().requires_grad_(True), ls_w.clone().requires_grad_(True)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    vit = lambda role: dict(image_size=64, patch_size=8, dim=256, depth=cfgs[role]["depth"], heads=8, channels=3, labels=labels,  # noqa: E731
                            **VIT_ROLES[role])
    h = O.vit_forward(x, pcpu["enc"], **vit("enc"))
    z, prior_l = O.cond_gaussian_prior_encode(h, eps, mw, lw, labels, 0.1, 0, 0)
    preds = O.vit_forward(z, pcpu["dec"], **vit("dec"))
    prior_loss = prior_l.mean() / float(x[0].numel())
    recon = torch.nn.functional.mse_loss(preds, x)
    (recon + prior_loss).backward()
    # the same step in float64: the truth both fp32 sides are measured against
    p64 = {role: {k: v.detach().double().requires_grad_(True) for k, v in pcpu[role].items()} for role in ("enc", "dec")}
    mw64, lw64 = mu_w.double().requires_grad_(True), ls_w.double().requires_grad_(True)
    h64 = O.vit_forward(x.double(), p64["enc"], **vit("enc"))
    z64, prior_l64 = O.cond_gaussian_prior_encode(h64, eps.double(), mw64, lw64, labels, 0.1, 0, 0)
    preds64 = O.vit_forward(z64, p64["dec"], **vit("dec"))
    (torch.nn.functional.mse_loss(preds64, x.double()) + prior_l64.mean() / float(x[0].numel())).backward()
    # product: the captured step
    nets = {}
    for role in ("enc", "dec"):
        net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., **cfgs[role], **VIT_ROLES[role])
        net.load_state_dict({k: v.detach() for k, v in pcpu[role].items()})
        nets[role] = net
    prior = A.ConditionalGaussianPrior(dim=(1, 256), num_classes=ncls, loss_coeff=0.1)
    with torch.no_grad():
        prior._mu.weight.copy_(mu_w)
        prior._log_std.weight.copy_(ls_w)
    model = A.VAE(encoder=nets["enc"], decoder=nets["dec"], prior=prior, conditional=True).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(B, 3, 64, 64), use_graph=True, batch_kwargs={"labels": labels.cuda()})
    out = tr.step(x.cuda(), eps.cuda(), labels=labels.cuda()).clone()
    torch.cuda.synchronize()
    rep.check("loss [total, recon, prior]", out, torch.stack([recon + prior_loss, recon, prior_loss]).detach(), 1e-5)
    rep.check("latents", tr.latents, z.detach(), 1e-5)
    names, got, want, truth = [], [], [], []
    for pre, net, ref, r64 in (("encoder.", model.encoder, pcpu["enc"], p64["enc"]), ("decoder.", model.decoder, pcpu["dec"], p64["dec"])):
        for k, p_ in net.named_parameters():
            if ref[k].grad is None:
                continue
            names.append(pre + k)
            got.append(p_._otvae_grad_view())
            want.append(ref[k].grad)
            truth.append(r64[k].grad)
    for k, p_, ref, r64 in (("prior._mu.weight", prior._mu.weight, mw, mw64), ("prior._log_std.weight", prior._log_std.weight, lw, lw64)):
        names.append(k)
        got.append(p_._otvae_grad_view())
        want.append(ref.grad)
        truth.append(r64.grad)
    # Round 3 bounded this at 8e-3 / 1e-2 because ONE tensor -- the LayerNorm in front of the decoder's transformer, whose parameter
    # gradients sum 64 x 65 x 256 products in fp32 on both sides -- sat 4.7e-3 from the fp32 oracle.  Now both fp32 sides are held to
    # the float64 truth: every tensor's bound is the standing 2e-3 / 5e-3 or 1.5x the reference arithmetic's own fp32 error on it
    # (The evidence-based bound found a real defect here: the two parameters of that one LayerNorm sat 4.5e-3 / 2.6e-3 from the truth
    # where the reference's fp32 arithmetic sits 2.9e-4 / 1.0e-3.  Cause: the attention backward took delta = gout . out from the
    # forward pass's output, which is not consistent with the p and dP the backward recomputes when a softmax row is peaked
    # (tools/diag/vit_ln_grad.py, profiles/r04_vit_attention_delta.txt); with a consistent delta (csrc/attention.hip, bare kernel)
    # they sit 2.8e-4 / 9.1e-4 and every tensor of the step holds the rule.)
    rep.check_grads_vs_truth("gradients (captured step)", got, want, truth, names)
    tr.close()
    rep.finish()


def test_vit_vae_with_dropout_draws_fresh_masks_in_a_captured_step(A):
    """The reference's ViT configuration trains with dropout 0.1 (configs/vae/vit.yaml): token dropouts between the
    kernels, attention-probability dropout inside ``otvae_attn_dropout_*``.  Through the captured HipTrainer step every
    replay must draw new masks (the mask key's counter lives in device memory and is advanced by a captured add), so the
    same batch gives different losses step after step; in eval mode nothing is dropped and the output is deterministic."""
    cfg = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.1, emb_dropout=0.1)
    x = normal((64, 3, 16, 16), 81).cuda()
    eps = normal((64, 1, 32), 82).cuda()
    torch.manual_seed(6)
    enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
    dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(64, 3, 16, 16), use_graph=True, lr=0.0)      # lr 0: only the masks differ between steps
    losses = torch.stack([tr.step(x, eps).clone() for _ in range(5)])[:, 1]           # reconstruction term
    torch.cuda.synchronize()
    assert torch.isfinite(losses).all()
    assert len({float(v) for v in losses}) == 5, losses                                # five replays, five different masks
    assert float(losses.max() - losses.min()) < 0.2 * float(losses.mean())             # ... of the same model on the same batch
    key = enc.__dict__["_dropout_key"]
    assert int(key[1]) >= 5 and key.dtype == torch.int64
    model.eval()
    with torch.no_grad():
        a, b = model.encoder(x), model.encoder(x)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_cross_attention_vit_vae_trains_through_hip_trainer(A, dropout):
    """A ViT VAE whose encoder and decoder are the cross-attention variant (``preprocess_depth``: nn.TransformerDecoder layers,
    reference networks/vit.py:171-181) through the engine: the captured step equals the eagerly issued one bit for bit (dropout
    0), the loss goes down, and with the reference's training dropout the replays draw fresh masks."""
    vcfg = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=dropout, emb_dropout=0.)
    x = normal((64, 3, 16, 16), 61).cuda()
    eps = normal((64, 1, 32), 62).cuda()

    def run(graph, steps, lr=1e-3):
        torch.manual_seed(3)
        enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False,
                    preprocess_depth=1, **vcfg)
        dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True,
                    preprocess_depth=0, **vcfg)
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(64, 3, 16, 16), use_graph=graph, data_parallel=False, lr=lr)
        losses = torch.stack([tr.step(x, eps).clone() for _ in range(steps)])
        torch.cuda.synchronize()
        out = tr.pflat.clone(), losses
        tr.close()
        return out

    if dropout == 0.0:
        pe, le = run(False, 6)
        pg, lg = run(True, 6)
        assert torch.equal(pe, pg) and torch.equal(le, lg)
        assert torch.isfinite(le).all() and float(le[-1, 0]) < float(le[0, 0])
    else:
        _, losses = run(True, 5, lr=0.0)
        rec = losses[:, 1]
        assert torch.isfinite(rec).all() and len({float(v) for v in rec}) == 5, rec     # five replays, five masks


def test_dp_overlap_two_phase_backward_equals_single_phase(A):
    """Data-parallel overlap path (backward cut at the encoder output, decoder gradients all-reduced under the encoder's
    backward, three captured graphs) rehearsed on ONE GPU with a 1-rank RCCL process group: parameters, moments and
    losses must be identical bits to the single-graph path, eager and captured; with global-norm clipping too.  Runs
    IN this process and tears the communicator down explicitly (HipTrainer.close() then destroy_process_group): round 1's
    abort here was the RCCL watchdog's event poll landing in a global-mode stream capture, which
    capture_error_mode="thread_local" removed."""
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    x = [mnist_like(64, 90 + i).cuda() for i in range(3)]
    eps = [normal((64, 128, 1, 1), 95 + i).cuda() for i in range(3)]
    trainers = []

    def run(overlap, graph, clip=None):
        torch.manual_seed(11)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(64, 1, 32, 32), use_graph=graph, dp_overlap=overlap, gradient_clip_val=clip)
        assert tr.dp_overlap == overlap
        losses = [tr.step(x[i], eps[i]).clone() for i in range(3)]
        torch.cuda.synchronize()
        trainers.append(tr)
        return tr.pflat.clone(), tr.m.clone(), tr.v.clone(), torch.stack(losses)

    # a model whose decoder also has plain-autograd ("loose") gradients -- LayerNorm weights, learned tokens of the ViT:
    # phase 2 must not touch the decoder's slots, which are under their all-reduce by then
    vcfg = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.0, emb_dropout=0.)
    xv = normal((64, 3, 16, 16), 71).cuda()
    epsv = [normal((64, 1, 32), 72 + i).cuda() for i in range(3)]

    def run_vit(overlap, graph):
        torch.manual_seed(5)
        enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **vcfg)
        dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **vcfg)
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(64, 3, 16, 16), use_graph=graph, dp_overlap=overlap)
        assert tr.dp_overlap == overlap
        losses = [tr.step(xv, epsv[i]).clone() for i in range(3)]
        torch.cuda.synchronize()
        trainers.append(tr)
        return tr.pflat.clone(), tr.m.clone(), tr.v.clone(), torch.stack(losses)

    ref = run(False, False)
    ref_vit = run_vit(False, False)
    ref_clip = run(False, False, clip=0.05)
    assert not torch.equal(ref[0], ref_clip[0])            # the clip is active at this threshold
    port = 29617 + os.getpid() % 300
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        for graph in (False, True):
            got = run(True, graph)
            for g, r, name in zip(got, ref, ("params", "m", "v", "losses")):
                assert torch.equal(g, r), (graph, name, float((g - r).abs().max()))
        plain = run(False, True)  # the two-graph path with the collective in between
        for g, r in zip(plain, ref):
            assert torch.equal(g, r)
        got = run(True, True, clip=0.05)
        for g, r in zip(got, ref_clip):
            assert torch.equal(g, r)
        for graph in (False, True):
            got = run_vit(True, graph)
            for g, r, name in zip(got, ref_vit, ("params", "m", "v", "losses")):
                assert torch.equal(g, r), ("vit", graph, name, float((g - r).abs().max()))
    finally:
        for tr in trainers:
            tr.close()
        dist.destroy_process_group()
    assert not dist.is_initialized()


def test_side_stream_weight_gradients_equal_single_stream(A, monkeypatch):
    """The weight-gradient jobs of the backward pass run on a second HIP stream (functional._PendingReduce: forked per backward
    call, partial reductions on the side stream, joined before the optimizer).  Same bits as the single-stream order -- eagerly
    issued with the side stream forced on, captured (where it is on by default), and with every fork shared by three calls."""
    from ot_vae_lightning_amd import functional as HF
    x = [mnist_like(64, 40 + i).cuda() for i in range(3)]
    eps = [normal((64, 128, 1, 1), 45 + i).cuda() for i in range(3)]

    def run(mode, graph, group=1, reduce_group=16):
        monkeypatch.setattr(HF, "WGRAD_SIDE_STREAM", mode)
        monkeypatch.setattr(HF, "WGRAD_GROUP", group)
        monkeypatch.setattr(HF, "WGRAD_REDUCE_GROUP", reduce_group)
        torch.manual_seed(11)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(64, 1, 32, 32), use_graph=graph, data_parallel=False)
        losses = [tr.step(x[i], eps[i]).clone() for i in range(3)]
        torch.cuda.synchronize()
        out = tr.pflat.clone(), tr.m.clone(), tr.v.clone(), torch.stack(losses)
        tr.close()
        return out

    ref = run(0, False)
    HF._PendingReduce._side.clear()
    for mode, graph, group, rg in ((2, False, 1, 16), (2, False, 3, 4), (1, True, 1, 16), (2, True, 2, 1000), (0, True, 1, 16)):
        got = run(mode, graph, group, rg)
        for g, r, name in zip(got, ref, ("params", "m", "v", "losses")):
            assert torch.equal(g, r), (mode, graph, group, rg, name, float((g - r).abs().max()))
    assert HF._PendingReduce._side, "the side stream was never used"
    dev = next(iter(HF._PendingReduce._state))
    assert not HF._PendingReduce._state[dev] and not HF._PendingReduce._held.get(dev) and not HF._PendingReduce._wq[dev][0]


@pytest.mark.parametrize("prior_kind", ["sinkhorn", "gaussian_w2"])
def test_prior_lane_equals_in_line_order(A, prior_kind, monkeypatch):
    """inside a captured step the prior's optimal-transport work (Sinkhorn solve / statistics + eigh + W2 tail) and
    the loss vector run on a stream of their own beside the decoder (functional.PriorLane).  Same kernels, same inputs: losses,
    gradients and parameters after three replays must be bit-identical to the in-line order (OTVAE_PRIOR_STREAM=0)."""
    from ot_vae_lightning_amd import functional as HF
    B = 256
    xs = [mnist_like(B, 90 + i).cuda() for i in range(3)]

    def run(mode):
        monkeypatch.setattr(HF, "PRIOR_SIDE_STREAM", mode)
        torch.manual_seed(11)
        enc = A.CNN(1, 128, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, seed=7) if prior_kind == "sinkhorn" else A.GaussianW2Prior(loss_coeff=0.1)
        model = A.VAE(encoder=enc, decoder=dec, prior=prior).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=True)
        outs = [tr.step(x).clone() for x in xs]
        torch.cuda.synchronize()
        res = (torch.stack(outs), tr.gflat.clone(), tr.pflat.clone())
        assert tr.skipped_steps == 0 and torch.isfinite(res[0]).all()
        tr.close()
        return res

    lane, inline = run(1), run(0)
    for name, a, b in zip(("losses", "gradients", "parameters"), lane, inline):
        assert torch.equal(a, b), f"{prior_kind}: {name} differ between the prior lane and the in-line order"
    assert not HF.PriorLane.is_open(torch.device("cuda", torch.cuda.current_device()))


def test_graphed_nelbo_gives_the_unmodified_loop_the_graph_route(A):
    """``training_step`` -> ``loss.backward()`` -> a stock ``torch.optim.Adam`` (the reference's loop: model/base.py:
    122-129, model/vae.py:148-156) with ``model.enable_graphed_step()``: the loss vector and every gradient must be BIT-equal
    to ``HipTrainer``'s for the same weights / batch / eps over several steps, ``p.grad`` must be populated for any optimizer, the
    parameters after stock Adam agree with the fused Adam kernel to rounding, BatchNorm's running buffers advance alike, and an
    evaluation call goes through the plain ``nelbo``."""
    B = 64
    xs = [mnist_like(B, 110 + i).cuda() for i in range(3)]
    es = [normal((B, 128, 1, 1), 120 + i).cuda() for i in range(3)]

    def make():
        torch.manual_seed(21)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()

    ref_model = make()
    tr = A.HipTrainer(ref_model, batch_shape=(B, 1, 32, 32), use_graph=True)
    model = make().enable_graphed_step()
    opt = torch.optim.Adam(model.optim_parameters(), lr=1e-3, betas=(0.9, 0.999))
    for i in range(3):
        want = tr.step(xs[i], es[i]).clone()
        want_g = tr.gflat.clone()
        opt.zero_grad()
        # the reference's training_step, called the way Lightning does (batch_preprocess builds {samples, target, kwargs})
        model.batch_preprocess = lambda b: {"samples": b[0], "target": b[0], "kwargs": {"eps": b[1]}}
        out = model.training_step((xs[i], es[i]), i)
        out["loss"].backward()
        got = torch.stack([out["train/loss/total"], out["train/loss/recon"], out["train/loss/prior"]]).detach()
        assert torch.equal(got, want), (i, got, want)
        eng = model.loss._cap.engine
        assert torch.equal(eng.gflat, want_g), f"step {i}: gradients differ from HipTrainer's"
        assert all(p.grad is not None and p.grad.data_ptr() == p._otvae_grad_view().data_ptr() for p in eng.params)
        assert out["preds"].shape == xs[i].shape and out["latents"].shape == (B, 128, 1, 1)
        opt.step()
        err = float((eng.pflat - tr.pflat).abs().max())
        assert err < 5e-6, (i, err)     # stock Adam vs the fused kernel: same update to rounding (a step moves a weight by ~1e-3)
        with torch.no_grad():           # the next step's bit comparison starts from identical weights again
            eng.pflat.copy_(tr.pflat)
    for (ka, va), (kb, vb) in zip(model.state_dict().items(), ref_model.state_dict().items()):
        if "running" in ka or "num_batches" in ka:
            assert torch.allclose(va.float(), vb.float(), rtol=1e-5, atol=1e-6), ka
    # no graph outside training: evaluation runs the plain nelbo (BatchNorm in inference mode)
    model.eval()
    with torch.no_grad():
        loss, logs, art = model.loss({"samples": xs[0], "target": xs[0], "kwargs": {"eps": es[0]}}, 0)
    assert torch.isfinite(loss) and art["preds"].shape == xs[0].shape
    tr.close()


def test_graphed_nelbo_with_a_conditional_vit_vae(A):
    """The graph route of the reference's own loop for configs[4]'s model: a conditional ViT VAE (class tokens, ConditionalGaussianPrior)
    whose `labels` travel as a batch keyword -- `enable_graphed_step()` must replay them from resident copies: loss vector and gradients
    bit-equal to HipTrainer's for the same weights, batch, labels and noise over three steps with changing labels."""
    torch.manual_seed(4)
    B, D = 32, 32
    cfg = dict(image_size=16, patch_size=4, dim=D, depth=1, heads=4, mlp_dim=64, channels=3, dropout=0., emb_dropout=0., num_classes=10)

    def make():
        torch.manual_seed(9)
        enc = A.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
        dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
        prior = A.ConditionalGaussianPrior(dim=(1, D), num_classes=10, loss_coeff=0.1)
        return A.VAE(encoder=enc, decoder=dec, prior=prior, conditional=True).cuda().train()

    xs = [normal((B, 3, 16, 16), 40 + i).cuda() for i in range(3)]
    es = [normal((B, 1, D), 50 + i).cuda() for i in range(3)]
    ys = [((torch.arange(B) * (i + 3)) % 10).cuda() for i in range(3)]
    tr = A.HipTrainer(make(), batch_shape=(B, 3, 16, 16), use_graph=True, batch_kwargs={"labels": ys[0]})
    model = make().enable_graphed_step()
    for i in range(3):
        want = tr.step(xs[i], es[i], labels=ys[i]).clone()
        want_g = tr.gflat.clone()
        for p in model.parameters():
            p.grad = None
        loss, logs, art = model.loss({"samples": xs[i], "target": xs[i], "kwargs": {"eps": es[i], "labels": ys[i]}}, i)
        loss.backward()
        got = torch.stack([logs["train/loss/total"], logs["train/loss/recon"], logs["train/loss/prior"]]).detach()
        assert torch.equal(got, want), (i, got, want)
        eng = model.loss._cap.engine
        assert torch.equal(eng.gflat, want_g), f"step {i}: gradients differ from HipTrainer's"
        with torch.no_grad():
            eng.pflat.copy_(tr.pflat)
    tr.close()


@pytest.mark.parametrize("prior_kind", ["gaussian", "sinkhorn"])
def test_segmented_capture_equals_forked_capture(A, prior_kind, monkeypatch):
    """the captured step as a chain of linear hipGraphs + side graphs ordered by events between graph launches
    (engine/segments.py) against the single graph with one fork per layer (round 2) and against the eagerly issued step: same
    kernels, same order per stream -> bit-identical losses, gradients and parameters over several steps (different segment
    lengths included; with the Sinkhorn prior the first cut has to wait for the prior lane's join)."""
    from ot_vae_lightning_amd.engine import segments, trainer as trainer_mod
    B = 128
    xs = [mnist_like(B, 130 + i).cuda() for i in range(3)]
    es = [normal((B, 128, 1, 1), 140 + i).cuda() for i in range(3)]

    def run(seg_calls, graph=True):
        monkeypatch.setattr(segments, "SEGMENT_CALLS", seg_calls)
        monkeypatch.setattr(trainer_mod, "SEGMENT_CALLS", seg_calls)
        torch.manual_seed(31)
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        if prior_kind == "gaussian":
            enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
            prior = A.GaussianPrior(loss_coeff=0.1)
        else:
            enc = A.CNN(1, 128, 32, 1, capacity=8, down_sample=True, residual="add")
            prior = A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, seed=9)
        model = A.VAE(encoder=enc, decoder=dec, prior=prior).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=graph)
        outs = [tr.step(x, e).clone() for x, e in zip(xs, es)]
        torch.cuda.synchronize()
        if graph:
            assert (tr._segments is not None) == (seg_calls > 0)
            if seg_calls > 0:
                assert len(tr._segments.main) >= (3 if seg_calls <= 6 else 2) and any(g is not None for g in tr._segments.side)
        res = (torch.stack(outs), tr.gflat.clone(), tr.pflat.clone())
        tr.close()
        return res

    forked = run(0)
    for calls in (6, 2, 50):
        seg = run(calls)
        for name, a, b in zip(("losses", "gradients", "parameters"), seg, forked):
            assert torch.equal(a, b), f"{prior_kind}: {name} differ between {calls}-call segments and the forked graph"
    if prior_kind == "gaussian":  # (the Sinkhorn prior draws its samples from a device counter that the capture's warm-up advances)
        eager = run(6, graph=False)
        for name, a, b in zip(("losses", "gradients", "parameters"), eager, forked):
            assert torch.equal(a, b), f"{prior_kind}: {name} differ between the eager step and the captured one"


def test_prior_lane_with_a_vit_autoencoder_equals_in_line_order(A, monkeypatch):
    """the same lane with token networks around it (ViT encoder / decoder, reconstruction re-laid out before the loss kernel):
    SinkhornPrior on the latent token, captured step, lane vs in-line order bit for bit"""
    from ot_vae_lightning_amd import functional as HF
    cfg = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.0, emb_dropout=0.)
    xs = [normal((64, 3, 16, 16), 171 + i).cuda() for i in range(3)]

    def run(mode):
        monkeypatch.setattr(HF, "PRIOR_SIDE_STREAM", mode)
        torch.manual_seed(15)
        enc = A.ViT(n_embed_tokens=1, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
        dec = A.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
        model = A.VAE(encoder=enc, decoder=dec, prior=A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0, seed=3)).cuda().train()
        tr = A.HipTrainer(model, batch_shape=(64, 3, 16, 16), use_graph=True)
        outs = torch.stack([tr.step(x).clone() for x in xs])
        torch.cuda.synchronize()
        res = (outs, tr.gflat.clone(), tr.pflat.clone())
        tr.close()
        return res

    lane, inline = run(1), run(0)
    assert torch.isfinite(lane[0]).all() and float(lane[0][:, 2].min()) > 0
    for name, a, b in zip(("losses", "gradients", "parameters"), lane, inline):
        assert torch.equal(a, b), f"{name} differ between the prior lane and the in-line order"


def test_gradient_clipping_matches_clip_grad_norm(A):
    """Global-norm clipping of the step (reference configs/ddp.yaml:4 -> Lightning -> torch.nn.utils.clip_grad_norm_): the
    norm the kernel reports, the coefficient and the clipped Adam update against torch arithmetic on the same gradient."""
    x, eps = mnist_like(32, 61).cuda(), normal((32, 128, 1, 1), 62).cuda()

    def make(clip):
        torch.manual_seed(3)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        return A.HipTrainer(model, batch_shape=(32, 1, 32, 32), use_graph=False, gradient_clip_val=clip)

    free = make(None)
    p0 = free.pflat.clone()
    free.step(x, eps)
    g = free.gflat.clone()
    norm = float(g.double().norm())
    for clip in (0.25 * norm, 4.0 * norm):                     # active / inactive
        tr = make(clip)
        tr.step(x, eps)
        assert torch.equal(tr.gflat, g)
        coef = min(1.0, clip / (norm + 1e-6))
        got = tr.clip_out.tolist()
        assert abs(got[1] - norm) <= 1e-6 * norm and abs(got[0] - coef) <= 2e-6 * coef, (got, norm, coef)
        ref_p = p0.clone().requires_grad_(True)
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], clip)
        opt = torch.optim.Adam([ref_p], lr=1e-3)
        opt.step()
        err = float((tr.pflat - ref_p.detach()).abs().max())
        assert err < 2e-7, (clip, err)                          # one Adam step moves a weight by <= lr = 1e-3
    assert float((make(0.25 * norm).pflat - p0).abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["loss-graph", "loss-eager", "full-graph"])
def test_step_guard_skips_a_bad_step_and_leaves_no_trace(A, mode):
    """VERDICT r2 #5 / ADVICE r2: a bad step must be survivable inside a captured step.  Trainer A sees [good, NaN batch, good],
    trainer B only the two good batches: parameters, Adam moments, the step counter and every BatchNorm running buffer must be
    BIT-identical afterwards, and A reports one skipped step.  Without the guard the NaN batch lands in every parameter."""
    guard, graph = mode.split("-")
    xs = [mnist_like(32, 71).cuda(), mnist_like(32, 72).cuda()]
    es = [normal((32, 128, 1, 1), 73).cuda(), normal((32, 128, 1, 1), 74).cuda()]
    bad = xs[0].clone()
    bad[3, 0, 5, 7] = float("nan")

    def make(g):
        torch.manual_seed(5)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        return A.HipTrainer(model, batch_shape=(32, 1, 32, 32), use_graph=graph == "graph", step_guard=g)

    ta, tb = make(guard), make(guard)
    ta.step(xs[0], es[0])
    out_bad = ta.step(bad, es[1]).clone()
    ta.step(xs[1], es[1])
    tb.step(xs[0], es[0])
    tb.step(xs[1], es[1])
    torch.cuda.synchronize()
    assert torch.isnan(out_bad[0]), "the bad step's loss is what the host sees"
    assert ta.skipped_steps == 1 and tb.skipped_steps == 0 and int(ta.guard[1]) == 2
    assert int(ta.step_count) == int(tb.step_count) == 2
    for name in ("pflat", "m", "v"):
        assert torch.equal(getattr(ta, name), getattr(tb, name)), name
    for (ka, va), (kb, vb) in zip(ta.model.state_dict().items(), tb.model.state_dict().items()):
        if "num_batches_tracked" in ka:
            continue  # counts forward passes, like torch's (the skipped step did run its forward pass)
        assert torch.equal(va, vb), ka
    # unguarded: the same batch poisons the parameters (what round 2 shipped)
    tc = make(None)
    tc.step(bad, es[1])
    assert torch.isnan(tc.pflat).any()
    for t in (ta, tb, tc):
        t.close()


@pytest.mark.parametrize("graph", [True, False])
def test_step_guard_covers_latent_statistics_and_the_w2_prior_warm_start(A, graph):
    """a refused step must leave no trace in ANYTHING a step mutates -- the running sums of the latent operator
    (``HipTrainer(latent_stats=...)``: n, sum x, sum x x^T in fp64), GaussianW2Prior's warm-start basis and flag, BatchNorm's counters.
    Trainer A sees [good, good, NaN batch, good], trainer B the three good batches: everything bit-identical afterwards."""
    B = 192
    xs = [mnist_like(B, 171 + i).cuda() for i in range(3)]
    bad = xs[0].clone()
    bad[5, 0, 9, 9] = float("nan")

    def make():
        torch.manual_seed(8)
        enc = A.CNN(1, 64, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(64, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianW2Prior(loss_coeff=0.1)).cuda().train()
        op = A.GaussianTransport(64, source_cfg=dict(dtype=torch.double, reduce_on_update=False),
                                 target_cfg=dict(dtype=torch.double, reduce_on_update=False)).cuda()
        return A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=graph, latent_stats=op), op

    (ta, opa), (tb, opb) = make(), make()
    ta.step(xs[0]); ta.step(xs[1])
    out_bad = ta.step(bad).clone()
    ta.step(xs[2])
    for x in xs:
        tb.step(x)
    torch.cuda.synchronize()
    assert torch.isnan(out_bad[0]) and ta.skipped_steps == 1 and tb.skipped_steps == 0
    assert int(ta.step_count) == int(tb.step_count) == 3
    for name in ("pflat", "m", "v"):
        assert torch.equal(getattr(ta, name), getattr(tb, name)), name
    for (ka, va), (kb, vb) in zip(ta.model.state_dict().items(), tb.model.state_dict().items()):
        assert torch.equal(va, vb), ka                       # incl. num_batches_tracked now
    pa, pb = ta.model.prior, tb.model.prior
    assert int(pa._warm) == int(pb._warm) == 1 and torch.equal(pa._v_prev, pb._v_prev) and torch.isfinite(pa._v_prev).all()
    sa, sb = dict(opa.named_buffers()), dict(opb.named_buffers())
    assert sa.keys() == sb.keys() and any("_running_sum_cov" in k for k in sa)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
        assert torch.isfinite(sa[k].double()).all(), k
    n_obs = [v for k, v in sa.items() if k.endswith("target_model._n_obs")][0]
    assert float(n_obs.sum()) == 3 * B
    ta.close(); tb.close()


def test_step_guard_survives_a_starved_sinkhorn_solve_in_a_captured_step(A, monkeypatch):
    """The persistent Sinkhorn solver's workgroups wait for each other; starved (poll budget 0, baked into the captured step)
    every replay ends in NaN-poisoned plan / loss: the guarded Adam must leave the parameters untouched and count the steps."""
    from ot_vae_lightning_amd.ot import w2_utils as W
    monkeypatch.setenv("OTVAE_SK_SPIN_LIMIT", "0")
    torch.manual_seed(6)
    enc = A.CNN(1, 128, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.SinkhornPrior(reg=0.05, max_iter=50, threshold=0.0)).cuda().train()
    tr = A.HipTrainer(model, batch_shape=(128, 1, 32, 32), use_graph=True)
    p0 = tr.pflat.clone()
    for i in range(3):
        out = tr.step(mnist_like(128, 80 + i).cuda())
    torch.cuda.synchronize()
    assert torch.isnan(out[0]) and tr.skipped_steps == 3 and int(tr.step_count) == 0
    assert torch.equal(tr.pflat, p0) and not torch.isnan(tr.m).any()
    with pytest.raises(W.SinkhornSolverStarved):
        tr.close()


def test_bench_two_rank_call_sequence_on_one_gpu():
    """``bench.py --gpus 2`` exactly as the driver launches it (torch.distributed.run, one process per rank), rehearsed on
    this box's single GPU: both ranks use cuda:0 and the collectives go over gloo (OTVAE_BENCH_SHARE_GPU=1).  Checks the
    N > 1 call sequence end to end -- rendezvous from the environment, replica broadcast, the three captured graphs with
    the gradient all-reduce in between, max-over-ranks timing, ONE JSON line from rank 0, exit code 0 -- not performance."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OTVAE_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6",
                        "--warmup", "3"], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 3 and d["scaling"] == "weak" and d["unit"] == "images/s"
    assert d["config"]["global_batch"] == 2048 and "dp2" in d["config"]["parallelism"] and "rehearsal" in d
    assert d["value"] > 0 and abs(d["value"] - 2048 * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    assert all(0 < v < 10 for v in d["final_loss"])
    for key in ("roofline", "roofline_issue"):
        assert key in d
