"""Timings of the validation-epoch OT paths (SURVEY.md section 8 f rows) on one MI355X, each next to the same arithmetic on
the box's host cores with torch's CPU ops (LAPACK eigh etc.) -- the way the reference runs them when no GPU is present.
Sizes are the reference's: latent statistics of BASELINE configs[1] (D = 128) and of tests/test_latent_transport.py
(D = 64*4*4 = 1024, GMM with 10 components on 64-dim needles, codebook of 1024 atoms on 16-dim channels).
Sanity numbers for the f rows, not the headline metric.  Usage: python tests/ot_paths_timing.py [--no-cpu]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ot_vae_lightning_amd as A  # noqa: E402

CPU = "--no-cpu" not in sys.argv
if CPU:
    import otvae_oracle as O  # the CPU restatement: timing context only


def gpu_ms(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def cpu_ms(fn, reps=2):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def line(name, g, c=None):
    print(f"{name:78s} GPU {g:9.3f} ms" + (f"   CPU {c:10.2f} ms   x{c / g:7.1f}" if c is not None else ""), flush=True)


def gaussian(D, B):
    g = torch.Generator().manual_seed(9)
    mix = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    src = torch.randn(B, D, generator=g, dtype=torch.float64) @ mix * 1.5 + 0.3
    tgt = torch.randn(B, D, generator=g, dtype=torch.float64)
    op = A.GaussianTransport(D, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double),
                             transport_cfg=dict(make_pd=True)).cuda()
    sg, tg = src.cuda(), tgt.cuda()

    def update():
        op.update(source_samples=sg, target_samples=tg)

    c = None
    if CPU:
        c = cpu_ms(lambda: (O.gaussian_stats(src), O.gaussian_stats(tgt)))
    line(f"GaussianModel.update x2 (fp64 sums + outer products), [{B} x {D}]", gpu_ms(update), c)
    op.reset()
    update()
    if CPU:
        ms, cs = O.gaussian_fit(*O.gaussian_stats(src))
        mt, ct = O.gaussian_fit(*O.gaussian_stats(tgt))
        c = cpu_ms(lambda: (O.w2_gaussian(ms, mt, cs, ct, make_pd=True), O.transport_operator_full(cs, ct)), reps=1)
    line(f"GaussianTransport.compute: W2^2 + eq.17 operator (4 Jacobi eigh + GEMMs), D = {D}", gpu_ms(op.compute, reps=3, warm=1), c)
    probe = sg[:2048]
    if CPU:
        T = O.transport_operator_full(cs, ct)
        c = cpu_ms(lambda: O.apply_transport(src[:2048], ms, mt, T))
    line(f"GaussianTransport.transport, [2048 x {D}]", gpu_ms(lambda: op.transport(probe)), c)


def codebook(K, d, B):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, d, generator=g).cuda()
    model = A.CodebookModel(d, mixture_cfg=dict(n_components=K, training_mode="argmax")).cuda().train()
    model.update(x)
    line(f"CodebookModel.update (assign + k-means statistics), K = {K}, [{B} x {d}]", gpu_ms(lambda: model.update(x)))
    model.eval()
    line(f"CodebookModel.predict (eval, argmax), K = {K}, [{B} x {d}]", gpu_ms(lambda: model(x)))


def gmm(K, d, B):
    g = torch.Generator().manual_seed(4)
    centres = torch.randn(K, d, generator=g, dtype=torch.float64) * 3
    xs = (centres[torch.randint(0, K, (B,), generator=g)] + 0.4 * torch.randn(B, d, generator=g, dtype=torch.float64)).cuda()
    xt = (centres.flip(0)[torch.randint(0, K, (B,), generator=g)] * 0.8 + 0.4 * torch.randn(B, d, generator=g, dtype=torch.float64)).cuda()
    mix = dict(metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax", inference_mode="argmax", n_components=K)
    w2 = dict(diag=True, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
    op = A.GMMTransport(d, transport_type="argmax", transport_cfg=w2, source_cfg=dict(dtype=torch.double, mixture_cfg=mix),
                        target_cfg=dict(dtype=torch.double, mixture_cfg=mix)).cuda().train()
    op.update(source_samples=xs, target_samples=xt)
    line(f"GMMTransport.update x2, K = {K}, [{B} x {d}] fp64", gpu_ms(lambda: op.update(source_samples=xs, target_samples=xt)))
    line(f"GMMTransport.compute (fit + K x K Gaussian W2 costs + Sinkhorn plan), K = {K}", gpu_ms(op.compute, reps=3, warm=1))
    op.eval()
    line(f"GMMTransport.transport ('argmax'), [{B} x {d}]", gpu_ms(lambda: op.transport(xs)))


if __name__ == "__main__":
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    print(f"device {torch.cuda.get_device_name(0)}; CPU context on {torch.get_num_threads()} threads" if CPU else "GPU only")
    gaussian(128, 1024)
    gaussian(1024, 3000)
    codebook(1024, 16, 50 * 64)
    gmm(10, 64, 50 * 16)
