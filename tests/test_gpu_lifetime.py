"""Lifetime of captured hipGraphs (engine/lifetime.py).

The abort these tests pin: ``at::cuda::CUDAGraph::~CUDAGraph`` (HIPGraph.cpp:324 of torch 2.10 for ROCm) ends with a device
synchronize; run while the calling thread has a stream capture open it throws ``hipErrorStreamCaptureUnsupported`` out of the
destructor and the process dies with SIGABRT (``tools/probe/graph_teardown.py``, ``profiles/r04_graph_teardown.txt``).  The
scenario is run ONCE, in a child process (an abort must fail this test, not end the test session), with its stderr kept under
``gpurun_out/``.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DROP_SCRIPT = r"""
import faulthandler, gc, os, sys, weakref, torch
faulthandler.enable()
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import ot_vae_lightning_amd as A
from ot_vae_lightning_amd.engine import lifetime
from detfill import mnist_like, normal
gc.disable()   # the collector runs only where this script calls it: the outcome does not depend on allocation counts
B = 32
xs = [mnist_like(B, 70 + i).cuda() for i in range(2)]
es = [normal((B, 128, 1, 1), 80 + i).cuda() for i in range(2)]

def make(seed):
    torch.manual_seed(seed)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()

# 1. an engine dropped WITHOUT close() by plain reference counting: its graphs go at once, through the ordered path
tr0 = A.HipTrainer(make(1), batch_shape=(B, 1, 32, 32), use_graph=True)
tr0.step(xs[0], es[0])
gs0 = tr0._gs
assert gs0, "captured"
del tr0
assert not gs0 and lifetime.parked() == 0, "dropping the engine releases its graphs immediately (finalizer), nothing parked"
print("1 ok: refcount drop released the graphs", flush=True)

# 2. THE ROUND-3 ABORT: a captured, un-closed engine in a dead reference cycle; the collector runs INSIDE the next engine's capture
tr1 = A.HipTrainer(make(2), batch_shape=(B, 1, 32, 32), use_graph=True)
tr1.step(xs[0], es[0]); tr1.step(xs[1], es[1])
torch.cuda.synchronize()
tr1._me = tr1                      # dead cycle: only the cyclic collector can free it
alive = weakref.ref(tr1)
del tr1
assert alive() is not None
model2 = make(3)
plain_nelbo, hits = model2.nelbo, []
def nelbo_with_collector(batch, idx):
    if torch.cuda.is_current_stream_capturing():
        hits.append(gc.collect())  # destroys engine 1 here, with this thread's capture open
    return plain_nelbo(batch, idx)
model2.nelbo = nelbo_with_collector
tr2 = A.HipTrainer(model2, batch_shape=(B, 1, 32, 32), use_graph=True)
out = [tr2.step(xs[i], es[i]).clone() for i in range(2)]
torch.cuda.synchronize()
assert hits and alive() is None, "the collector ran inside the capture and freed the first engine"
assert lifetime.parked() == 1, "its graphs were parked, not destroyed inside the capture"
assert all(torch.isfinite(o).all() for o in out)
# the same two steps from an engine that saw no such thing: bit-equal (the capture was not disturbed)
model3 = make(3)
tr3 = A.HipTrainer(model3, batch_shape=(B, 1, 32, 32), use_graph=True)
assert lifetime.parked() == 1
ref = [tr3.step(xs[i], es[i]).clone() for i in range(2)]
assert lifetime.parked() == 0, "the next capture destroyed the parked graphs before it began"
assert all(torch.equal(a, b) for a, b in zip(out, ref)), (out, ref)
tr2.close(); tr3.close()
print("2 ok: collector inside a capture parks the dead engine's graphs", flush=True)

# 3. the graph route of the unmodified loop: a model that goes out of scope (the model IS a cycle: self.loss = bound method in the
#    reference's design) while its loss tensor is still alive; collected inside the next capture
m4 = make(5).enable_graphed_step()
m4.batch_preprocess = lambda b: {"samples": b[0], "target": b[0], "kwargs": {"eps": b[1]}}
o4 = m4.training_step((xs[0], es[0]), 0)
o4["loss"].backward()
cap4 = weakref.ref(m4.loss._cap)
del m4, o4
m5 = make(6)
plain5 = m5.nelbo
def nelbo5(batch, idx):
    if torch.cuda.is_current_stream_capturing():
        gc.collect()
    return plain5(batch, idx)
m5.nelbo = nelbo5
m5.enable_graphed_step()
m5.batch_preprocess = lambda b: {"samples": b[0], "target": b[0], "kwargs": {"eps": b[1]}}
o5 = m5.training_step((xs[1], es[1]), 0)
o5["loss"].backward()
torch.cuda.synchronize()
assert cap4() is None
m5.disable_graphed_step()
assert lifetime.parked() == 0
print("3 ok: GraphedNelbo released with its model", flush=True)
print("LIFETIME-OK", flush=True)
""" % ROOT


def test_dropped_uncaptured_engines_never_destroy_a_graph_inside_a_capture():
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    log = os.path.join(out_dir, "r04_lifetime_test.log")
    with open(log, "w") as f:
        r = subprocess.run([sys.executable, "-c", DROP_SCRIPT], stdout=f, stderr=subprocess.STDOUT, timeout=600)
    text = open(log).read()
    assert r.returncode == 0 and "LIFETIME-OK" in text, f"exit {r.returncode}\n{text[-3000:]}"


def test_graphed_nelbo_accumulates_gradients_over_micro_batches():
    """without ``zero_grad`` between two micro-batches (Lightning's ``accumulate_grad_batches`` = 2, a flag of the
    reference's CLI) ``p.grad`` must hold the SUM of both micro-batches' gradients, as in the eagerly issued route."""
    import ot_vae_lightning_amd as A
    from detfill import mnist_like, normal
    B = 32
    xs = [mnist_like(B, 30 + i).cuda() for i in range(3)]
    es = [normal((B, 128, 1, 1), 40 + i).cuda() for i in range(3)]

    def make():
        torch.manual_seed(33)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        m = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
        m.batch_preprocess = lambda b: {"samples": b[0], "target": b[0], "kwargs": {"eps": b[1]}}
        return m

    def run(graphed, zero_with_none):
        m = make()
        if graphed:
            m.enable_graphed_step()
        params = list(m.optim_parameters())
        opt = torch.optim.SGD(params, lr=0.0)
        # one ordinary step first, so that p.grad of the graph route is the slot view when accumulation starts
        m.training_step((xs[0], es[0]), 0)["loss"].backward()
        opt.zero_grad(set_to_none=zero_with_none)
        for i in (1, 2):
            m.training_step((xs[i], es[i]), i)["loss"].backward()
        g = torch.cat([p.grad.detach().float().flatten() for p in params]).clone()
        if graphed:
            m.disable_graphed_step()
        return g

    eager = run(False, True)
    for zero_with_none in (True, False):
        got = run(True, zero_with_none)
        err = float((got - eager).abs().max() / eager.abs().max())
        assert err < 2e-6, (zero_with_none, err)   # same kernels; the sum's association differs by one rounding at most


def test_parameter_ema_in_the_optimizer_kernel_vs_the_torch_ema_rule():
    """``ema_decay`` (reference model/base.py:99,146-190): HipTrainer folds the average into its Adam kernel.  Held
    against the oracle's restatement of torch_ema's update rule applied to the SAME parameter trajectory (bit-exact: three fp32
    roundings either way), then the store / copy_to / restore swap of the evaluation hooks, then the step guard (a refused step
    leaves the average alone), then the host-driven route (``on_fit_start`` + ``on_before_zero_grad`` around a stock optimizer)."""
    import otvae_oracle as O
    import ot_vae_lightning_amd as A
    from detfill import mnist_like, normal
    B = 32
    xs = [mnist_like(B, 50 + i).cuda() for i in range(4)]
    es = [normal((B, 128, 1, 1), 60 + i).cuda() for i in range(4)]

    def make(**kw):
        torch.manual_seed(17)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1), **kw).cuda().train()

    for graph in (False, True):
        model = make(ema_decay=0.9)
        tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=graph)
        assert model._ema is tr.ema and tr.ema.in_optimizer
        ref = O.ParamEMARef([tr.pflat.detach().cpu()], 0.9)
        for i in range(4):
            tr.step(xs[i], es[i])
            ref.update([tr.pflat.detach().cpu()])
        assert torch.equal(tr.ema.shadow.cpu(), ref.shadow[0]), f"graph={graph}: shadow differs from the torch_ema rule"
        assert not torch.equal(tr.ema.shadow, tr.pflat)
        # the evaluation swap (reference hooks: store + copy_to at epoch start, restore at its end)
        live = tr.pflat.clone()
        model.on_validation_epoch_start()
        assert torch.equal(tr.pflat, tr.ema.shadow)
        assert torch.equal(next(model.encoder.parameters()).detach().flatten().sort()[0],
                           tr.ema.shadow[: next(model.encoder.parameters()).numel()].sort()[0])   # the modules see the average
        model.on_validation_epoch_end()
        assert torch.equal(tr.pflat, live)
        # a refused step (NaN batch) leaves the average and the count alone
        bad = xs[0].clone()
        bad[1, 0, 3, 3] = float("nan")
        before = tr.ema.shadow.clone()
        tr.step(bad, es[0])
        torch.cuda.synchronize()
        assert tr.skipped_steps == 1 and torch.equal(tr.ema.shadow, before) and tr.ema.state_dict()["num_updates"] == 4
        tr.close()

    # host-driven route: the reference's loop with a stock optimizer -- issued eagerly, and through the graph route (whose first call
    # moves the parameters into a flat buffer AFTER on_fit_start made the average)
    for graphed in (False, True):
        model = make(ema_decay=0.95)
        model.batch_preprocess = lambda b: {"samples": b[0], "target": b[0], "kwargs": {"eps": b[1]}}
        if graphed:
            model.enable_graphed_step()
        params = list(model.optim_parameters())
        opt = torch.optim.SGD(params, lr=0.05)
        model.on_fit_start()
        ref = O.ParamEMARef([p.detach().cpu().contiguous() for p in params], 0.95)
        for i in range(3):
            model.training_step((xs[i], es[i]), i)["loss"].backward()
            opt.step()
            model.on_before_zero_grad(opt)
            opt.zero_grad()
            ref.update([p.detach().cpu().contiguous() for p in params])
        for s_, r_ in zip(model._ema.shadow, ref.shadow):
            assert torch.equal(s_.cpu().contiguous(), r_), f"graphed={graphed}"
        live = [p.detach().clone() for p in params]
        model.on_validation_epoch_start()
        assert all(torch.equal(p.detach(), s_) for p, s_ in zip(params, model._ema.shadow))
        model.on_validation_epoch_end()
        assert all(torch.equal(p.detach(), l_) for p, l_ in zip(params, live))
        if graphed:
            model.disable_graphed_step()


def test_fresh_streams_never_alias_each_other_or_the_framework_pool():
    """``torch.cuda.Stream()`` takes the next of 32 pooled streams round-robin: the 33rd is the first again.  A side stream that IS the
    capture stream (or the prior lane) gave a captured step whose fork waited on itself, and the HIP graph executor died in
    hip::Graph::UpdateStreams at the first replay (round 4, once the suite had built enough engines).  ``_lib.fresh_stream`` hands out
    HIP streams of the library's own: distinct while alive, recycled when their wrapper dies."""
    import gc
    from ot_vae_lightning_amd import _lib
    pool = [torch.cuda.Stream() for _ in range(40)]
    assert len({s.cuda_stream for s in pool}) <= 32, "the framework's pool wraps around (the premise of this test)"
    mine = [_lib.fresh_stream("cuda") for _ in range(48)]
    handles = {s.cuda_stream for s in mine}
    assert len(handles) == 48 and not (handles & {s.cuda_stream for s in pool}) and 0 not in handles
    # they are real streams: work, events, capture
    a = torch.ones(1024, device="cuda")
    with torch.cuda.stream(mine[0]):
        b = a * 2
    mine[1].wait_stream(mine[0])
    with torch.cuda.stream(mine[1]):
        c = b + 1
    torch.cuda.current_stream().wait_stream(mine[1])
    assert float(c.sum()) == 3 * 1024
    g = torch.cuda.CUDAGraph()
    out = torch.zeros(1024, device="cuda")
    with torch.cuda.graph(g, stream=mine[2], capture_error_mode="thread_local"):
        out.copy_(a * 5)
    g.replay()
    torch.cuda.synchronize()
    assert float(out.sum()) == 5 * 1024
    del g
    # a wrapper that dies hands its stream back; the next request reuses it instead of creating the 49th
    victim = mine.pop().cuda_stream
    gc.collect()
    again = _lib.fresh_stream("cuda")
    assert again.cuda_stream == victim
