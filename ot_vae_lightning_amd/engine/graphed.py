"""``GraphedNelbo``: the hipGraph speed for an UNMODIFIED training loop.

The reference trains through Lightning: ``training_step`` (model/base.py:122-129) calls ``self.loss`` (= ``VAE.nelbo``,
model/vae.py:158-189), Lightning calls ``loss.backward()`` and steps whatever optimizer ``configure_optimizers`` returned
(model/vae.py:148-156).  Issued eagerly that route costs ~250 launches through Python per step (7 ms at batch 1024, host-bound);
``HipTrainer`` is fast (one replayed graph) but replaces the loop.  This class keeps the loop:

    model.loss = GraphedNelbo(model)            # or model.enable_graphed_step()
    out = model.training_step(batch, i); out["loss"].backward(); optimizer.step()      # any torch optimizer

``nelbo`` becomes ONE autograd node whose forward replays a captured forward graph and whose backward replays a captured backward
graph (in the manner of ``torch.cuda.make_graphed_callables``): static input slots, the parameters' gradients written by the
backward kernels straight into their slots of a flat buffer and handed to autograd as views, so that they land in ``p.grad``
without a copy and any optimizer can consume them.  Everything a step mutates besides the parameters (BatchNorm running
statistics, RNG counters) advances inside the replayed graphs exactly as in the eager route.
"""
from __future__ import annotations

import os
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from .. import _lib

from .lifetime import GraphSet, capture_guard
from .segments import SEGMENT_CALLS, SegmentedStep
from .trainer import HipTrainer, _dense_view

__all__ = ["GraphedNelbo"]


class _Replay(torch.autograd.Function):
    """forward = replay of the captured forward graph; backward = replay of the captured backward graph.  The parameters are
    inputs only so that autograd routes their gradients: the graphs read them in place."""

    @staticmethod
    def forward(ctx, cap, *params):
        cap.graph("f").replay()
        ctx.cap = cap
        ctx.set_materialize_grads(False)
        return cap.out3.detach()

    @staticmethod
    def backward(ctx, g):
        cap = ctx.cap
        if g is None:
            return (None,) * (1 + len(cap.params))
        cap.seed.copy_(g)
        cap.graph("b").replay()
        # fresh view objects: autograd keeps ("steals") a gradient it holds the only reference to instead of cloning it
        return (None, *[p._otvae_grad_view() for p in cap.params])


class _Capture:
    """One captured (forward graph "f", backward graph "b") pair and its static tensors.  The graphs live in a ``GraphSet``
    (engine/lifetime.py): released by ``GraphedNelbo.close()`` / a re-capture, or by the finalizer when the last holder -- the
    ``GraphedNelbo`` or an autograd node of a loss that is still alive -- lets go."""
    __slots__ = ("key", "engine", "graphs", "out3", "seed", "params", "artifacts", "logs_keys", "__weakref__")

    def graph(self, name: str):
        g = self.graphs.get(name)
        if g is None:
            raise RuntimeError("this loss belongs to a captured step that has been released (GraphedNelbo.close() or a re-capture "
                               "for another batch shape): call the model again")
        return g


class GraphedNelbo:
    """Callable with ``VAE.nelbo``'s signature and return value; see the module docstring.  Falls back to the plain ``nelbo``
    whenever a replay would not be the same computation: evaluation / no-grad calls, host tensors, a cosine-annealed prior
    coefficient (a kernel argument that changes per step), a batch whose shapes differ from the captured ones (captured anew)."""

    def __init__(self, model, warmup: int = 2):
        # the model holds this object (model.loss): a strong reference back -- or a bound method of the model -- would close a
        # reference cycle and leave the captured graphs to the cyclic collector (engine/lifetime.py)
        self._model_ref = weakref.ref(model)
        self.warmup = warmup
        self._cap: Optional[_Capture] = None
        nelbo = model.nelbo  # before anything re-points model.loss
        self._nelbo_func = nelbo.__func__ if getattr(nelbo, "__self__", None) is model else None
        self._nelbo_obj = None if self._nelbo_func is not None else nelbo

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise ReferenceError("the model of this GraphedNelbo no longer exists")
        return m

    def __deepcopy__(self, memo):
        """``copy.deepcopy(model)`` copies ``model.loss`` with it: the copy must point at the COPIED model (a weak reference is atomic
        for deepcopy: it would keep pointing at the original) and starts without captured graphs (it captures on its first call)."""
        import copy
        m = self._model_ref()
        twin = memo.get(id(m)) if m is not None else None
        if twin is None:
            raise copy.Error("a GraphedNelbo can only be deep-copied as part of its model (copy.deepcopy(model))")
        new = GraphedNelbo.__new__(GraphedNelbo)
        new._model_ref, new.warmup, new._cap = weakref.ref(twin), self.warmup, None
        new._nelbo_func = self._nelbo_func
        new._nelbo_obj = None if self._nelbo_obj is None else copy.deepcopy(self._nelbo_obj, memo)
        return new

    def __reduce__(self):
        raise TypeError("a GraphedNelbo holds captured hipGraphs and cannot be pickled: save model.state_dict() (what a Lightning "
                        "checkpoint holds), or call model.disable_graphed_step() first")

    def _nelbo(self, batch, batch_idx):
        if self._nelbo_func is not None:
            return self._nelbo_func(self.model, batch, batch_idx)
        return self._nelbo_obj(batch, batch_idx)

    def close(self) -> None:
        """Releases the captured graphs now (device synchronize, then destruction in reverse capture order).  The next training
        call captures again."""
        cap, self._cap = self._cap, None
        if cap is not None:
            cap.graphs.release()

    # ---- capture ---------------------------------------------------------------------------------------------
    @staticmethod
    def _key(batch) -> Tuple:
        kw = batch["kwargs"]
        return (tuple(batch["samples"].shape), batch["target"] is batch["samples"],
                tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in kw.items() if isinstance(v, Tensor))))

    def _capture(self, batch) -> _Capture:
        from .. import functional as HF
        model = self.model
        samples, kwargs = batch["samples"], batch["kwargs"]
        old = self._cap
        if old is not None:  # re-capture for a new batch shape: the parameters stay where the first engine put them
            engine = old.engine
            old.graphs.release()   # waits for the device: replays of the old graphs may still be in flight
            if tuple(engine.x.shape) != tuple(samples.shape):
                engine.x = torch.zeros_like(samples)
                engine.eps = torch.zeros((samples.shape[0] * max(1, int(getattr(model, "expansion", 1) or 1)), *model.latent_size),
                                         device=samples.device, dtype=torch.float32)
        else:
            # the engine's layout (flat parameter / gradient buffers, gradient slots, resident transposed weights, static batch);
            # its optimizer is never run
            engine = HipTrainer(model, batch_shape=tuple(samples.shape), use_graph=False, data_parallel=False, step_guard=None,
                                weak_model=True)
        engine.batch_kwargs = {k: v.detach().clone() for k, v in kwargs.items() if isinstance(v, Tensor) and k != "eps"}
        engine._rng_key = None
        cap = _Capture()
        cap.key, cap.engine = self._key(batch), engine
        cap.graphs = GraphSet(engine.device)
        weakref.finalize(cap, GraphSet.release, cap.graphs).atexit = False   # (see HipTrainer: not at interpreter exit)
        cap.params = engine.params
        cap.seed = torch.tensor([1.0, 0.0, 0.0], device=engine.device)
        engine._seed = cap.seed
        explicit_eps = "eps" in kwargs

        def forward():
            if not explicit_eps:
                engine._draw_eps()      # device-side generator: every replay draws fresh noise
            engine._refresh_wd()
            HF.SlotArena.begin_step(engine.device)   # the BatchNorm statistic slots of this forward pass
            try:
                loss, logs, art = self._nelbo(engine._batch(), 0)
            finally:
                HF.SlotArena.end_step(engine.device)
            return loss, logs, art

        def backward(loss):
            for p in engine.params:
                p.grad = None
            # the backward pass's statistic slots lie behind the forward pass's; zeroed again here: this graph may be replayed more than
            # once per forward replay (retain_graph)
            HF.SlotArena.resume(engine.device, rezero=True)
            try:
                engine._backward(loss)
            finally:
                HF.SlotArena.end_step(engine.device)
            HF._PendingReduce.flush(engine.device)
            engine._collect_loose_grads()
            for p in engine.params:    # the captured pass's own p.grad objects must not outlive the capture
                p.grad = None
            # nor may the model's handles keep this pass's autograd graph -- and its AccumulateGrad nodes, bound to the stream
            # they were created on -- alive into the next one (warm-up stream -> capture stream)
            model._last_cut = None
            model._last_nelbo = None

        # everything a step mutates besides what the optimizer owns: restored after warm-up + capture
        flat_ids = {id(p) for p in engine.params}
        state = [t for t in model.buffers()] + [p.data for p in model.parameters() if id(p) not in flat_ids]
        state += [m.__dict__["_dropout_key"] for m in model.modules() if isinstance(m.__dict__.get("_dropout_key"), Tensor)]
        snap = [t.clone() for t in state]
        s = _lib.fresh_stream(engine.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                backward(forward()[0])
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        mode = dict(capture_error_mode=os.environ.get("OTVAE_CAPTURE_ERROR_MODE", "thread_local"))
        cstream = _lib.fresh_stream(engine.device)   # ONE capture stream for the forward and the backward graphs (segments.py)
        with capture_guard():
            with torch.cuda.graph(cap.graphs.new("f"), stream=cstream, **mode):
                loss, logs, art = forward()
                cap.out3 = model._last_nelbo
            pool = cap.graphs.get("f").pool()
            if SEGMENT_CALLS > 0 and HF.WGRAD_SIDE_STREAM == 1:
                # the backward pass as a chain of linear graphs + side graphs (engine/segments.py): ~0.1 ms of host time per replay
                # instead of ~0.8 ms for one graph with a fork per layer, which matters on this host-bound route
                seg = SegmentedStep(engine.device, HF._PendingReduce.side_stream(engine.device), pool=pool, stream=cstream)
                cap.graphs.put("b", seg)
                HF._PendingReduce.begin_segments(engine.device, seg)
                try:
                    with seg:
                        backward(loss)
                finally:
                    HF._PendingReduce.end_segments(engine.device)
                del seg
            else:
                with torch.cuda.graph(cap.graphs.new("b"), pool=pool, stream=cstream, **mode):
                    backward(loss)
        cap.out3 = cap.out3.detach()
        # tensors only (preds, latents, preds_mean ...): a prior's lazily built distribution objects could hold the captured pass's
        # autograd graph -- and with it AccumulateGrad nodes bound to the capture stream -- alive for good
        cap.artifacts = {k: v.detach() for k, v in art.items() if isinstance(v, Tensor) and k not in ("samples", "target")}
        del loss, logs, art
        model._last_nelbo = None
        model._last_cut = None
        with torch.no_grad():
            for t, v in zip(state, snap):
                t.copy_(v)
        torch.cuda.synchronize()
        return cap

    # ---- call ------------------------------------------------------------------------------------------------
    def __call__(self, batch, batch_idx: int):
        model = self.model
        samples = batch["samples"]
        prior = getattr(model, "prior", None)
        annealing = getattr(prior, "annealing_steps", 0) > int(getattr(model, "global_step", 0) or 0)
        if not (torch.is_grad_enabled() and model.training and samples.is_cuda) or annealing:
            return self._nelbo(batch, batch_idx)
        if self._cap is None or self._cap.key != self._key(batch):
            self._cap = self._capture(batch)
        cap, eng = self._cap, self._cap.engine
        # A gradient that IS its slot of the flat buffer (autograd kept the view the last backward returned) would be overwritten by
        # this step's backward graph and then added to itself.  It is either stale zeros (zero_grad(set_to_none=False)) or the sum
        # of earlier micro-batches (gradient accumulation, Lightning's accumulate_grad_batches > 1): move ALL such gradients out
        # of the buffer with one copy, as views of the copy -- autograd then accumulates this step's slots onto them.
        acc = None
        for p, off in zip(cap.params, eng.offsets):
            g = p.grad
            if g is not None and g.data_ptr() == p._otvae_grad_view().data_ptr():
                if acc is None:
                    acc = eng.gflat.clone()
                p.grad = _dense_view(acc, off, p.data)
        eng.x.copy_(samples, non_blocking=True)
        if batch["target"] is not samples:
            raise ValueError("GraphedNelbo replays VAE.nelbo with target = samples (what batch_preprocess builds)")
        for k, v in batch["kwargs"].items():
            if k == "eps":
                eng.eps.copy_(v.reshape(eng.eps.shape), non_blocking=True)
            elif isinstance(v, Tensor):
                eng.batch_kwargs[k].copy_(v, non_blocking=True)
        out3 = _Replay.apply(cap, *cap.params)
        model._last_out3 = out3.detach()
        loss = out3[0]
        logs = {"train/loss/total": loss, "train/loss/recon": out3[1], "train/loss/prior": out3[2]}
        return loss, logs, {**batch, **cap.artifacts}
