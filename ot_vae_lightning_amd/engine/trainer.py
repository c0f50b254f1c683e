"""MI355X training engine for the VAE hot path: one training step = forward + backward + (gradient all-reduce) + Adam,
replayed as a hipGraph.

Design (DESIGN.md section "engine"):
  * all trainable parameters live in ONE flat fp32 buffer (each ``nn.Parameter`` is a strided view into it, conv weights
    in HWIO order); gradients, Adam moments and the dgrad-layout weight copies have flat buffers of their own;
  * backward kernels write parameter gradients straight into the flat gradient buffer (``p._otvae_grad_view``), so a
    step needs no gradient zeroing, no per-parameter accumulate kernels and exactly one Adam launch;
  * data parallel = one process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI): every rank runs the
    step on its shard of the batch with rank-local BatchNorm statistics (the reference does not sync BN, SURVEY.md
    section 2.2), then ONE all-reduce(SUM) of the flat gradient buffer (6.6 MiB for the MNIST config) on a side stream,
    and Adam consumes ``grad / world_size``;
  * the whole step is captured once (``torch.cuda.CUDAGraph`` -> hipGraph) and replayed: no Python, no allocator and no
    launch-latency gaps between the ~400 kernels of a step.  With more than one rank the graph is split around the
    collective (graph A: forward+backward, eager RCCL all-reduce, graph B: Adam).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream
from .. import functional as HF
from ..networks.cnn import ConvLayer
from .dp import FlatGradReducer
from .lifetime import GraphSet, capture_guard
from .segments import SEGMENT_CALLS, SegmentedStep

DP_OVERLAP_MIN_BYTES = int(os.environ.get("OTVAE_DP_OVERLAP_MIN_BYTES", str(32 << 20)))


def default_dp_overlap(world: int, grad_bytes: int, env: Optional[str] = None) -> bool:
    """Whether a data-parallel step cuts its backward pass to overlap the decoder's all-reduce (see HipTrainer.__init__): never alone,
    as the environment says when it says anything, else only for gradient buffers of at least DP_OVERLAP_MIN_BYTES."""
    if world <= 1:
        return False
    if env is not None:
        return env != "0"
    return grad_bytes >= DP_OVERLAP_MIN_BYTES

# OTVAE_STATS_SIDE=0 (A/B switch): the loss value and the latent-statistics update of a captured step stay on the launch stream
STATS_ON_SIDE = os.environ.get("OTVAE_STATS_SIDE", "1") != "0"

__all__ = ["HipTrainer", "flatten_parameters"]


def _dense_view(flat: Tensor, off: int, like: Tensor) -> Tensor:
    """A view of ``flat[off: off+numel]`` with ``like``'s shape and (dense, possibly permuted) strides."""
    order = sorted(range(like.dim()), key=lambda d: (-like.stride(d), d))
    phys_shape = [like.shape[d] for d in order]
    inv = [order.index(d) for d in range(like.dim())]
    v = flat[off: off + like.numel()].view(phys_shape)
    return v.permute(inv) if like.dim() > 0 else v.view(())


def flatten_parameters(params: List[torch.nn.Parameter], align: int = 4):
    """Moves ``params`` into one flat buffer (keeping values, shapes and stride order).  Returns (flat, offsets)."""
    device, dtype = params[0].device, params[0].dtype
    offsets, total = [], 0
    for p in params:
        if p.dtype != dtype or p.device != device:
            raise TypeError("all parameters must share dtype and device")
        offsets.append(total)
        total += (p.numel() + align - 1) // align * align
    flat = torch.zeros(total, device=device, dtype=dtype)
    with torch.no_grad():
        for p, off in zip(params, offsets):
            v = _dense_view(flat, off, p.data)
            v.copy_(p.data)
            p.data = v
    return flat, offsets


class HipTrainer:
    """Runs ``model.nelbo`` training steps on the GPU.

    trainer = HipTrainer(model, batch_shape=(1024, 1, 32, 32))
    out = trainer.step(x)          # device tensor [total, recon, prior]; no host sync
    """

    def __init__(self, model, batch_shape, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group=None, use_graph: bool = True, latent_stats=None, dp_overlap: Optional[bool] = None,
                 batch_kwargs: Optional[Dict[str, Tensor]] = None, gradient_clip_val: Optional[float] = None,
                 data_parallel: bool = True, step_guard: Optional[str] = "auto", weak_model: bool = False):
        self.lib = _lib.load()
        # weak_model: the engine is owned BY the model (GraphedNelbo sits in model.loss): a strong reference back would close a
        # cycle model -> loss -> capture -> engine -> model and leave the captured graphs to the cyclic collector (lifetime.py)
        self._model_ref = weakref.ref(model)
        self._model_strong = None if weak_model else model
        self.params = list(model.optim_parameters())
        if not self.params:
            raise ValueError("model has no trainable parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("HipTrainer needs the model on the GPU (model.cuda()); there is no CPU path")
        self.device = dev
        self.pflat, self.offsets = flatten_parameters(self.params)
        self.gflat = torch.zeros_like(self.pflat)
        self.m = torch.zeros_like(self.pflat)
        self.v = torch.zeros_like(self.pflat)
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps], device=dev, dtype=torch.float32)
        self.step_count = torch.zeros(1, device=dev, dtype=torch.int32)
        # the reference's `ema_decay` (model/base.py:99,146-190): the moving average of the parameters is updated by the optimizer
        # kernel itself (otvae_adam_step_ema); the model's epoch hooks swap it in and out around evaluation (engine/ema.py)
        self.ema = None
        if getattr(model, "ema_decay", None) is not None and not weak_model:
            from .ema import ParamEMA
            self.ema = ParamEMA(self.params, model.ema_decay, flat=self.pflat, in_optimizer=True, step_tensor=self.step_count)
            model._ema = self.ema
        # global-norm clipping of the (rank-averaged) gradient, the reference's DDP overlay (configs/ddp.yaml:4); the two
        # floats are {scale Adam applies to the summed gradient, norm of the averaged gradient} of the last step
        self.gradient_clip_val = None if not gradient_clip_val else float(gradient_clip_val)
        self.clip_out = torch.zeros(2, device=dev, dtype=torch.float32)
        self._clip_ws = torch.empty(self.lib.otvae_grad_clip_ws(), device=dev, dtype=torch.float64)
        # Device-side step guard: a captured step has no host in the loop that could stop on a NaN loss as
        # Lightning's does, so the Adam kernel itself refuses a bad step (parameters, moments and the step counter stay as they
        # were; BatchNorm's running buffers refuse non-finite batch statistics on their own) and counts it.
        #   "loss": the step's loss must be finite (free: Adam reads the loss vector the step produced anyway; covers a starved
        #           Sinkhorn solve, whose outputs are NaN-poisoned, and NaN / inf inputs);
        #   "full": also the L2 norm of the (all-reduced) gradient, two small launches unless clipping runs them already --
        #           identical on every rank, hence the mode for world > 1 (a rank-local loss test would let ranks diverge);
        #   None:   unguarded (round-2 behaviour).  "auto" = "loss" on one rank, "full" on several.
        self._guard_arg = step_guard
        self.guard = torch.zeros(2, device=dev, dtype=torch.int32)  # {skipped steps, step number of the last skip}
        # EVERYTHING a step mutates besides what the optimizer owns lives in ONE flat range of 4-byte words (`_rebase_step_state`,
        # called at the end of this constructor): a guarded step keeps a copy from its start and a refused step puts it back
        self.rflat = self.rbackup = None
        # a FRESH tensor object per call (autograd steals a gradient it holds the only reference to and clones it otherwise),
        # made from a slot view built once: detach() is one shallow copy instead of slice + view + permute
        for p, off in zip(self.params, self.offsets):
            p._otvae_grad_view = _dense_view(self.gflat, off, p.data).detach
        # dgrad-layout ([T][Cout][Cin]) copies of every conv weight, refreshed by ONE launch at the start of each step
        # (the ViT's Linear weights are 1x1 layers on the same kernels: [out, in] = [Cn, Cs], one tap)
        # (grouped / dilated layers hand the kernels an expanded temporary, functional._WeightExpandFn: nothing resident to refresh)
        self.conv_weights = [mod.weight for mod in model.modules() if isinstance(mod, ConvLayer) and getattr(mod, "_expand", None) is None]
        self.conv_weights += [p_ for n_, p_ in model.named_parameters()
                              if p_.dim() == 2 and (n_.endswith("in_proj_weight") or getattr(p_, "_otvae_linear", False))]
        flat_ids = {id(p): off for p, off in zip(self.params, self.offsets)}
        wd_total = sum((w.numel() + 3) // 4 * 4 for w in self.conv_weights)
        self.wdflat = torch.empty(max(4, wd_total), device=dev, dtype=torch.float32)
        off, table, self._wd_loose = 0, [], []
        for w in self.conv_weights:
            w._otvae_wd = self.wdflat[off: off + w.numel()]
            cn, cs, kh, kw = w.shape if w.dim() == 4 else (*w.shape, 1, 1)
            if id(w) in flat_ids:
                table.append([flat_ids[id(w)], off, kh * kw, cs, cn])
            else:  # a frozen conv weight is not in the flat buffer: transposed on its own
                self._wd_loose.append(w)
            off += (w.numel() + 3) // 4 * 4
        self._wd_table = torch.tensor(table, device=dev, dtype=torch.int64) if table else None
        self._wd_max = max((t[2] * t[3] * t[4] for t in table), default=0)
        # distributed
        self.group = process_group
        # data_parallel=False: this engine is rank-local even when a process group exists (no gradient exchange)
        self.reducer = FlatGradReducer(self.gflat, process_group, enabled=data_parallel)
        self.world = self.reducer.world
        if self._guard_arg not in (None, "auto", "loss", "full"):
            raise ValueError("step_guard must be None, 'auto', 'loss' or 'full'")
        self.step_guard = ("full" if self.world > 1 else "loss") if self._guard_arg == "auto" else self._guard_arg
        # Data-parallel overlap: backward runs in two phases cut at the encoder's output (loss, decoder and prior side
        # first); the decoder's gradient range is all-reduced on the reducer's stream WHILE the encoder's backward runs,
        # the rest afterwards.
        # The cut costs the step 0.18 ms on its own (three graphs and three collective calls instead of two and one: 2.954 against
        # 2.775 ms with a 1-rank RCCL group, profiles/r03_dp_host_enqueue.txt) and hides about half of the all-reduce, so it pays only
        # when the all-reduce is longer than ~0.36 ms -- tens of MB of gradients on xGMI.  Default since round 4: on for world > 1 when
        # the flat gradient buffer holds at least DP_OVERLAP_MIN_BYTES (the MNIST / CIFAR networks of the BASELINE configs: 6.9 / 27 MB,
        # below it: ONE all-reduce between the forward-backward graph and the optimizer graph); OTVAE_DP_OVERLAP=1 / 0 or
        # dp_overlap=True / False decide it outright.  (Which side of the threshold an 8-GPU node really is on is unmeasured here.)
        if dp_overlap is None:
            dp_overlap = default_dp_overlap(self.world, self.gflat.numel() * self.gflat.element_size(), os.environ.get("OTVAE_DP_OVERLAP"))
        self._dec_range = self._decoder_range() if dp_overlap else None
        self.dp_overlap = self._dec_range is not None
        # static I/O
        self.x = torch.zeros(batch_shape, device=dev, dtype=torch.float32)
        # one noise draw per latent the prior sees: `expansion` replicas of every image (VAE(expansion=n), model/vae.py:158-167)
        lat = (batch_shape[0] * max(1, int(getattr(model, "expansion", 1) or 1)), *model.latent_size)
        self.eps = torch.zeros(lat, device=dev, dtype=torch.float32)
        # further per-batch keyword tensors of ``model.nelbo`` (e.g. ``labels`` of a conditional model): resident copies of the
        # examples given here, refreshed by ``step(..., name=tensor)``
        self.batch_kwargs = {k: v.to(dev).clone() for k, v in (batch_kwargs or {}).items()}
        self._rng_key = None
        self.latent_stats = latent_stats  # optional TransportOperator fed with the step's latents (LatentTransport)
        if latent_stats is not None and self.world > 1 and use_graph:
            # reduce_on_update=True would all-reduce the batch statistics inside the captured step; the statistics are
            # additive, so the captured step accumulates rank-locally and ``fit`` reduces once
            for mod in latent_stats.modules():
                if getattr(mod, "reduce_on_update", False):
                    raise ValueError("HipTrainer(latent_stats=...) with more than one rank captures the statistics update into "
                                     "the step's hipGraph: build the operator's models with reduce_on_update=False (they are "
                                     "reduced once, in fit())")
        # [1, 0, 0]: the gradient of the loss with respect to the nelbo kernel's output vector (resident: see _backward)
        self._seed = torch.tensor([1.0, 0.0, 0.0], device=self.device, dtype=torch.float32)
        self.out: Optional[Tensor] = None
        self.latents: Optional[Tensor] = None
        self.use_graph = use_graph
        # the captured step's hipGraphs: "fb" (forward + backward [+ Adam]), "b2" (the encoder's backward, data-parallel overlap),
        # "opt" (Adam behind the all-reduce), or "segments" (a SegmentedStep).  The GraphSet is their ONLY strong owner; an engine
        # dropped without close() releases them through the same ordered path (engine/lifetime.py)
        self._gs = GraphSet(dev)
        self._finalizer = weakref.finalize(self, GraphSet.release, self._gs)
        # not at interpreter exit: no capture can be open there, and what is still alive then goes down with the process as it
        # always has (the ordered path is for owners that die while the program runs)
        self._finalizer.atexit = False
        self._captured = False
        self.n_steps = 0
        if self.step_guard is not None:
            self._rebase_step_state()

    def _step_state_tensors(self) -> List[Tensor]:
        """Every tensor OBJECT (buffer, parameter outside the flat buffer, RNG key ...) whose storage a training step writes besides
        parameters / moments / gradients: all buffers of the model (BatchNorm running statistics and counters, the EMA embeddings of
        a ConditionalGaussianPrior, GaussianW2Prior's warm-start flag), parameters the optimizer does not own (frozen ones an EMA
        rewrites), the dropout key counters, the running statistics of the latent operator, and what modules declare through
        ``_otvae_step_state(latent_shape, device)`` (state that is not a registered buffer: GaussianW2Prior's warm-start basis)."""
        model = self.model
        flat_ids = {id(p) for p in self.params}
        ts: List[Tensor] = [b for b in model.buffers()]
        ts += [p for p in model.parameters() if id(p) not in flat_ids]
        for mod in model.modules():
            key = mod.__dict__.get("_dropout_key")
            if isinstance(key, Tensor):
                ts.append(key)
            decl = getattr(mod, "_otvae_step_state", None)
            if decl is not None:
                ts += list(decl(tuple(self.eps.shape), self.device))
        if self.latent_stats is not None:
            ts += [b for b in self.latent_stats.buffers()] + [p for p in self.latent_stats.parameters()]
        seen, out = set(), []
        for t in ts:
            if t is None or id(t) in seen or t.numel() == 0 or t.device != self.device or not t.is_contiguous():
                continue
            seen.add(id(t))
            out.append(t)
        return out

    def _rebase_step_state(self) -> None:
        """Moves the storage of every tensor of ``_step_state_tensors`` into one flat byte range (16-byte aligned slots, any dtype:
        the guard kernels copy 4-byte words bit for bit), ``rflat``; ``rbackup`` holds the copy a guarded step takes at its start."""
        ts = self._step_state_tensors()
        if not ts:
            return
        offs, total = [], 0
        for t in ts:
            offs.append(total)
            total += (t.numel() * t.element_size() + 15) // 16 * 16
        flat = torch.zeros(total, device=self.device, dtype=torch.uint8)
        with torch.no_grad():
            for t, off in zip(ts, offs):
                v = flat[off: off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
                v.copy_(t.data)
                t.data = v
        self.rflat = flat.view(torch.float32)
        self.rbackup = torch.empty_like(self.rflat)

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise ReferenceError("the model of this engine no longer exists")
        return m

    @property
    def _segments(self) -> Optional[SegmentedStep]:
        return self._gs.get("segments")

    # -- pieces of a step ----------------------------------------------------------------------------------------
    def _refresh_wd(self):
        lib = self.lib
        if self._wd_table is not None:
            check(lib.otvae_weight_transpose_batched(ptr(self.pflat), ptr(self.wdflat), ptr(self._wd_table),
                                                     self._wd_table.shape[0], self._wd_max, stream()),
                  "otvae_weight_transpose_batched")
        for w in self._wd_loose:
            cn, cs, kh, kw = w.shape if w.dim() == 4 else (*w.shape, 1, 1)
            check(lib.otvae_weight_transpose(ptr(w), ptr(w._otvae_wd), kh * kw, cs, cn, stream()),
                  "otvae_weight_transpose")

    def _step_begin(self):
        # one launch: the step counter, the step guard's backup of the running state, the zeroing of this step's BatchNorm statistic
        # slots (functional.SlotArena)
        z = HF.SlotArena.begin_step(self.device, fused_zero=True)
        guarded = self.step_guard is not None and self.rflat is not None
        check(self.lib.otvae_step_begin_slots(ptr(self.step_count), ptr(self.rflat) if guarded else None,
                                              ptr(self.rbackup) if guarded else None, self.rflat.numel() if guarded else 0,
                                              ptr(z), z.numel() if z is not None else 0, stream()), "otvae_step_begin_slots")

    def _forward_backward(self):
        self._step_begin()
        for p in self.params:
            p.grad = None
        from ..functional import PriorLane, _PendingReduce
        PriorLane.enabled = True  # the prior's OT work may run beside the decoder: this method joins it (functional.PriorLane)
        # (Round 3 tried to put the step's small off-chain launches -- the refresh of the transposed weights, the loss vector, the
        # latent statistics -- on the same lane: 2.80 -> 2.86 ms at batch 1024, the extra graph branches cost more than the ~40 us
        # of launches they take off the chain; removed.)
        try:
            self._refresh_wd()
            from ..functional import _PendingReduce as _PR
            # what depends on the forward pass only and is read by nothing before the optimizer -- the loss value (ops._nelbo_fwd_launch)
            # and the latent statistics -- goes to the side stream with the first weight-gradient fork of the backward pass (joined with
            # it in flush); only in a captured single-graph step, where that fork exists
            # ... and only for the package's own `VAE.nelbo`: between `nelbo()` returning and the join in `flush` the loss vector is not
            # yet written on the launch stream, so a subclass whose nelbo READS the loss it just computed (combining terms, stacking
            # logs) would capture garbage -- for such a model the value is launched in line
            from ..model.vae import VAE as _VAE
            own_nelbo = getattr(type(self.model), "nelbo", None) is _VAE.nelbo
            on_side = (STATS_ON_SIDE and own_nelbo and self._segments is None and HF.WGRAD_SIDE_STREAM == 1 and
                       torch.cuda.is_current_stream_capturing())
            _PR._defer[self.device] = on_side
            try:
                loss, logs, art = self.model.nelbo(self._batch(), 0)
            finally:
                _PR._defer[self.device] = False
            self.latents = art["latents"].detach()
            if on_side and self.latent_stats is not None:
                lat_, stats_ = self.latents.flatten(1), self.latent_stats
                _PR._defer[self.device] = True
                _PR.defer_to_side(self.device, lambda: stats_.update(target_samples=lat_), lat_)
                _PR._defer[self.device] = False
            self._backward(loss)
        except BaseException:
            # a pass that raised (in the forward pass as well: the deferred loss / statistics launches are queued there) must leave
            # nothing behind for the next step's first fork to run against this step's tensors
            _PendingReduce.reset(self.device)
            raise
        finally:
            HF.SlotArena.end_step(self.device)
            PriorLane.enabled = False
            PriorLane.join(self.device)  # (already joined by the prior's backward when it took part in the pass)
        # the model's handle on the encoder output would keep this step's autograd graph -- and with it the parameters'
        # AccumulateGrad nodes, bound to the stream they were created on -- alive into the next step (and from the
        # warm-up stream into the capture: "AccumulateGrad node's stream does not match ...")
        if hasattr(self.model, "_last_cut"):
            self.model._last_cut = None
        _PendingReduce.flush(self.device)  # normally already done by the autograd-engine callback at the end of backward
        self._collect_loose_grads()
        self._logs = {k: v.detach() for k, v in logs.items()}  # no reference into the autograd graph survives the step
        _PendingReduce.run_deferred_inline(self.device)  # (a pass without forks: nothing took the deferred work)
        if self.latent_stats is not None and not on_side:
            lat = self.latents.flatten(1)  # [B, D] with transport_dims = (1, 2, 3)
            self.latent_stats.update(target_samples=lat)
        return self._logs

    def _batch(self):
        return {"samples": self.x, "target": self.x, "kwargs": {"eps": self.eps, **self.batch_kwargs}}

    def _collect_loose_grads(self, params=None) -> None:
        """Gradients that reached a parameter through plain autograd (p.grad) instead of being written into the flat
        buffer by a kernel (embeddings, learned tokens, LayerNorm weights of the ViT ...): copied into their slot.  The
        convolution / BatchNorm kernels write their slots directly (then p.grad IS the slot): nothing to do for them.
        ``params``: the parameters whose backward has just run (two-phase backward: a slot of the other phase may already
        be under its all-reduce and must not be written)."""
        loose = self.__dict__.setdefault("_loose_params", set())
        pairs = []
        for p in (self.params if params is None else params):
            g = p.grad
            if g is None:
                if id(p) in loose:  # took no part in this step: its slot must not keep the previous step's gradient
                    p._otvae_grad_view().zero_()
                continue
            slot = p._otvae_grad_view()
            if g.data_ptr() != slot.data_ptr():
                if g.dtype == torch.float32 and g.is_contiguous() and slot.is_contiguous():
                    pairs.append((g, slot))
                else:
                    slot.copy_(g)
                loose.add(id(p))
        if pairs:  # one launch for all of them (the ViT: embed / class tokens, position embeddings, the prior's class embeddings)
            n = len(pairs)
            src = (C.c_void_p * n)(*[g.data_ptr() for g, _ in pairs])
            dst = (C.c_void_p * n)(*[s_.data_ptr() for _, s_ in pairs])
            cnt = (C.c_int64 * n)(*[g.numel() for g, _ in pairs])
            check(self.lib.otvae_copy_batched(n, src, dst, cnt, stream()), "otvae_copy_batched")

    def _backward(self, loss, **kw) -> None:
        """``loss.backward()`` seeded at the nelbo kernel's [total, recon, prior] vector with a resident [1, 0, 0]: the
        select / ones_like / zeros that autograd would build between the forward and the backward pass are three launches."""
        out3 = getattr(self.model, "_last_nelbo", None)
        try:
            if out3 is None:
                torch.autograd.backward(loss, **kw)
            else:
                torch.autograd.backward(out3, grad_tensors=[self._seed], **kw)
        except BaseException:
            # the deferred weight-gradient jobs / reductions of the interrupted pass must not leak into the next one
            from ..functional import _PendingReduce
            _PendingReduce.reset(self.device)
            raise
        finally:
            if out3 is not None:
                self.model._last_nelbo = None

    def _decoder_range(self):
        """(lo, hi) of the decoder's gradients in the flat buffer, or None when they are not one contiguous range"""
        dec = getattr(self.model, "decoder", None)
        if dec is None or not hasattr(self.model, "encoder"):
            return None
        ids = {id(p) for p in dec.parameters()}
        idx = [i for i, p in enumerate(self.params) if id(p) in ids]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            return None
        lo = self.offsets[idx[0]]
        hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else self.gflat.numel()
        enc_ids = {id(p) for p in self.model.encoder.parameters()}
        self._enc_params = [p for p in self.params if id(p) in enc_ids]
        self._post_params = [p for p in self.params if id(p) not in enc_ids]  # decoder, prior: downstream of the cut
        if not self._enc_params:
            return None
        return lo, hi

    def _phase1(self):
        """forward + backward of everything downstream of the encoder's output h (loss, decoder, prior): gradients of
        their parameters and dL/dh.  Same kernels in the same order as ``_forward_backward`` up to the cut."""
        self._step_begin()
        self._refresh_wd()
        for p in self.params:
            p.grad = None
        from ..functional import PriorLane
        PriorLane.enabled = True
        try:
            loss, logs, art = self.model.nelbo(self._batch(), 0)
            h = getattr(self.model, "_last_cut", None)
            if h is None:  # nothing upstream of the cut needs a gradient: one-phase backward
                self._backward(loss)
            else:
                # retain_graph: without it the engine also releases the saved tensors of the node that produced h
                self._backward(loss, inputs=self._post_params + [h], retain_graph=True)
        finally:
            HF.SlotArena.end_step(self.device)
            PriorLane.enabled = False
            PriorLane.join(self.device)
        self._cut = h
        from ..functional import _PendingReduce
        _PendingReduce.flush(self.device)  # the decoder's weight gradients are complete before their all-reduce starts
        # with a cut only the downstream parameters have their gradient now; the encoder's slots are filled by phase 2
        self._collect_loose_grads(self._post_params if h is not None else None)
        self._logs = {k: v.detach() for k, v in logs.items()}
        self.latents = art["latents"].detach()
        return self._logs

    def _phase2(self):
        """the encoder's backward, from dL/dh that phase 1 left at the cut"""
        h = self._cut
        if h is not None:
            HF.SlotArena.resume(self.device)   # (the encoder's backward adds into slots behind the ones phase 1 took)
            try:
                torch.autograd.backward(h, h.grad, inputs=self._enc_params)
            finally:
                HF.SlotArena.end_step(self.device)
            h.grad = None
            from ..functional import _PendingReduce
            _PendingReduce.flush(self.device)
            # ONLY the encoder's: the decoder's range is under its all-reduce on the side stream right now, and its
            # p.grad (still set from phase 1) holds the rank-local value
            self._collect_loose_grads(self._enc_params)
        self._cut = None
        self.model._last_cut = None
        if self.latent_stats is not None:
            self.latent_stats.update(target_samples=self.latents.flatten(1))

    def _allreduce_split_tail(self):
        lo, hi = self._dec_range
        self.reducer.allreduce_range(0, lo, wait=False)
        self.reducer.allreduce_range(hi, self.gflat.numel(), wait=False)
        self.reducer.join()

    def _adam(self):
        lib = self.lib
        norm = self.gradient_clip_val is not None or self.step_guard == "full"
        if norm:  # max_norm 0: the norm is only reported (out[0] = grad_scale)
            check(lib.otvae_grad_clip_coef(ptr(self.gflat), self.gflat.numel(), self.reducer.grad_scale, self.gradient_clip_val or 0.0,
                                           ptr(self._clip_ws), ptr(self.clip_out), stream()), "otvae_grad_clip_coef")
        if self.ema is not None:
            guarded = self.step_guard is not None
            watch = getattr(self.model, "_last_out3", None) if guarded else None
            self._watch = watch
            check(lib.otvae_adam_step_ema(ptr(self.pflat), ptr(self.gflat), ptr(self.m), ptr(self.v), self.pflat.numel(),
                                          ptr(self.hyper), ptr(self.step_count), self.reducer.grad_scale,
                                          ptr(self.clip_out) if norm else None, ptr(watch), ptr(self.guard) if guarded else None,
                                          ptr(self.rflat) if guarded else None, ptr(self.rbackup) if guarded else None,
                                          0 if (self.rflat is None or not guarded) else self.rflat.numel(),
                                          ptr(self.ema.shadow), self.ema.decay, stream()), "otvae_adam_step_ema")
        elif self.step_guard is not None:
            watch = getattr(self.model, "_last_out3", None)  # [total, recon, prior] of this step (static inside a captured step)
            self._watch = watch
            check(lib.otvae_adam_step_guarded(ptr(self.pflat), ptr(self.gflat), ptr(self.m), ptr(self.v), self.pflat.numel(),
                                              ptr(self.hyper), ptr(self.step_count), self.reducer.grad_scale,
                                              ptr(self.clip_out) if norm else None, ptr(watch), ptr(self.guard),
                                              ptr(self.rflat), ptr(self.rbackup), 0 if self.rflat is None else self.rflat.numel(),
                                              stream()), "otvae_adam_step_guarded")
        elif norm:
            check(lib.otvae_adam_step_dev(ptr(self.pflat), ptr(self.gflat), ptr(self.m), ptr(self.v), self.pflat.numel(),
                                          ptr(self.hyper), ptr(self.step_count), ptr(self.clip_out), stream()),
                  "otvae_adam_step_dev")
        else:
            check(lib.otvae_adam_step(ptr(self.pflat), ptr(self.gflat), ptr(self.m), ptr(self.v), self.pflat.numel(),
                                      ptr(self.hyper), ptr(self.step_count), self.reducer.grad_scale, stream()),
                  "otvae_adam_step")

    @property
    def skipped_steps(self) -> int:
        """steps the device-side guard refused so far (one host read)"""
        return int(self.guard[0].item())

    def _allreduce(self):
        self.reducer.allreduce()

    def _eager_step(self):
        if self.dp_overlap:
            logs = self._phase1()
            self.reducer.allreduce_range(*self._dec_range, wait=False)  # flies under phase 2
            self._phase2()
            self._allreduce_split_tail()
        else:
            logs = self._forward_backward()
            self._allreduce()
        self._adam()
        return logs

    # -- graph capture -------------------------------------------------------------------------------------------
    def _loss_vector(self, logs):
        out3 = getattr(self.model, "_last_out3", None)  # the fused nelbo kernel's [total, recon, prior] vector
        if out3 is not None:
            return out3
        return torch.stack([logs["train/loss/total"].detach(), logs["train/loss/recon"].detach(),
                            logs["train/loss/prior"].detach()])

    def capture(self, warmup: int = 2):
        """Warm-up eagerly on a side stream (allocator + lazy kernel loading), then capture."""
        if self._captured or not self.use_graph:
            return
        self._gs.release()   # (graphs of an earlier attempt that raised half-way)
        snap = (self.pflat.clone(), self.m.clone(), self.v.clone(), self.step_count.clone(), self.guard.clone())
        # everything else a step mutates: every buffer of the model (BatchNorm running statistics, EMA embeddings of a
        # ConditionalGaussianPrior ...), parameters outside the flat buffer (frozen ones an EMA rewrites), the dropout
        # key counters, and the running statistics of the latent operator (which may hold earlier eager steps' samples)
        flat_ids = {id(p) for p in self.params}
        state = [t for t in self.model.buffers()] + [p.data for p in self.model.parameters() if id(p) not in flat_ids]
        state += [mod.__dict__["_dropout_key"] for mod in self.model.modules() if isinstance(mod.__dict__.get("_dropout_key"), Tensor)]
        if self.latent_stats is not None:
            state += [t for t in self.latent_stats.buffers()] + [p.data for p in self.latent_stats.parameters()]
        if self.ema is not None:
            state.append(self.ema.shadow)
        state_snap = [t.clone() for t in state]
        s = _lib.fresh_stream(self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        # "thread_local": only this thread's calls are policed while the stream captures.  Under the default ("global")
        # an event query from another thread -- the RCCL watchdog polling the collectives of earlier steps -- is an
        # illegal call that kills the capture, and the process with it, whenever the poll happens to land inside it
        mode = dict(capture_error_mode=os.environ.get("OTVAE_CAPTURE_ERROR_MODE", "thread_local"))
        gs = self._gs
        mode["stream"] = _lib.fresh_stream(self.device)   # a capture stream nothing else of this process aliases (_lib.fresh_stream)
        # capture_guard: graphs parked by engines that died inside an earlier capture are destroyed first, and the cyclic collector
        # stays off until the capture ends (a graph destroyed inside an open capture aborts the process: engine/lifetime.py)
        with capture_guard():
            if self.dp_overlap:
                with torch.cuda.graph(gs.new("fb"), **mode):
                    logs = self._phase1()
                    self._out_static = self._loss_vector(logs)
                pool = gs.get("fb").pool()
                with torch.cuda.graph(gs.new("b2"), pool=pool, **mode):  # the cut tensors live in graph 1's pool
                    self._phase2()
                with torch.cuda.graph(gs.new("opt"), pool=pool, **mode):
                    self._adam()
            elif self.world == 1 and not self.reducer.active and SEGMENT_CALLS > 0 and HF.WGRAD_SIDE_STREAM == 1:
                # a chain of linear graphs + side graphs instead of one graph with a fork per layer (engine/segments.py)
                side = HF._PendingReduce.side_stream(self.device)
                seg = SegmentedStep(self.device, side)
                gs.put("segments", seg)
                dev_ = self.device
                HF._PendingReduce.begin_segments(dev_, seg, may_cut=lambda: not HF.PriorLane.is_open(dev_))
                try:
                    with seg:
                        logs = self._forward_backward()   # its backward pass cuts the graph and ends with a joining cut
                        self._adam()
                        self._out_static = self._loss_vector(logs)
                finally:
                    HF._PendingReduce.end_segments(dev_)
                del seg
            elif self.world == 1 and not self.reducer.active:
                with torch.cuda.graph(gs.new("fb"), **mode):
                    logs = self._forward_backward()
                    self._adam()
                    self._out_static = self._loss_vector(logs)
            else:
                with torch.cuda.graph(gs.new("fb"), **mode):
                    logs = self._forward_backward()
                    self._out_static = self._loss_vector(logs)
                with torch.cuda.graph(gs.new("opt"), **mode):
                    self._adam()
        # the warm-up/capture must not count as training: restore parameters, moments, step and BN buffers
        with torch.no_grad():
            self.pflat.copy_(snap[0]); self.m.copy_(snap[1]); self.v.copy_(snap[2]); self.step_count.copy_(snap[3])
            self.guard.copy_(snap[4])
            for t, v in zip(state, state_snap):
                t.copy_(v)
        torch.cuda.synchronize()
        self._captured = True

    # -- public --------------------------------------------------------------------------------------------------
    def load_batch(self, x: Tensor, eps: Optional[Tensor] = None):
        """Device-to-device copy of the next batch into the static input (and optional explicit eps)."""
        self.x.copy_(x, non_blocking=True)
        if eps is not None:
            self.eps.copy_(eps.reshape(self.eps.shape), non_blocking=True)
        else:
            self._draw_eps()

    def _draw_eps(self) -> None:
        """re-parametrisation noise of the next step from the device-side generator (one launch of this library)"""
        from .. import functional as HF
        from ..utils import hasarg
        if self._rng_key is None:
            self._rng_key = HF.new_rng_key(self.device)
            prior = getattr(self.model, "prior", None)
            self._wants_eps = prior is not None and hasarg(prior, "eps")
        if self._wants_eps:
            HF.normal_fill_(self.eps, self._rng_key)

    def step(self, x: Optional[Tensor] = None, eps: Optional[Tensor] = None, **batch_kwargs) -> Tensor:
        """One optimisation step.  Returns a device tensor [total, recon, prior] (valid until the next step)."""
        if x is not None:
            self.load_batch(x, eps)
        elif eps is None:
            self._draw_eps()
        for k, v in batch_kwargs.items():
            if k not in self.batch_kwargs:
                raise KeyError(f"`{k}` was not declared in HipTrainer(batch_kwargs=...)")
            self.batch_kwargs[k].copy_(v, non_blocking=True)
        annealing = getattr(getattr(self.model, "prior", None), "annealing_steps", 0) > self.n_steps
        if self.use_graph and not annealing:  # the annealing coefficient is a kernel argument: not replayable
            if not self._captured:
                self.capture()
            gs = self._gs
            (gs.get("segments") or gs.get("fb")).replay()
            if self.dp_overlap:
                self.reducer.allreduce_range(*self._dec_range, wait=False)  # side stream, under the encoder's backward
                gs.get("b2").replay()
                self._allreduce_split_tail()
                gs.get("opt").replay()
            elif gs.get("opt") is not None:
                self._allreduce()
                gs.get("opt").replay()
            out = self._out_static
        else:
            logs = self._eager_step()
            out = self._loss_vector(logs)
        self.n_steps += 1
        try:
            self.model.global_step = self.n_steps
        except AttributeError:  # Lightning owns global_step
            pass
        return out

    def close(self) -> None:
        """Releases what the step holds on the device, in a stated order: wait for the reducer's stream, then
        ``GraphSet.release`` (device synchronize, graphs destroyed in reverse capture order -- or parked, should this be called
        while a capture is open), then the step's static tensors (they live in the graphs' private pool, which goes back to the
        allocator with the last of them).  After ``close()`` the process group may be destroyed; the trainer must not be stepped
        again.  An engine that is dropped WITHOUT ``close()`` takes the same path through its finalizer (engine/lifetime.py says
        why the order matters: a graph destroyed while a capture is open aborts the process)."""
        if self.reducer.stream is not None:
            self.reducer.stream.synchronize()
        self._gs.release()
        self._out_static = None
        self._cut = None
        self._logs = None
        self.latents = None
        self._watch = None
        model = self._model_ref()
        if model is not None and hasattr(model, "_last_cut"):
            model._last_cut = None
            model._last_nelbo = None
        self._captured = False
        self.use_graph = False
        check_solver = getattr(getattr(model, "prior", None), "raise_if_starved", None)
        if check_solver is not None:
            check_solver()

    def set_lr(self, lr: float):
        self.hyper[0] = lr
