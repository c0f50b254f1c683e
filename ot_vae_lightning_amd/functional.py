"""Autograd-aware wrappers over the C ABI (``include/otvae.h``): the only place where tensors meet kernels.

Layout contract: every 4-D activation handled here has the reference's logical shape ``[N, C, H, W]`` but
channels-last (NHWC) memory; conv weights keep the logical ``[Cout, Cin, KH, KW]`` shape on HWIO memory.
``as_nhwc`` / ``hwio_weight`` convert foreign tensors once at the boundary.

Nothing here computes on the CPU: tensors must be on the GPU and the HIP library must be loadable.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _lib
from . import ops as _ops  # noqa: F401  (registers torch.ops.otvae.*)
from ._lib import ConvGeom, check, ptr, ptr_array, stream

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ------------------------------------------------------------------------------------------------ layout helpers
def is_nhwc(x: Tensor) -> bool:
    """True if the memory of the logical [N,C,H,W] tensor is dense NHWC (size-1 dims may have any stride)."""
    if x.dim() != 4:
        return False
    n, c, h, w = x.shape
    want = (h * w * c, 1, w * c, c)
    for size, st, ws in zip(x.shape, x.stride(), want):
        if size != 1 and st != ws:
            return False
    return True


def as_nhwc(x: Tensor) -> Tensor:
    """Returns a tensor with the same logical shape/values whose memory is dense NHWC (no copy if it already is)."""
    if is_nhwc(x):
        return x
    return x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def empty_nhwc(n: int, c: int, h: int, w: int, like: Tensor, dtype=None) -> Tensor:
    return torch.empty((n, h, w, c), device=like.device, dtype=dtype or like.dtype).permute(0, 3, 1, 2)


def is_hwio(w: Tensor) -> bool:
    co, ci, kh, kw = w.shape
    want = (1, co, kw * ci * co, ci * co)
    return all(size == 1 or st == ws for size, st, ws in zip(w.shape, w.stride(), want))


def hwio_weight(w: Tensor) -> Tensor:
    """Logical OIHW tensor on HWIO memory (no copy if it already is)."""
    if is_hwio(w):
        return w
    return w.permute(2, 3, 1, 0).contiguous().permute(3, 2, 0, 1)


def new_hwio(co: int, ci: int, kh: int, kw: int, device=None, dtype=torch.float32) -> Tensor:
    return torch.empty((kh, kw, ci, co), device=device, dtype=dtype).permute(3, 2, 0, 1)


def _grad_buffer(param: Optional[Tensor], like: Tensor) -> Tensor:
    """Where a parameter gradient is written: the trainer's flat-buffer slot if the parameter has one
    (``param._otvae_grad_view()``), else a fresh tensor with the parameter's strides."""
    if param is not None:
        getter = getattr(param, "_otvae_grad_view", None)
        if getter is not None:
            return getter()
    return torch.empty_strided(like.shape, like.stride(), device=like.device, dtype=like.dtype)


def _geom(x: Tensor, weight: Tensor, stride: int, pad: int, up: int) -> Tuple[ConvGeom, int, int]:
    n, cs, hs, ws = x.shape
    cn, ci, kh, kw = weight.shape
    if ci != cs:
        raise ValueError(f"conv: input has {cs} channels but weight expects {ci}")
    ho = (hs * up + 2 * pad - kh) // stride + 1
    wo = (ws * up + 2 * pad - kw) // stride + 1
    return ConvGeom(n, hs, ws, cs, up, ho, wo, cn, kh, kw, stride, pad), ho, wo


_PLAN_CACHE: dict = {}


def _conv_plan(g: ConvGeom, has_bias: bool):
    """Workspace shapes the library wants for this geometry -- (forward statistics partials, their row length, weight-gradient
    partials, data-gradient BatchNorm partials, their row length, dead-tap mask) -- asked once per geometry: four C calls per
    layer and pass otherwise, on a host-bound eagerly issued step."""
    env = os.environ  # the kernel-selection switches are read per call by the library and change the plan
    key = (g.N, g.Hs, g.Ws, g.Cs, g.up, g.Ho, g.Wo, g.Cn, g.KH, g.KW, g.stride, g.pad, has_bias,
           "OTVAE_NO_TILE" in env, "OTVAE_NO_WTILE" in env, "OTVAE_TILE_ALL" in env, "OTVAE_WTILE_ALL" in env)
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        lib = _lib.load()
        p_s, ld, p_w, p_d, cp, dead = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0), C.c_uint32(0)
        check(lib.otvae_conv_fwd_stats_ws(C.byref(g), C.byref(p_s), C.byref(ld)), "otvae_conv_fwd_stats_ws")
        check(lib.otvae_conv_bwd_weight_ws(C.byref(g), int(has_bias), C.byref(p_w)), "otvae_conv_bwd_weight_ws")
        check(lib.otvae_conv_bwd_data_ws(C.byref(g), C.byref(p_d), C.byref(cp)), "otvae_conv_bwd_data_ws")
        check(lib.otvae_conv_dead_taps(C.byref(g), C.byref(dead)), "otvae_conv_dead_taps")
        plan = _PLAN_CACHE[key] = (p_s.value, ld.value, p_w.value, p_d.value, cp.value, dead.value)
    return plan


# ------------------------------------------------------------------------------------------------ BatchNorm stats
class BNBranch:
    """What one ConvLayer contributes to a (possibly shared) BatchNorm statistics pass."""
    __slots__ = ("gamma", "beta", "running_mean", "running_var", "num_batches_tracked")

    def __init__(self, gamma, beta, running_mean=None, running_var=None, num_batches_tracked=None):
        self.gamma, self.beta = gamma, beta
        self.running_mean, self.running_var, self.num_batches_tracked = running_mean, running_var, num_batches_tracked


# OTVAE_BN_SLOTS.  Inside a training engine's step the cross-block sums of BatchNorm can go to STATISTIC SLOTS (csrc/common.h: int64
# fixed-point limbs added with integer atomics: order-independent, bit-reproducible) and the consumer kernel folds the finalize
# arithmetic into its own prologue, so that the finalize launch disappears from the dependent chain:
#   1 (default)  the forward pass: a layer output's (sum, sum of squares) -> the next conv / attention-stage launch.  30 launches
#                fewer in the MNIST step, 2.67 -> 2.59 ms (same box, profiles/r04_bn_slots_ab.txt);
#   2            also the backward pass: a data gradient's (sum gv, sum gv * xhat) -> otvae_bn_bwd_apply_slots (finalize + apply in one
#                launch).  30 more launches gone, but SLOWER than 1 (2.62 ms): the backward finalize launches were hidden behind the
#                weight-gradient stream, the slot atomics and the per-block prologue are not.  Kept, tested, off;
#   0            per-block fp64 partials + finalize launches everywhere (rounds 1-3).
# The slots of a step come from one arena that the engine zeroes in its step-begin launch; outside an engine's step (plain module
# calls) the partial route is taken.
BN_SLOTS_MODE = int(os.environ.get("OTVAE_BN_SLOTS", "1"))


class Slots:
    """one tensor's statistic slots: int64 view of the arena, channel stride, slots in use"""
    __slots__ = ("buf", "ld", "n")

    def __init__(self, buf: Tensor, ld: int, n: int):
        self.buf, self.ld, self.n = buf, int(ld), int(n)


class SlotArena:
    """int64 words for the statistic slots of one step (per device).  ``begin_step`` rewinds it and zeroes what earlier steps used
    (in the engine's own step-begin launch when ``fused_zero``); ``take`` hands out ranges in call order -- a captured step bakes the
    addresses its warm-up steps used -- and zeroes on demand whatever lies beyond the part already zeroed (first step, growing shapes)."""
    _state: dict = {}    # device -> [buf, offset, active, high-water mark, zeroed-up-to]
    WORDS = 1 << 20       # 8 MiB; a tensor takes nslots * 4 * ld + 2 words

    @staticmethod
    def _zero(t: Tensor) -> None:
        if t.numel():
            check(_lib.load().otvae_zero_words(ptr(t), t.numel(), stream()), "otvae_zero_words")

    @staticmethod
    def ensure(device):
        st = SlotArena._state.get(device)
        if st is None and BN_SLOTS_MODE != 0 and not torch.cuda.is_current_stream_capturing():   # (never allocated from a graph's pool)
            st = SlotArena._state[device] = [torch.zeros(SlotArena.WORDS, device=device, dtype=torch.int64), 0, False, 0, 0]
        return st

    @staticmethod
    def begin_step(device, fused_zero: bool = False) -> Optional[Tensor]:
        """Returns the range the caller zeroes in a launch of its own (``fused_zero``), else None."""
        st = SlotArena.ensure(device) if BN_SLOTS_MODE != 0 else None
        if st is None:
            return None
        st[1], st[2], st[4] = 0, True, st[3]
        if st[3] == 0:
            return None
        rng = st[0][: st[3]]
        if fused_zero:
            return rng
        SlotArena._zero(rng)
        return None

    @staticmethod
    def resume(device, rezero: bool = False) -> None:
        """The step continues (backward pass) behind what ``begin_step`` .. ``end_step`` handed out.  ``rezero``: the ranges from here
        on are zeroed again first -- a captured backward pass that may be replayed more than once per forward replay."""
        st = SlotArena._state.get(device)
        if st is None or BN_SLOTS_MODE == 0:
            return
        st[2] = True
        if rezero and st[4] > st[1]:
            SlotArena._zero(st[0][st[1]: st[4]])

    @staticmethod
    def end_step(device) -> None:
        st = SlotArena._state.get(device)
        if st is not None:
            st[2] = False

    # producer blocks per slot address: atomics on one address are performed one after the other (~0.1 us each), so the tail a
    # producer kernel pays is about this many tenths of a microsecond (profiles/r04_bn_slots_ab.txt)
    ADDS_PER_SLOT = int(os.environ.get("OTVAE_BN_SLOT_ADDS", "16"))

    @staticmethod
    def pick(blocks: int) -> int:
        """slots in use for a producer of ``blocks`` blocks: a power of two in 4 .. 64"""
        n = 4
        while n < 64 and n * SlotArena.ADDS_PER_SLOT < blocks:
            n *= 2
        return n

    @staticmethod
    def take(device, ld: int, blocks: int) -> Optional[Slots]:
        st = SlotArena._state.get(device)
        if st is None or not st[2] or ld > 1024:
            return None
        n = SlotArena.pick(blocks)
        words = (int(_lib.load().otvae_bn_slots_words(int(ld), n)) + 1) // 2 * 2   # 16-byte aligned ranges
        end = st[1] + words
        if end > st[0].numel():
            return None   # (full: the caller keeps the partial route)
        if end > st[4]:
            SlotArena._zero(st[0][st[4]: end])
            st[4] = end
        v = st[0][st[1]: end]
        st[1] = end
        st[3] = max(st[3], end)
        return Slots(v, ld, n)


class PendingFold:
    """Statistics of x that sit in slots and have not been turned into (mean, invstd, scale, shift) yet: the consumer launch folds that
    into its prologue and ITS first block fills the four tensors (``fold_struct``); a consumer that cannot calls ``materialize``."""
    __slots__ = ("slots", "count", "branches", "update", "mean", "invstd", "scales", "shifts", "done")

    def __init__(self, slots: Slots, count, branches, update, mean, invstd, scales, shifts):
        self.slots, self.count, self.branches, self.update = slots, count, list(branches), update
        self.mean, self.invstd, self.scales, self.shifts, self.done = mean, invstd, scales, shifts, False

    def fold_struct(self, j: int, publish: bool) -> "_lib.BnFold":
        br = self.branches[j]
        f = _lib.BnFold()
        f.slots, f.ld, f.nslots = ptr(self.slots.buf), self.slots.ld, self.slots.n
        f.count, f.eps, f.momentum = int(self.count), BN_EPS, BN_MOMENTUM
        f.gamma, f.beta = ptr(br.gamma), ptr(br.beta)
        if self.update:
            f.running_mean, f.running_var, f.num_batches_tracked = ptr(br.running_mean), ptr(br.running_var), ptr(br.num_batches_tracked)
        if publish:
            f.mean_out, f.invstd_out = ptr(self.mean), ptr(self.invstd)
        f.scale_out, f.shift_out = ptr(self.scales[j]), ptr(self.shifts[j])
        return f

    def materialize(self) -> None:
        if self.done:
            return
        n = len(self.branches)
        arr = (_lib.BnFold * n)(*[self.fold_struct(j, j == 0) for j in range(n)])
        check(_lib.load().otvae_bn_finalize_slots(n, arr, int(self.mean.numel()), stream()), "otvae_bn_finalize_slots")
        self.done = True


@torch.no_grad()
def bn_batch_stats(x: Tensor, branches: Sequence[BNBranch], update_running: bool = True, allow_fold: bool = False):
    """Training-mode statistics of x (NHWC) shared by all ``branches``: returns mean, invstd and one
    (scale, shift) pair per branch; updates the running buffers like nn.BatchNorm2d (networks/cnn.py:122).
    If x was produced by one of our conv kernels, its per-channel partial sums were already written by that kernel's
    epilogue (``x._otvae_stats``) and no pass over x is needed.
    ``allow_fold``: the caller's kernel can fold the finalize arithmetic into its prologue -- then, when the sums sit in statistic
    slots, a fifth value (a ``PendingFold``) is returned and the four tensors are filled by that kernel's first block."""
    lib = _lib.load()
    n, c, h, w = x.shape
    m = n * h * w
    mean = torch.empty(c, device=x.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    scales = [torch.empty_like(mean) for _ in branches]
    shifts = [torch.empty_like(mean) for _ in branches]
    nb = len(branches)
    upd = update_running
    pre = getattr(x, "_otvae_stats", None)
    slots = None
    if isinstance(pre, Slots):      # the producer's epilogue used the statistic slots
        slots = pre
    elif pre is None and 1 <= nb <= 2:
        slots = SlotArena.take(x.device, c, lib.otvae_bn_stats_nparts(m, c))
        if slots is not None:
            check(lib.otvae_bn_stats_slots(ptr(x), m, c, ptr(slots.buf), slots.ld, slots.n, stream()), "otvae_bn_stats_slots")
    if slots is not None:
        fold = PendingFold(slots, m, branches, upd, mean, invstd, scales, shifts)
        if allow_fold and 1 <= nb <= 2:
            return mean, invstd, scales, shifts, fold
        fold.materialize()
        return (mean, invstd, scales, shifts, None) if allow_fold else (mean, invstd, scales, shifts)
    if pre is not None:
        partial, p, ld = pre
    else:
        p = lib.otvae_bn_stats_nparts(m, c)
        ld = c
        partial = torch.empty((p, 2, c), device=x.device, dtype=torch.float64)
        check(lib.otvae_bn_stats(ptr(x), m, c, ptr(partial), stream()), "otvae_bn_stats")
    check(lib.otvae_bn_finalize(
        ptr(partial), p, ld, m, c, BN_EPS, BN_MOMENTUM, ptr(mean), ptr(invstd), nb,
        ptr_array([b.gamma for b in branches]), ptr_array([b.beta for b in branches]),
        ptr_array([b.running_mean if upd else None for b in branches]),
        ptr_array([b.running_var if upd else None for b in branches]),
        ptr_array([b.num_batches_tracked if upd else None for b in branches]),
        ptr_array(scales), ptr_array(shifts), stream()), "otvae_bn_finalize")
    return (mean, invstd, scales, shifts, None) if allow_fold else (mean, invstd, scales, shifts)


@torch.no_grad()
def bn_eval_affine(branch: BNBranch):
    """Inference-mode BatchNorm as a fixed affine (running statistics)."""
    invstd = torch.rsqrt(branch.running_var + BN_EPS)
    scale = branch.gamma * invstd
    shift = branch.beta - branch.running_mean * scale
    return branch.running_mean.clone(), invstd, scale, shift


# ------------------------------------------------------------------------------------------------ side stream
_DENSE_REDUCE = os.environ.get("OTVAE_DENSE_REDUCE", "0") == "1"  # A/B switch: ignore the dead-tap information


# OTVAE_WGRAD_STREAM: 1 (default) = weight-gradient jobs go to the side stream while a step is being captured (a replayed step
# is device-bound; an eagerly issued one is host-bound and would only pay for the extra events); 2 = always; 0 = never (the
# jobs share the data-gradient jobs' launch) -- A/B switch
WGRAD_SIDE_STREAM = int(os.environ.get("OTVAE_WGRAD_STREAM", "1"))
# backward calls (ConvBlock stages) whose weight-gradient jobs share one fork: every fork costs the launch stream ~4.5 us
WGRAD_GROUP = max(1, int(os.environ.get("OTVAE_WGRAD_GROUP", "1")))
WGRAD_REDUCE_GROUP = max(1, int(os.environ.get("OTVAE_WGRAD_REDUCE_GROUP", "16")))  # layers per side-stream partial reduction
WGRAD_STREAMS = max(1, int(os.environ.get("OTVAE_WGRAD_STREAMS", "1")))  # side streams taking the forks in turn


class _PendingReduce:
    """Weight-gradient partials of the current backward pass whose destination is a trainer-owned flat gradient
    buffer, reduced together by ``flush`` (queued as an autograd-engine callback and called by the trainer before the
    optimizer).  Destinations are kept as raw addresses: holding the tensors would make autograd clone them.

    Nothing downstream of a layer's weight-gradient job reads its output before ``flush``, so those jobs are off the backward
    pass's critical path (data gradient -> BatchNorm backward -> next layer): they are issued on a second HIP stream
    (``side_stream``), forked from the launch stream where the layer's output gradient is complete and joined in ``flush``.  The
    kernels of a step are latency-bound at 1-3 waves per SIMD, so the two streams share the chip almost for
    free; in a captured step the fork / join become edges of the hipGraph.  Every tensor a side-stream job reads is held
    here until the join (the caching allocator would otherwise hand its memory to a later launch-stream kernel)."""
    _state = {}
    _side = {}
    _held = {}
    _wq = {}       # device -> [queued weight-gradient jobs, backward calls they came from]
    _side_prologue = {}  # device -> callables run once on the side stream at the next fork (see issue)
    _defer = {}          # device -> True while a training engine's captured forward pass runs: defer_to_side accepts work
    _forked = {}   # device -> the side stream has work of this backward pass
    _reduced = {}  # device -> leading entries of _state whose partials the side stream has already reduced

    @staticmethod
    def side_stream(device) -> "torch.cuda.Stream":
        """Next side stream of the device's pool (round robin: successive forks may run beside each other, too)."""
        pool = _PendingReduce._side.get(device)
        if pool is None:
            pool = _PendingReduce._side[device] = [[_lib.fresh_stream(device) for _ in range(WGRAD_STREAMS)], 0]
        pool[1] = (pool[1] + 1) % len(pool[0])
        return pool[0][pool[1]]

    @staticmethod
    def queue_weight_jobs(device, jobs, n, tensors) -> bool:
        """Defers the weight-gradient jobs of one backward call; ``tensors`` (everything the jobs read or write) stay alive until
        flush.  True when WGRAD_GROUP calls have accumulated and the caller should fork (``fork_point``) and ``issue``."""
        q = _PendingReduce._wq.setdefault(device, [[], 0])
        q[0].extend(_lib.ConvJob.from_buffer_copy(jobs[i]) for i in range(n))
        q[1] += 1
        _PendingReduce._held.setdefault(device, []).extend(t for t in tensors if t is not None)
        return q[1] >= WGRAD_GROUP

    @staticmethod
    def defer_to_side(device, fn, *tensors) -> bool:
        """Queues ``fn`` (launches on the then-current stream) for the side stream's next fork if a training engine allows it for this
        step (``_defer``); ``tensors`` = everything it touches, held until the join.  False: the caller launches in line."""
        if not _PendingReduce._defer.get(device):
            return False
        _PendingReduce._side_prologue.setdefault(device, []).append(fn)
        _PendingReduce._held.setdefault(device, []).extend(t for t in tensors if t is not None)
        return True

    @staticmethod
    def run_deferred_inline(device) -> None:
        """what no fork has taken (a pass without weight-gradient forks) runs on the current stream"""
        for pro in _PendingReduce._side_prologue.pop(device, None) or ():
            pro()

    @staticmethod
    def fork_point(device):
        if device in _PendingReduce._segment:
            return True  # segmented capture: the order between the streams is set between graph launches, not by an event here
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        return ev

    # Segmented capture (engine/segments.py): while a SegmentedStep is capturing on `device` the queued jobs are not launched from
    # a fork inside the capture; they accumulate into the segment's side work, and every `segment_calls` backward calls the
    # launch stream's graph is cut there.  _segment[device] = [SegmentedStep, jobs, reduce entries, backward calls, may_cut()]
    _segment = {}

    @staticmethod
    def begin_segments(device, step, may_cut=None) -> None:
        _PendingReduce._segment[device] = [step, [], [], 0, may_cut]

    @staticmethod
    def end_segments(device) -> None:
        _PendingReduce._segment.pop(device, None)

    @staticmethod
    def _segment_work(jobs, entries):
        """closure issuing one segment's weight-gradient jobs + the partial reductions of its layers on the current stream"""
        n = len(jobs)
        arr = (_lib.ConvJob * n)(*jobs) if n else None

        def work():
            if n:
                check(_lib.load().otvae_conv_multi(n, arr, stream()), "otvae_conv_multi(backward, weights, segment)")
            if entries:
                _PendingReduce._reduce(entries, stream())
        return work

    @staticmethod
    def _segment_close(device, final: bool) -> None:
        seg = _PendingReduce._segment[device]
        step, jobs, entries, _, may_cut = seg
        if not final and may_cut is not None and not may_cut():
            return  # a forked lane of the capture is still open (the prior's): this graph cannot end here
        st, done = _PendingReduce._state.get(device, []), _PendingReduce._reduced.get(device, 0)
        entries = entries + st[done:]
        _PendingReduce._reduced[device] = len(st)
        work = _PendingReduce._segment_work(list(jobs), list(entries)) if (jobs or entries) else None
        seg[1], seg[2], seg[3] = [], [], 0
        step.cut(work, join=final)

    @staticmethod
    def issue(device, ev):
        """Launches the queued weight-gradient jobs on the side stream, ordered after the launch stream's work up to ``ev``."""
        q = _PendingReduce._wq.get(device)
        if not q or not q[0]:
            return
        seg = _PendingReduce._segment.get(device)
        if seg is not None:
            from .engine.segments import SEGMENT_CALLS
            seg[1].extend(q[0])
            seg[3] += q[1]
            if JOB_TRACE is not None:
                n_ = len(q[0])
                _trace_jobs((_lib.ConvJob * n_)(*q[0]), n_)
            q[0].clear()
            q[1] = 0
            if seg[3] >= SEGMENT_CALLS:
                _PendingReduce._segment_close(device, final=False)
            return
        side = _PendingReduce.side_stream(device)
        side.wait_event(ev)
        pros = _PendingReduce._side_prologue.pop(device, None)
        if pros:
            # work of the step that nothing on the launch stream waits for (the loss VALUE, the trainer's latent-statistics update): it
            # rides on this fork, in front of the weight-gradient jobs -- no graph branch of its own (a lane of its own cost more than
            # it took off the chain, engine/trainer.py)
            with torch.cuda.stream(side):
                for pro in pros:
                    pro()
        n = len(q[0])
        arr = (_lib.ConvJob * n)(*q[0])
        check(_lib.load().otvae_conv_multi(n, arr, C.c_void_p(side.cuda_stream)), "otvae_conv_multi(backward, weights)")
        if JOB_TRACE is not None:
            _trace_jobs(arr, n)
        q[0].clear()
        q[1] = 0
        _PendingReduce._forked[device] = True
        # every queued layer's job is now in the side stream: reduce the finished layers there as the pass goes, so that only
        # the last few layers' partials are left for the launch stream after the join
        st, done = _PendingReduce._state.get(device, []), _PendingReduce._reduced.get(device, 0)
        if len(st) - done >= WGRAD_REDUCE_GROUP:
            _PendingReduce._reduce(st[done:], C.c_void_p(side.cuda_stream))
            _PendingReduce._reduced[device] = len(st)

    @staticmethod
    def add(device, partial, p, k, kp, cn, gw, gb, cs, dead):
        st = _PendingReduce._state.setdefault(device, [])
        if not st:
            torch.autograd.Variable._execution_engine.queue_callback(lambda dev=device: _PendingReduce.flush(dev))
        st.append((partial, p, k, kp, cn, gw, gb, cs, dead))

    @staticmethod
    def reset(device) -> None:
        """Forgets whatever an interrupted backward pass left behind (an exception between ``add`` and ``flush``): the queued jobs
        hold raw addresses of tensors that the next pass no longer owns.  Joins the side streams first."""
        pool = _PendingReduce._side.get(device)
        if pool is not None and _PendingReduce._forked.get(device):
            for side in pool[0]:
                torch.cuda.current_stream(device).wait_stream(side)
        _PendingReduce._forked[device] = False
        for table in (_PendingReduce._state, _PendingReduce._held):
            if table.get(device):
                table[device].clear()
        q = _PendingReduce._wq.get(device)
        if q:
            q[0].clear()
            q[1] = 0
        _PendingReduce._side_prologue.pop(device, None)
        _PendingReduce._reduced[device] = 0

    @staticmethod
    def _reduce(entries, stream_ptr):
        lib = _lib.load()
        n = len(entries)
        ia = lambda i: (C.c_int * n)(*[e[i] for e in entries])  # noqa: E731
        def raw(vals):  # host array of raw device addresses (gradient slots live in the trainer's flat buffer)
            arr = (C.c_void_p * n)()
            for i, v in enumerate(vals):
                arr[i] = v
            return arr

        check(lib.otvae_wgrad_reduce_batched(n, ptr_array([e[0] for e in entries]), ia(1), ia(2), ia(3), ia(4),
                                             raw([e[5] for e in entries]), raw([e[6] for e in entries]), ia(7),
                                             (C.c_uint32 * n)(*[e[8] for e in entries]), stream_ptr),
              "otvae_wgrad_reduce_batched")

    @staticmethod
    def flush(device):
        st = _PendingReduce._state.get(device)
        if not st:
            return
        held = _PendingReduce._held.get(device)
        q = _PendingReduce._wq.get(device)
        if q and q[0]:  # the tail of the backward pass
            _PendingReduce.issue(device, _PendingReduce.fork_point(device))
        seg = _PendingReduce._segment.get(device)
        if seg is not None:
            # the last segment's side work; the launch stream's next graph (the optimizer's) starts behind the side stream.  The
            # tensors the side graphs touch stay held until the step's capture is over (SegmentedStep.__exit__ captures them)
            _PendingReduce._segment_close(device, final=True)
            _PendingReduce._reduced[device] = 0
            st.clear()
            seg[0]._held = list(held) if held else []
            if held:
                held.clear()
            return
        if _PendingReduce._forked.get(device):  # join: the partials below are complete once the side stream's jobs are
            for side in _PendingReduce._side[device][0]:
                torch.cuda.current_stream(device).wait_stream(side)
            _PendingReduce._forked[device] = False
        done = _PendingReduce._reduced.get(device, 0)
        if done < len(st):
            _PendingReduce._reduce(st[done:], stream())
        _PendingReduce._reduced[device] = 0
        st.clear()
        if held:
            held.clear()


# ------------------------------------------------------------------------------------------------ prior lane
# OTVAE_PRIOR_STREAM: 1 (default) = inside a training engine's CAPTURED step the prior's optimal-transport work (Sinkhorn cost +
# solve + read-out, or batch statistics + eigendecomposition + W2 tail) and the loss vector run on a stream of their own beside
# the decoder's forward and backward pass; 2 = also in the engine's eagerly issued steps; 0 = in line on the launch stream (the
# round-2 order) -- A/B switch.  Nothing in the decoder depends on that work (it needs the latents, which the prior passes
# through), and the decoder's backward does not need the loss VALUE (nelbo's gradient is 2 (pred - target) / numel and a
# constant for the prior term), so the launch stream meets the lane again only where the prior's own backward starts.
PRIOR_SIDE_STREAM = int(os.environ.get("OTVAE_PRIOR_STREAM", "1"))


class PriorLane:
    """Fork / join bookkeeping of the prior's side stream (one per device).  Only a training engine switches it on
    (``PriorLane.enabled``): it is the engine that guarantees the join (``join`` at the prior's backward and again after the
    backward pass) before anything reads what the lane produced.

    Memory: work issued inside ``with lane`` allocates from the LANE stream's pool (``torch.cuda.stream``), so a temporary freed
    when an operator returns can only be handed to a later allocation on the same stream -- ordered behind its last use.  The
    autograd nodes above were created on the launch stream and run their backward there."""
    enabled = False          # set by HipTrainer around the step it issues
    _streams: dict = {}
    _open: dict = {}         # device -> event after the lane's last piece of work (None: nothing forked)
    _held: dict = {}         # device -> launch-stream tensors a lane kernel reads: kept alive until the join

    @staticmethod
    def hold(device, *tensors) -> None:
        """Operands of a lane kernel that were allocated on the LAUNCH stream must outlive that kernel: a temporary (e.g. the
        channels-last copy of a ViT's reconstruction handed to the loss kernel) that autograd releases when the node's own backward
        has run would go back to the launch stream's pool and be handed to the next backward tensor while the lane still reads
        it (found in round 3: recon came out as 2e-9 in a captured ViT + SinkhornPrior step)."""
        PriorLane._held.setdefault(device, []).extend(t for t in tensors if t is not None)

    @staticmethod
    def active(device) -> bool:
        if not PriorLane.enabled or PRIOR_SIDE_STREAM == 0 or device.type != "cuda":
            return False
        return PRIOR_SIDE_STREAM == 2 or torch.cuda.is_current_stream_capturing()

    @staticmethod
    def stream(device) -> "torch.cuda.Stream":
        st = PriorLane._streams.get(device)
        if st is None:
            st = PriorLane._streams[device] = _lib.fresh_stream(device)
        return st

    @staticmethod
    def is_open(device) -> bool:
        return PriorLane._open.get(device) is not None

    class _Section:
        def __init__(self, device):
            self.device = device
            self.ctx = None

        def __enter__(self):
            lane = PriorLane.stream(self.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))   # everything the section reads is complete here
            lane.wait_event(ev)
            self.ctx = torch.cuda.stream(lane)
            self.ctx.__enter__()
            return lane

        def __exit__(self, *exc):
            lane = PriorLane.stream(self.device)
            done = torch.cuda.Event()
            done.record(lane)
            PriorLane._open[self.device] = done
            return self.ctx.__exit__(*exc)

    @staticmethod
    def section(device) -> "PriorLane._Section":
        """``with PriorLane.section(dev):`` -- the body is issued on the lane, ordered after the launch stream's work so far"""
        return PriorLane._Section(device)

    @staticmethod
    def join(device) -> None:
        """the launch stream waits for the lane's work (no-op when nothing is outstanding)"""
        done = PriorLane._open.get(device)
        if done is not None:
            torch.cuda.current_stream(device).wait_event(done)
            PriorLane._open[device] = None
        held = PriorLane._held.get(device)
        if held:
            held.clear()


# ------------------------------------------------------------------------------------------------ fused ConvLayer(s)
class ConvSpec:
    """Static description of one ConvLayer branch (everything that is not a tensor)."""
    __slots__ = ("stride", "pad", "up", "relu", "has_norm", "has_bias", "has_residual", "out_stats")

    def __init__(self, stride, pad, up, relu, has_norm, has_bias, has_residual=False, out_stats=False):
        self.stride, self.pad, self.up, self.relu = stride, pad, up, relu
        self.has_norm, self.has_bias, self.has_residual = has_norm, has_bias, has_residual
        self.out_stats = out_stats  # also emit per-channel partial sums of the output (next layer's BatchNorm)


class _ConvBNFn(torch.autograd.Function):
    """1 or 2 ConvLayers reading the same input x (ConvBlock.block[0] and ConvBlock.skip normalise and convolve the
    same tensor, networks/cnn.py:311-335).  Per branch the tensor inputs are (weight, bias, gamma, beta, residual),
    missing ones passed as None."""

    @staticmethod
    def forward(ctx, x, specs, stats, params_ref, stats_out, *tensors):
        ctx.specs, ctx.stats, ctx.params_ref = specs, stats, params_ref
        outs, ctx.geoms, out_stats = conv_forward_launch(x, specs, stats, tensors)
        stats_out.extend(out_stats)
        ctx.save_for_backward(x, *tensors)
        return outs[0] if len(specs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, *gys):
        saved = ctx.saved_tensors
        dx, per = conv_backward_launch(saved[0], saved[1:], ctx.specs, ctx.geoms, ctx.stats, ctx.params_ref, gys,
                                       ctx.needs_input_grad[0])
        out: List[Optional[Tensor]] = [dx, None, None, None, None]
        for gw, gb, dgam, dbet, gres in per:
            out += [gw, gb, dgam, dbet, gres]
        return tuple(out)


JOB_TRACE: Optional[list] = None  # measurement hook (bench.py): when a list, every otvae_conv_multi call appends its jobs here


def _trace_jobs(jobs, njobs):
    lib = _lib.load()
    mask, ut = C.c_uint32(0), C.c_int(0)
    check(lib.otvae_conv_multi_last(C.byref(mask), C.byref(ut)), "otvae_conv_multi_last")
    desc = []
    for i in range(njobs):
        jb = jobs[i]
        g = jb.geom
        desc.append(dict(kind=int(jb.kind), relu=int(jb.relu), has_bias=int(jb.has_bias), has_norm=bool(jb.scale), bn_sums=bool(jb.mean),
                         geom=tuple(int(getattr(g, f)) for f, _ in ConvGeom._fields_)))
    JOB_TRACE.append(dict(jobs=desc, packed_mask=int(mask.value), uniform_tap=int(ut.value)))


def conv_forward_launch(x, specs, stats, tensors):
    """The launches of 1 or 2 ConvLayer branches reading the same x (one ``otvae_conv_multi`` call).  ``tensors`` holds
    (weight, bias, gamma, beta, residual) per branch.  Returns (outputs, geometries, per-output statistics partials)."""
    lib = _lib.load()
    nbr = len(specs)
    mean, invstd, scales, shifts, training = stats[:5]
    fold = stats[5] if len(stats) > 5 else None     # PendingFold: the BatchNorm finalize of x rides in this launch's prologue
    if fold is not None and (fold.done or x.shape[1] > 1024):
        fold.materialize()
        fold = None
    outs, geoms, out_stats = [], [], []
    jobs = (_lib.ConvJob * nbr)()
    keep = []
    jn = 0   # index among the normalised branches (= the PendingFold's branch order)
    for b, sp in enumerate(specs):
        w, bias, gamma, beta, res = tensors[5 * b: 5 * b + 5]
        g, ho, wo = _geom(x, w, sp.stride, sp.pad, sp.up)
        y = empty_nhwc(x.shape[0], w.shape[0], ho, wo, x)
        part, st, slots_out = None, None, None
        if sp.out_stats:
            p_s, ld = _conv_plan(g, bias is not None)[:2]
            slots_out = SlotArena.take(x.device, ld, p_s)
            if slots_out is not None:
                st = slots_out
            else:
                part = torch.empty((p_s, 2, ld), device=x.device, dtype=torch.float64)
                st = (part, p_s, ld)
        jb = jobs[b]
        jb.kind, jb.relu, jb.geom = _lib.JOB_FWD, int(sp.relu), g
        jb.x = ptr(x)
        if sp.has_norm and fold is not None:
            jb.fold = fold.fold_struct(jn, jn == 0)
        else:
            jb.scale = ptr(scales[b]) if sp.has_norm else None
            jb.shift = ptr(shifts[b]) if sp.has_norm else None
        jn += 1 if sp.has_norm else 0
        jb.w, jb.bias, jb.residual, jb.y, jb.stat_partial = ptr(w), ptr(bias), ptr(res), ptr(y), ptr(part)
        if slots_out is not None:
            jb.stat_slots, jb.stat_nslots = ptr(slots_out.buf), slots_out.n
        keep.append(part)
        outs.append(y)
        geoms.append(g)
        out_stats.append(st)
    # both branches of a ConvBlock read the same x and are independent: one launch (otvae_conv_multi)
    check(lib.otvae_conv_multi(nbr, jobs, stream()), "otvae_conv_multi(forward)")
    if fold is not None:
        fold.done = True
    if JOB_TRACE is not None:
        _trace_jobs(jobs, nbr)
    return outs, geoms, out_stats


def conv_backward_launch(x, tensors, specs, geoms, stats, params_ref, gys, need_dx, pre_dgrad=None, fork="now"):
    """Weight / bias / BatchNorm / data gradients of 1 or 2 ConvLayer branches (one ``otvae_conv_multi`` call + the BatchNorm
    backward pair).  Returns (dx | None, [(gw, gb, dgamma, dbeta, gresidual) per branch]).  ``pre_dgrad[b]`` = (gv, bn_partial, rows,
    ld): branch b's data gradient (and BatchNorm-backward partial sums) already computed by another kernel (the fused AttentionBlock
    backward): no data-gradient job is issued for it.  ``fork`` (side-stream route only): "now" = fork the weight-gradient jobs where the
    data-gradient job is launched; "queue" = leave them queued for the next call's fork; "after_bn" = the fork point is here but the
    side stream's launches are recorded after the BatchNorm backward pair (with no data-gradient job in this call they would otherwise
    become the first successor of the launch stream's last kernel, and a captured graph keeps only the first successor on its queue)."""
    lib = _lib.load()
    mean, invstd, scales, shifts, training = stats[:5]
    nbr = len(specs)
    n, cs, hs, ws = x.shape
    m_in = n * hs * ws
    grads: List[Optional[Tensor]] = []
    gvs, partials, ps = [], [], []
    cspad = 0
    per_branch = []
    # Weight- and data-gradient of every branch are independent of each other: all of them go into ONE launch
    # (otvae_conv_multi).  When the weight gradient lands in a trainer-owned flat buffer (persistent memory, read by
    # the optimizer and not through autograd's accumulation) its partial -> gradient reduction is deferred so that
    # all layers of the backward pass reduce in one launch; otherwise it runs right after the multi launch:
    # autograd may clone / accumulate the returned tensor before a deferred kernel would have filled it.
    wjobs, djobs = (_lib.ConvJob * nbr)(), (_lib.ConvJob * nbr)()
    bwd_slots = BN_SLOTS_MODE >= 2 and cs <= 1024 and sum(1 for sp in specs if sp.has_norm) <= 2
    slot_jobs: list = []
    nw = nd = 0
    order = []       # (is_weight_job, index) in issue order of the single-launch variant
    all_deferred = True
    keep = []
    for b, sp in enumerate(specs):
        w, bias, gamma, beta, res = tensors[5 * b: 5 * b + 5]
        pw, pb, pgam, pbet = params_ref[b]
        gy = gys[b]
        if gy is None:
            gy = torch.zeros_like(empty_nhwc(n, w.shape[0], geoms[b].Ho, geoms[b].Wo, x))
        gy = as_nhwc(gy)
        g = geoms[b]
        # --- weight / bias gradient
        _, _, p_w, p_d, cp, dead_taps = _conv_plan(g, sp.has_bias)
        kk = g.KH * g.KW * g.Cs + (1 if sp.has_bias else 0)
        wpart = torch.empty((p_w, kk, g.Cn), device=x.device, dtype=torch.float32)
        gw = _grad_buffer(pw, w)
        gb = _grad_buffer(pb, bias) if sp.has_bias else None
        defer = (pw is not None and getattr(pw, "_otvae_grad_view", None) is not None and
                 (not sp.has_bias or getattr(pb, "_otvae_grad_view", None) is not None))
        all_deferred = all_deferred and defer
        jb = wjobs[nw]
        order.append((True, nw))
        nw += 1
        # deferred reductions know the taps that never touch the image (1x1 / 2x2 maps): their partial rows may stay unwritten
        jb.kind, jb.relu, jb.has_bias, jb.geom = _lib.JOB_BWD_WEIGHT, int(sp.relu), int(sp.has_bias), g
        jb.defer_reduce = (_lib.DEFER_DENSE if _DENSE_REDUCE else _lib.DEFER_SPARSE) if defer else 0
        jb.x, jb.gy = ptr(x), ptr(gy)
        jb.scale = ptr(scales[b]) if sp.has_norm else None
        jb.shift = ptr(shifts[b]) if sp.has_norm else None
        jb.wpartial, jb.gw, jb.gb = ptr(wpart), ptr(gw), ptr(gb)
        keep += [wpart, gy, scales[b] if sp.has_norm else None, shifts[b] if sp.has_norm else None]
        if defer:
            _PendingReduce.add(x.device, wpart, p_w, kk - (1 if sp.has_bias else 0), kk, g.Cn,
                               gw.data_ptr(), gb.data_ptr() if gb is not None else None, g.Cs, 0 if _DENSE_REDUCE else dead_taps)
        # --- data gradient (needed for dx and for the BatchNorm parameter gradients)
        gv = None
        part = None    # the BatchNorm-backward sums of this branch: fp64 partials (Tensor) or statistic slots (Slots)
        pre = pre_dgrad[b] if pre_dgrad is not None else None
        if pre is not None:
            gv, part, p_pre, cp_pre = pre
            if sp.has_norm:
                cspad = cp_pre
                ps.append(p_pre)
        elif need_dx or sp.has_norm:
            wd = getattr(pw, "_otvae_wd", None) if pw is not None else None
            if wd is None:
                wd = torch.empty(g.KH * g.KW * g.Cn * g.Cs, device=x.device, dtype=torch.float32)
                check(lib.otvae_weight_transpose(ptr(w), ptr(wd), g.KH * g.KW, g.Cs, g.Cn, stream()),
                      "otvae_weight_transpose")
            gv = empty_nhwc(n, cs, hs, ws, x)
            if sp.has_norm:
                # slots only when every normalised branch of this call gets them (one consumer launch reads them all)
                part = SlotArena.take(x.device, cp, p_d) if bwd_slots else None
                if part is None:
                    bwd_slots = False
                    part = torch.empty((p_d, 2, cp), device=x.device, dtype=torch.float64)
                    for b0, k0, p0, cp0 in slot_jobs:   # (the arena ran out between two branches: the earlier one goes back to partials)
                        part0 = torch.empty((p0, 2, cp0), device=x.device, dtype=torch.float64)
                        djobs[k0].bn_slots, djobs[k0].bn_nslots, djobs[k0].bn_partial = None, 0, ptr(part0)
                        per_branch[b0] = per_branch[b0][:3] + (part0,) + per_branch[b0][4:]
                    slot_jobs = []
                else:
                    slot_jobs.append((b, nd, p_d, cp))
                cspad = cp
                ps.append(p_d)
            jb = djobs[nd]
            order.append((False, nd))
            nd += 1
            jb.kind, jb.relu, jb.geom = _lib.JOB_BWD_DATA, int(sp.relu), g
            jb.gy, jb.w, jb.x = ptr(gy), ptr(wd), ptr(x)
            jb.scale = ptr(scales[b]) if sp.has_norm else None
            jb.shift = ptr(shifts[b]) if sp.has_norm else None
            jb.mean = ptr(mean) if sp.has_norm else None
            jb.invstd = ptr(invstd) if sp.has_norm else None
            jb.gv = ptr(gv)
            if isinstance(part, Slots):
                jb.bn_slots, jb.bn_nslots = ptr(part.buf), part.n
            else:
                jb.bn_partial = ptr(part)
            keep.append(wd)
        per_branch.append((gw, gb, gv, part, gy if sp.has_residual else None))
    if all_deferred and (WGRAD_SIDE_STREAM == 2 or (WGRAD_SIDE_STREAM == 1 and torch.cuda.is_current_stream_capturing())):
        # weight-gradient jobs on the side stream (see _PendingReduce), data-gradient jobs on the launch stream.  The data
        # job is issued FIRST: in a captured step the graph executor keeps the first child of a node on its parent's queue, so
        # the critical chain stays on one queue and only the side branch pays the cross-queue dependency (~12 us each)
        go = _PendingReduce.queue_weight_jobs(x.device, wjobs, nw, [x] + keep)
        ev = _PendingReduce.fork_point(x.device) if (go and fork != "queue") else None
        if nd:
            check(lib.otvae_conv_multi(nd, djobs, stream()), "otvae_conv_multi(backward, data)")
            if JOB_TRACE is not None:
                _trace_jobs(djobs, nd)
        if ev is not None and fork != "after_bn":
            _PendingReduce.issue(x.device, ev)
            ev = None
    else:
        ev = None
        njobs = nw + nd
        jobs = (_lib.ConvJob * njobs)()
        for i, (is_w, k) in enumerate(order):
            jobs[i] = wjobs[k] if is_w else djobs[k]
        check(lib.otvae_conv_multi(njobs, jobs, stream()), "otvae_conv_multi(backward)")
        if JOB_TRACE is not None:
            _trace_jobs(jobs, njobs)
    # --- BatchNorm backward over the branches that have one
    bn_idx = [b for b, sp in enumerate(specs) if sp.has_norm]
    dgam = {b: None for b in range(nbr)}
    dbet = {b: None for b in range(nbr)}
    dx = None
    kinds = {isinstance(per_branch[b][3], Slots) for b in bn_idx}
    if bn_idx and kinds == {True}:
        # sums in statistic slots: finalize + apply as one launch (otvae_bn_bwd_apply_slots), or the finalize half alone (dx not needed)
        nbn = len(bn_idx)
        for b in bn_idx:
            dgam[b] = _grad_buffer(params_ref[b][2], tensors[5 * b + 2])
            dbet[b] = _grad_buffer(params_ref[b][3], tensors[5 * b + 3])
        sl = [per_branch[b][3] for b in bn_idx]
        if any(s_.ld != sl[0].ld for s_ in sl):
            raise RuntimeError("BatchNorm-backward slots of one layer input with different channel strides")
        if need_dx:
            dx = empty_nhwc(n, cs, hs, ws, x)
        check(lib.otvae_bn_bwd_apply_slots(nbn, ptr_array([per_branch[b][2] for b in bn_idx]), ptr(x), ptr_array([s_.buf for s_ in sl]),
                                           (C.c_int * nbn)(*[s_.n for s_ in sl]), sl[0].ld, m_in, cs, ptr(mean), ptr(invstd),
                                           ptr_array([tensors[5 * b + 2] for b in bn_idx]), ptr_array([dgam[b] for b in bn_idx]),
                                           ptr_array([dbet[b] for b in bn_idx]), int(bool(training)), ptr(dx) if need_dx else None,
                                           stream()), "otvae_bn_bwd_apply_slots")
    elif bn_idx:
        if True in kinds:
            raise RuntimeError("BatchNorm-backward sums of one layer input split between slots and partials")
        nbn = len(bn_idx)
        coef = torch.empty((2 + nbn, cs), device=x.device, dtype=torch.float32)
        gam_t = [tensors[5 * b + 2] for b in bn_idx]
        for b in bn_idx:
            dgam[b] = _grad_buffer(params_ref[b][2], tensors[5 * b + 2])
            dbet[b] = _grad_buffer(params_ref[b][3], tensors[5 * b + 3])
        parr = (C.c_int * nbn)(*ps)
        check(lib.otvae_bn_bwd_finalize(nbn, ptr_array([per_branch[b][3] for b in bn_idx]), parr, cspad, m_in, cs,
                                        ptr(mean), ptr(invstd), ptr_array(gam_t),
                                        ptr_array([dgam[b] for b in bn_idx]), ptr_array([dbet[b] for b in bn_idx]),
                                        ptr(coef), stream()), "otvae_bn_bwd_finalize")
        if need_dx:
            if not training:
                coef[:2].zero_()  # eval mode: BatchNorm is a fixed affine, no batch-statistics terms
            dx = empty_nhwc(n, cs, hs, ws, x)
            check(lib.otvae_bn_bwd_apply(nbn, ptr_array([per_branch[b][2] for b in bn_idx]), ptr(x), ptr(coef),
                                         m_in, cs, ptr(dx), stream()), "otvae_bn_bwd_apply")
    if ev is not None:
        _PendingReduce.issue(x.device, ev)
    if need_dx:
        for b, sp in enumerate(specs):
            if not sp.has_norm:
                dx = per_branch[b][2] if dx is None else dx + per_branch[b][2]
    per = []
    for b_ in range(nbr):
        gw, gb, gv, part, gres = per_branch[b_]
        per.append((gw, gb, dgam[b_], dbet[b_], gres))
    return (dx if need_dx else None), per


# ---- activations other than ReLU, equalized_lr (reference networks/cnn.py:114-118,128-147,186-188): unfused around the kernels
ACT_KINDS = {None: 0, "relu": 1, "leaky": 2, "selu": 3, "gelu": 4, "silu": 5}


class _BnActFn(torch.autograd.Function):
    """a = act(BatchNorm(x)) (or act(x)) as one element-wise kernel, with the BatchNorm backward of the fused path (the same
    finalize / apply kernels) behind ``otvae_bn_act_bwd``.  ``stats`` = (mean, invstd, scale, shift, training) or None."""

    @staticmethod
    def forward(ctx, x, gamma, beta, stats, kind, params_ref):
        lib = _lib.load()
        n, c, h, w = x.shape
        a = empty_nhwc(n, c, h, w, x)
        scale, shift = (stats[2], stats[3]) if stats is not None else (None, None)
        check(lib.otvae_bn_act_fwd(ptr(x), ptr(scale), ptr(shift), kind, n * h * w, c, ptr(a), stream()), "otvae_bn_act_fwd")
        ctx.stats, ctx.kind, ctx.params_ref = stats, kind, params_ref
        ctx.save_for_backward(x, gamma)
        return a

    @staticmethod
    def backward(ctx, ga):
        lib = _lib.load()
        x, gamma = ctx.saved_tensors
        n, c, h, w = x.shape
        m = n * h * w
        ga = as_nhwc(ga)
        gv = empty_nhwc(n, c, h, w, x)
        stats = ctx.stats
        if stats is None:
            check(lib.otvae_bn_act_bwd(ptr(ga), ptr(x), None, None, None, None, ctx.kind, m, c, ptr(gv), None, stream()),
                  "otvae_bn_act_bwd")
            return gv, None, None, None, None, None
        mean, invstd, scale, shift, training = stats
        p = lib.otvae_bn_act_bwd_parts(m)
        part = torch.empty((2, c, p), device=x.device, dtype=torch.float64)
        check(lib.otvae_bn_act_bwd(ptr(ga), ptr(x), ptr(scale), ptr(shift), ptr(mean), ptr(invstd), ctx.kind, m, c, ptr(gv),
                                   ptr(part), stream()), "otvae_bn_act_bwd")
        coef = torch.empty((3, c), device=x.device, dtype=torch.float32)
        dgam = _grad_buffer(ctx.params_ref[0], gamma)
        dbet = _grad_buffer(ctx.params_ref[1], gamma)
        check(lib.otvae_bn_bwd_finalize(1, ptr_array([part]), (C.c_int * 1)(p), c, m, c, ptr(mean), ptr(invstd), ptr_array([gamma]),
                                        ptr_array([dgam]), ptr_array([dbet]), ptr(coef), stream()), "otvae_bn_bwd_finalize")
        dx = None
        if ctx.needs_input_grad[0]:
            if not training:
                coef[:2].zero_()  # eval mode: BatchNorm is a fixed affine
            dx = empty_nhwc(n, c, h, w, x)
            check(lib.otvae_bn_bwd_apply(1, ptr_array([gv]), ptr(x), ptr(coef), m, c, ptr(dx), stream()), "otvae_bn_bwd_apply")
        return dx, dgam, dbet, None, None, None


def _is_dense(t: Tensor) -> bool:
    """True if t's elements fill one contiguous block of memory in some dimension order (then an element-wise kernel may walk
    the memory and an output with the same strides has the same logical layout)"""
    order = sorted(range(t.dim()), key=lambda d: (-t.stride(d), d))
    expect = 1
    for d in reversed(order):
        if t.shape[d] != 1 and t.stride(d) != expect:
            return False
        expect *= t.shape[d]
    return True


class _GroupNormActFn(torch.autograd.Function):
    """a = act(GroupNorm(x)) / act(InstanceNorm(x)) (``gamma`` None) as one launch per pass (csrc/groupnorm.hip)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, kind, params_ref):
        lib = _lib.load()
        n, c, h, w = x.shape
        a = empty_nhwc(n, c, h, w, x)
        mean = torch.empty((n, groups), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        check(lib.otvae_group_norm_act_fwd(ptr(x), ptr(gamma), ptr(beta), n, h * w, c, groups, BN_EPS, kind, ptr(a), ptr(mean), ptr(rstd),
                                           stream()), "otvae_group_norm_act_fwd")
        ctx.cfg = (groups, kind, params_ref)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        return a

    @staticmethod
    def backward(ctx, ga):
        lib = _lib.load()
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        groups, kind, params_ref = ctx.cfg
        n, c, h, w = x.shape
        ga = as_nhwc(ga)
        dx = empty_nhwc(n, c, h, w, x)
        pg = pb = None
        if gamma is not None:
            pg = torch.empty((n, c), device=x.device, dtype=torch.float32)
            pb = torch.empty_like(pg)
        check(lib.otvae_group_norm_act_bwd(ptr(ga), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), n, h * w, c, groups, kind, ptr(dx),
                                           ptr(pg), ptr(pb), stream()), "otvae_group_norm_act_bwd")
        dgam = dbet = None
        if gamma is not None:
            dgam, dbet = _grad_buffer(params_ref[0], gamma), _grad_buffer(params_ref[1], beta)
            check(lib.otvae_colsum_f32(ptr(pg), n, c, ptr(dgam), stream()), "otvae_colsum_f32")
            check(lib.otvae_colsum_f32(ptr(pb), n, c, ptr(dbet), stream()), "otvae_colsum_f32")
        return dx, dgam, dbet, None, None, None


class _FilmFn(torch.autograd.Function):
    """x * scale[n, c] + bias[n, c] (FiLM conditioning of ConvLayer, reference cnn.py:160-181) on csrc/film_dropout2d.hip"""

    @staticmethod
    def forward(ctx, x, scale, bias):
        n, c, h, w = x.shape
        out = empty_nhwc(n, c, h, w, x)
        scale, bias = scale.contiguous(), bias.contiguous()
        check(_lib.load().otvae_film_fwd(ptr(x), ptr(scale), ptr(bias), n, h * w, c, ptr(out), stream()), "otvae_film_fwd")
        ctx.save_for_backward(x, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        x, scale = ctx.saved_tensors
        n, c, h, w = x.shape
        g = as_nhwc(g)
        gx = empty_nhwc(n, c, h, w, x)
        gs, gb = torch.empty_like(scale), torch.empty_like(scale)
        check(_lib.load().otvae_film_bwd(ptr(g), ptr(x), ptr(scale), n, h * w, c, ptr(gx), ptr(gs), ptr(gb), stream()), "otvae_film_bwd")
        return gx, gs, gb


class _Dropout2dFn(torch.autograd.Function):
    """nn.Dropout2d(p) in training mode: whole (sample, channel) maps dropped; mask recomputed from the call key in backward"""

    @staticmethod
    def forward(ctx, x, p, key, stream_id):
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h, w, x)
        used = torch.empty(1, device=x.device, dtype=torch.int64)
        check(_lib.load().otvae_dropout2d_fwd(ptr(x), n, h * w, c, float(p), ptr(key), int(stream_id), ptr(y), ptr(used), stream()),
              "otvae_dropout2d_fwd")
        ctx.save_for_backward(used)
        ctx.p = float(p)
        ctx.mark_non_differentiable(used)
        return y, used

    @staticmethod
    def backward(ctx, gy, _gused):
        (used,) = ctx.saved_tensors
        gy = as_nhwc(gy)
        n, c, h, w = gy.shape
        gx = empty_nhwc(n, c, h, w, gy)
        check(_lib.load().otvae_dropout2d_bwd(ptr(gy), n, h * w, c, ctx.p, ptr(used), ptr(gx), stream()), "otvae_dropout2d_bwd")
        return gx, None, None, None


def dropout2d(x: Tensor, p: float, key: Tensor, stream_id: int = 0, return_used: bool = False):
    """``nn.Dropout2d(p)`` (training mode) on a channels-last [N, C, H, W] tensor; ``key`` = ``new_dropout_key`` (the caller advances
    its counter); the keep mask [N, C] of a call is ``dropout2d_mask(used, N, C, p)``."""
    _lib.require_cuda(x, "dropout2d input")
    y, used = _Dropout2dFn.apply(as_nhwc(x), float(p), key, stream_id)
    return (y, used) if return_used else y


def dropout2d_mask(used: Tensor, n: int, c: int, p: float) -> Tensor:
    keep = torch.empty((n, c), device=used.device, dtype=torch.uint8)
    check(_lib.load().otvae_dropout2d_mask(n, c, float(p), ptr(used), ptr(keep), stream()), "otvae_dropout2d_mask")
    return keep.bool()


class _ScaleFn(torch.autograd.Function):
    """alpha * t on t's own memory order (the ``weight * conv_scale * lr_mult`` / ``bias * lr_mult`` of equalized_lr)"""

    @staticmethod
    def forward(ctx, t, alpha):
        ctx.alpha = alpha
        if not _is_dense(t):
            t = t.contiguous()
        out = torch.empty_strided(t.shape, t.stride(), device=t.device, dtype=t.dtype)
        check(_lib.load().otvae_scale_f32(ptr(t), alpha, t.numel(), ptr(out), stream()), "otvae_scale_f32")
        return out

    @staticmethod
    def backward(ctx, g):
        if not _is_dense(g):
            g = g.contiguous()
        out = torch.empty_strided(g.shape, g.stride(), device=g.device, dtype=g.dtype)
        check(_lib.load().otvae_scale_f32(ptr(g), ctx.alpha, g.numel(), ptr(out), stream()), "otvae_scale_f32")
        return out, None


class _WeightExpandFn(torch.autograd.Function):
    """The dense HWIO weight of a grouped and / or dilated ConvLayer from its nn.Conv2d-shaped parameter [Cout, Cin / groups, KH, KW]
    (csrc/weight_expand.hip): zeros between the groups and in the holes of the dilation."""

    @staticmethod
    def forward(ctx, w, cin, groups, dilation):
        cout, cig, kh, kw = w.shape
        if cig * groups != cin:
            raise ValueError(f"grouped weight {tuple(w.shape)} does not match {cin} input channels in {groups} groups")
        w = w.contiguous()
        dense = new_hwio(cout, cin, (kh - 1) * dilation + 1, (kw - 1) * dilation + 1, device=w.device)
        check(_lib.load().otvae_weight_expand_fwd(ptr(w), cout, cin, groups, kh, kw, dilation, ptr(dense), stream()),
              "otvae_weight_expand_fwd")
        ctx.geom = (cout, cin, groups, kh, kw, dilation)
        return dense

    @staticmethod
    def backward(ctx, g):
        cout, cin, groups, kh, kw, dilation = ctx.geom
        g = hwio_weight(g)
        gw = torch.empty((cout, cin // groups, kh, kw), device=g.device, dtype=g.dtype)
        check(_lib.load().otvae_weight_expand_bwd(ptr(g), cout, cin, groups, kh, kw, dilation, ptr(gw), stream()),
              "otvae_weight_expand_bwd")
        return gw, None, None, None


class _GenericConvFn(torch.autograd.Function):
    """y = conv(x, w) + bias for ANY stride and footprints up to 32 x 32 (csrc/conv_generic.hip): the fallback behind the tuned kernels
    (strides 1 / 2, footprints up to 7 x 7) for the layers `down_sample >= 4` / `CNN(scaling_factor >= 4)` make (cnn.py:98-101)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, params_ref):
        lib = _lib.load()
        n, cs, hs, ws_ = x.shape
        cn, _, kh, kw = w.shape
        ho, wo = (hs + 2 * pad - kh) // stride + 1, (ws_ + 2 * pad - kw) // stride + 1
        if ho <= 0 or wo <= 0:
            raise ValueError(f"a {kh} x {kw} kernel with stride {stride}, padding {pad} does not fit a {hs} x {ws_} map")
        g = ConvGeom(n, hs, ws_, cs, 1, ho, wo, cn, kh, kw, stride, pad)
        y = empty_nhwc(n, cn, ho, wo, x)
        check(lib.otvae_conv_generic_fwd(C.byref(g), ptr(x), ptr(w), ptr(bias), ptr(y), stream()), "otvae_conv_generic_fwd")
        ctx.save_for_backward(x, w, bias)
        ctx.g, ctx.pref = g, params_ref
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        lib = _lib.load()
        g = ctx.g
        gy = as_nhwc(gy)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_strided(x.shape, x.stride(), device=x.device, dtype=x.dtype)
            check(lib.otvae_conv_generic_bwd_data(C.byref(g), ptr(gy), ptr(w), ptr(gx), stream()), "otvae_conv_generic_bwd_data")
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            has_bias = bias is not None
            ws = torch.empty(lib.otvae_conv_generic_bwd_weight_ws(C.byref(g), int(has_bias)), device=x.device, dtype=torch.float32)
            gw = _grad_buffer(ctx.pref[0], w)
            gb = _grad_buffer(ctx.pref[1], bias) if has_bias else None
            check(lib.otvae_conv_generic_bwd_weight(C.byref(g), ptr(x), ptr(gy), int(has_bias), ptr(ws), ptr(gw), ptr(gb), stream()),
                  "otvae_conv_generic_bwd_weight")
        return gx, gw, gb, None, None, None


def _conv_layer_general(x: Tensor, br: dict, training: bool) -> Tensor:
    """One ConvLayer whose activation is not ReLU and / or whose weight and bias carry the equalized_lr multipliers: BatchNorm
    statistics (the fused path's kernels) -> ``_BnActFn`` -> the fused convolution kernels with neither BatchNorm nor activation."""
    kind = int(br.get("act", 1 if br["relu"] else 0))
    has_norm = br.get("gamma") is not None
    stats = None
    if has_norm:
        bn = BNBranch(br["gamma"], br["beta"], br.get("running_mean"), br.get("running_var"), br.get("num_batches_tracked"))
        if training:
            mean, invstd, sc, sh = bn_batch_stats(x, [bn])
        else:
            mean, invstd, s0, h0 = bn_eval_affine(bn)
            sc, sh = [s0], [h0]
        stats = (mean, invstd, sc[0], sh[0], training)
    a = x
    gn = br.get("group_norm")   # (groups, weight | None, bias | None): GroupNorm / InstanceNorm2d instead of BatchNorm
    film = br.get("film")       # (scale [N, C], bias [N, C]): FiLM conditioning sits between the normalisation and the activation
    norm_kind = 0 if film is not None else kind
    if gn is not None:
        a = _GroupNormActFn.apply(x, gn[1], gn[2], int(gn[0]), norm_kind, (gn[1], gn[2]))
    elif has_norm or norm_kind != 0:
        a = _BnActFn.apply(x, br.get("gamma"), br.get("beta"), stats, norm_kind, (br.get("gamma"), br.get("beta")))
    if film is not None:
        a = _FilmFn.apply(a, film[0], film[1])
        if kind != 0:
            a = _BnActFn.apply(a, None, None, None, kind, (None, None))
    up_module, down_module = br.get("up_module"), br.get("down_module")
    if up_module is not None:   # a user-supplied module between the activation and the convolution (cnn.py:187)
        a = as_nhwc(up_module(a))
    w = br["weight"]
    expand = br.get("expand")   # (groups, dilation): the parameter has nn.Conv2d's grouped shape, the kernels take the dense weight
    if expand is not None:
        w = _WeightExpandFn.apply(w, x.shape[1], int(expand[0]), int(expand[1]))
    w = w if is_hwio(w) else hwio_weight(w)
    bias = br.get("bias")
    ws, bs = float(br.get("wscale", 1.0)), float(br.get("bscale", 1.0))
    if ws != 1.0:
        w = _ScaleFn.apply(w, ws)
    if bias is not None and bs != 1.0:
        bias = _ScaleFn.apply(bias, bs)
    drop = br.get("dropout2d")  # (p, key): nn.Dropout2d behind the convolution, training mode only
    if down_module is not None and br.get("residual") is not None:
        raise ValueError("a residual cannot be fused in front of a down-sampling module")
    if br.get("generic"):   # a stride / footprint the tuned kernels do not take: the direct-convolution fallback
        direct = w is br["weight"] and (bias is None or bias is br.get("bias"))   # gradients may go straight into the parameters' slots
        y = _GenericConvFn.apply(as_nhwc(a), w, bias, int(br["stride"]), int(br["pad"]), (br["weight"], br.get("bias")) if direct else (None, None))
        if br.get("residual") is not None:
            y = y + as_nhwc(br["residual"])
    else:
        plain = dict(weight=w, bias=bias, residual=br.get("residual"), stride=br["stride"], pad=br["pad"], up=br["up"], relu=False,
                     out_stats=br.get("out_stats", False) and drop is None and down_module is None)
        y = conv_layers(a, [plain], training=training)[0]
    if down_module is not None:  # behind the convolution, in front of the dropout (cnn.py:190-191)
        y = as_nhwc(down_module(y))
    if drop is not None and training:
        y = dropout2d(y, drop[0], drop[1], stream_id=int(drop[2]))
    return y


def conv_layers(x: Tensor, branches: Sequence[dict], training: bool = True):
    """Runs 1 or 2 ConvLayer branches on the same input.  Each branch dict has:
    weight (OIHW/HWIO), bias|None, gamma|None, beta|None, running_mean/var/num_batches_tracked|None,
    residual|None, stride, pad, up, relu (+ optionally act: an ACT_KINDS value, wscale, bscale: see ``_conv_layer_general``).
    Returns one tensor per branch (logical NCHW, NHWC memory)."""
    _lib.require_cuda(x, "conv input")
    x = as_nhwc(x)
    if x.dtype != torch.float32:
        raise TypeError("the MI355X conv path computes in fp32")
    if any(br.get("act", 0) > 1 or br.get("wscale", 1.0) != 1.0 or br.get("bscale", 1.0) != 1.0 or br.get("group_norm") is not None
           or br.get("film") is not None or br.get("dropout2d") is not None or br.get("expand") is not None
           or br.get("up_module") is not None or br.get("down_module") is not None or br.get("generic") for br in branches):
        return tuple(_conv_layer_general(x, br, training) for br in branches)
    specs, tensors, params_ref, bns = [], [], [], []
    for br in branches:
        w = br["weight"]
        has_norm = br.get("gamma") is not None
        sp = ConvSpec(br["stride"], br["pad"], br["up"], bool(br["relu"]), has_norm, br.get("bias") is not None,
                      br.get("residual") is not None, bool(br.get("out_stats", False)) and training)
        specs.append(sp)
        wt = w if is_hwio(w) else hwio_weight(w)
        res = br.get("residual")
        if res is not None:
            res = as_nhwc(res)
        tensors += [wt, br.get("bias"), br.get("gamma"), br.get("beta"), res]
        params_ref.append((w if wt is w else None, br.get("bias"), br.get("gamma"), br.get("beta")))
        if has_norm:
            bns.append(BNBranch(br["gamma"], br["beta"], br.get("running_mean"), br.get("running_var"),
                                br.get("num_batches_tracked")))
    mean = invstd = None
    scales: List[Optional[Tensor]] = [None] * len(specs)
    shifts: List[Optional[Tensor]] = [None] * len(specs)
    if bns:
        fold = None
        if training:
            mean, invstd, sc, sh, fold = bn_batch_stats(x, bns, allow_fold=True)
        else:
            if len(bns) > 1:
                # running statistics of the branches may differ (loaded checkpoints): no shared (mean, invstd)
                return tuple(conv_layers(x, [br], training=False)[0] for br in branches)
            mean, invstd, s0, h0 = bn_eval_affine(bns[0])
            sc, sh = [s0], [h0]
        it = iter(range(len(bns)))
        for i, sp in enumerate(specs):
            if sp.has_norm:
                j = next(it)
                scales[i], shifts[i] = sc[j], sh[j]
    stats = (mean, invstd, scales, shifts, training, fold if bns else None)
    stats_out: List[Optional[tuple]] = []
    out = _ConvBNFn.apply(x, tuple(specs), stats, tuple(params_ref), stats_out, *tensors)
    out = out if isinstance(out, tuple) else (out,)
    for y, st in zip(out, stats_out):
        if st is not None:
            y._otvae_stats = st  # consumed by bn_batch_stats of the next layer (same tensor object, unmodified)
    return out


# ------------------------------------------------------------------------------------------------ attention
class _WidthError(AssertionError, ValueError):
    """a channel width the heads do not divide"""


def _attention_op(qkv4: Tensor, heads: int, scale: Optional[float]) -> Tensor:
    """``torch.ops.otvae.qkv_attention`` (ops.py) on a [N, 3*H*C, H, W] NHWC tensor: fused attention forward, its backward
    registered with ``torch.library.register_autograd``.  Head widths 1 and 2 (the 32x32 and 16x16 blocks, 83 % of the
    attention time): the forward pass also emits the per-query key moments from which the backward pass forms dq without
    another pass over the keys."""
    n, width, h, w = qkv4.shape
    if width % (3 * heads) != 0:
        # the reference checks this with `assert` (networks/nets_utils.py:71): an AssertionError to whoever catches that, and still a ValueError
        raise _WidthError(f"tensor width: {width} must be divisible by (3 * n_heads): {3 * heads}")
    c = width // (3 * heads)
    scale = 1.0 / c if scale is None else float(scale)  # 1/C = the two C^-1/2 factors of QKVAttention
    need_aux = qkv4.requires_grad and torch.is_grad_enabled()
    return torch.ops.otvae.qkv_attention(qkv4, heads, scale, need_aux)[0]


def qkv_attention(qkv: Tensor, n_heads: int) -> Tensor:
    """qkv: logical [N, 3*H*C, H, W] (NHWC memory) or [N, 3*H*C, T] -> same rank, H*C channels
    (reference networks/nets_utils.py:63-82)."""
    _lib.require_cuda(qkv, "qkv")
    if qkv.dim() == 3:
        n, width, t = qkv.shape
        q4 = as_nhwc(qkv.unsqueeze(-1))
        return _attention_op(q4, n_heads, None).squeeze(-1)
    return _attention_op(as_nhwc(qkv), n_heads, None)


# OTVAE_ATTN_STAGE=0 (A/B switch): the AttentionBlock runs as its three launches (qkv convolution, attention, output projection)
ATTN_STAGE = os.environ.get("OTVAE_ATTN_STAGE", "1") != "0"
ATTN_STAGE_BWD = os.environ.get("OTVAE_ATTN_STAGE_BWD", "1") != "0"  # ... and its backward pass likewise (one launch instead of three)
# OTVAE_ATTN_STAGE_QKV=1 (A/B switch): the forward kernel writes qkv and the backward kernel reads it instead of forming q / k / v again
ATTN_STAGE_KEEP_QKV = os.environ.get("OTVAE_ATTN_STAGE_QKV", "0") == "1"
_STAGE_PLAN_CACHE: dict = {}
_STAGE_BWD_PLAN_CACHE: dict = {}


def _plain_1x1(br: dict, cout: int, cin: int) -> bool:
    w = br["weight"]
    return (br.get("act", 0) <= 1 and not br["relu"] and br.get("wscale", 1.0) == 1.0 and br.get("bscale", 1.0) == 1.0
            and br.get("group_norm") is None and br.get("film") is None and br.get("dropout2d") is None and br.get("expand") is None
            and br.get("up_module") is None and br.get("down_module") is None and br.get("bias") is None
            and br["stride"] == 1 and br["pad"] == 0 and br["up"] == 1 and tuple(w.shape) == (cout, cin, 1, 1) and is_hwio(w))


def _stage_bwd_rows(n: int, t: int, heads: int, c: int) -> int:
    """partial-sum rows of the one-launch AttentionBlock backward (``otvae_attn_stage_bwd``), 0 when it does not take the shape"""
    key = (n, t, heads, c)
    rows = _STAGE_BWD_PLAN_CACHE.get(key)
    if rows is None:
        r = C.c_int(0)
        rows = _STAGE_BWD_PLAN_CACHE[key] = r.value if _lib.load().otvae_attn_stage_bwd_plan(n, t, heads, c, C.byref(r)) == 0 else 0
    return rows


class _AttnStageFn(torch.autograd.Function):
    """y = proj_out(attention(qkv(BN(x)))) [+ residual] in one launch (``otvae_attn_stage_fwd``); the backward pass is the three
    stages' own (output projection, attention, qkv convolution + BatchNorm), fed from what the fused kernel wrote."""

    @staticmethod
    def forward(ctx, x, meta, wq, gamma, beta, wp, res):
        lib = _lib.load()
        heads, scale, stats, pref_q, pref_p, rows, need_aux, stats_out = meta
        mean, invstd, scales, shifts, training = stats[:5]
        fold = stats[5] if len(stats) > 5 else None
        if fold is not None and fold.done:
            fold = None
        n, hc, hh, ww = x.shape
        t, c = hh * ww, hc // heads
        # the fused backward kernel forms q / k / v again from x: qkv is written only for a three-launch backward pass
        need_qkv = need_aux and (ATTN_STAGE_KEEP_QKV or not (ATTN_STAGE_BWD and _stage_bwd_rows(n, t, heads, c)))
        qkv = empty_nhwc(n, 3 * hc, hh, ww, x) if need_qkv else None
        out = empty_nhwc(n, hc, hh, ww, x)
        y = empty_nhwc(n, hc, hh, ww, x)
        lse = torch.empty((n, heads, t), device=x.device, dtype=torch.float32)
        aux = torch.empty((n, heads, t, c * c), device=x.device, dtype=torch.float32) if (need_aux and c <= 2) else None
        part = slots_out = None
        if stats_out is not None:
            slots_out = SlotArena.take(x.device, hc, rows)
            if slots_out is None:
                part = torch.empty((rows, 2, hc), device=x.device, dtype=torch.float64)
        if fold is not None or slots_out is not None:
            # the BatchNorm finalize of x folded into the launch's prologue and / or the output's sums into statistic slots
            fs = fold.fold_struct(0, True) if fold is not None else None
            check(lib.otvae_attn_stage_fwd_fold(ptr(x), C.byref(fs) if fs is not None else None,
                                                None if fold is not None else ptr(scales[0]), None if fold is not None else ptr(shifts[0]),
                                                ptr(wq), ptr(wp), ptr(res), n, t, heads, c, scale, ptr(qkv), ptr(out), ptr(lse), ptr(aux),
                                                ptr(y), ptr(part), ptr(slots_out.buf) if slots_out is not None else None,
                                                slots_out.n if slots_out is not None else 0, stream()), "otvae_attn_stage_fwd_fold")
            if fold is not None:
                fold.done = True
        else:
            check(lib.otvae_attn_stage_fwd(ptr(x), ptr(scales[0]), ptr(shifts[0]), ptr(wq), ptr(wp), ptr(res), n, t, heads, c, scale,
                                           ptr(qkv), ptr(out), ptr(lse), ptr(aux), ptr(y), ptr(part), stream()), "otvae_attn_stage_fwd")
        if stats_out is not None:
            stats_out.append(slots_out if slots_out is not None else (part, rows, hc))
        ctx.cfg = (heads, scale, stats, pref_q, pref_p, gamma is not None, res is not None)
        ctx.geoms = (_geom(x, wq, 1, 0, 1)[0], _geom(out, wp, 1, 0, 1)[0])
        ctx.save_for_backward(x, wq, gamma, beta, wp, res, qkv, out, lse, aux)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, wq, gamma, beta, wp, res, qkv, out, lse, aux = ctx.saved_tensors
        heads, scale, stats, pref_q, pref_p, has_norm, has_res = ctx.cfg
        training = stats[4]
        n, hc, hh, ww = x.shape
        t, c = hh * ww, hc // heads
        sp_p = ConvSpec(1, 0, 1, False, False, False, has_res, False)
        sp_q = ConvSpec(1, 0, 1, False, has_norm, False, False, False)
        no_stats = (None, None, [None], [None], training)
        gqkv = empty_nhwc(n, 3 * hc, hh, ww, x)
        rows = _stage_bwd_rows(n, t, heads, c)
        if qkv is None or (rows and ATTN_STAGE_BWD):
            # one launch for projection data gradient + attention backward + qkv data gradient with the BatchNorm sums
            gy = as_nhwc(gy)
            # both weight-gradient jobs share one side-stream launch, forked behind the fused kernel and recorded behind the launch
            # stream's next nodes (see fork=).  (A fork point of its own for the projection's job, in front of the fused kernel so that it
            # can run beside it, measured 0.01-0.02 ms SLOWER per step: profiles/r03_attn_stage_ab.txt.)
            _, per_p = conv_backward_launch(out, (wp, None, None, None, res), (sp_p,), (ctx.geoms[1],), no_stats, (pref_p,), (gy,), False,
                                            fork="queue")
            gv = empty_nhwc(n, hc, hh, ww, x)
            part = None
            if has_norm:
                part = SlotArena.take(x.device, hc, rows) if BN_SLOTS_MODE >= 2 else None   # statistic slots, else fp64 partials
                if part is None:
                    part = torch.empty((rows, 2, hc), device=x.device, dtype=torch.float64)
            norm_args = (ptr(stats[0]), ptr(stats[1]), ptr(stats[2][0]), ptr(stats[3][0])) if has_norm else (None, None, None, None)
            if isinstance(part, Slots):
                check(lib.otvae_attn_stage_bwd_slots(ptr(gy), ptr(wp), ptr(wq), ptr(x), *norm_args, ptr(qkv), ptr(out), ptr(lse), ptr(aux),
                                                     n, t, heads, c, scale, ptr(gqkv), ptr(gv), ptr(part.buf), part.n, stream()),
                      "otvae_attn_stage_bwd_slots")
            else:
                check(lib.otvae_attn_stage_bwd(ptr(gy), ptr(wp), ptr(wq), ptr(x), *norm_args, ptr(qkv), ptr(out), ptr(lse), ptr(aux),
                                               n, t, heads, c, scale, ptr(gqkv), ptr(gv), ptr(part), stream()), "otvae_attn_stage_bwd")
            dx, per_q = conv_backward_launch(x, (wq, None, gamma, beta, None), (sp_q,), (ctx.geoms[0],), stats, (pref_q,), (gqkv,),
                                             ctx.needs_input_grad[0], pre_dgrad=[(gv, part, rows, hc)], fork="after_bn")
            return dx, None, per_q[0][0], per_q[0][2], per_q[0][3], per_p[0][0], per_p[0][4]
        gout, per_p = conv_backward_launch(out, (wp, None, None, None, res), (sp_p,), (ctx.geoms[1],), no_stats, (pref_p,), (gy,), True)
        check(lib.otvae_attn_bwd_scaled(ptr(qkv), ptr(out), ptr(lse), ptr(gout), ptr(aux), n, t, heads, c, scale, ptr(gqkv), stream()),
              "otvae_attn_bwd")
        dx, per_q = conv_backward_launch(x, (wq, None, gamma, beta, None), (sp_q,), (ctx.geoms[0],), stats, (pref_q,), (gqkv,),
                                         ctx.needs_input_grad[0])
        return dx, None, per_q[0][0], per_q[0][2], per_q[0][3], per_p[0][0], per_p[0][4]


def attention_stage(x: Tensor, qkv_branch: dict, n_heads: int, proj_branch: dict, training: bool = True) -> Optional[Tensor]:
    """The reference's AttentionBlock (networks/cnn.py:212-240) on a channels-last x as ONE launch, or None when the fused kernel does
    not take this configuration (the caller then runs qkv convolution, attention and projection as three launches): both 1x1
    convolutions must be plain (no bias, activation, FiLM, equalized learning rate, group norm), the width a power of two <= 32."""
    if not ATTN_STAGE or x.dim() != 4 or x.dtype != torch.float32 or not x.is_cuda:
        return None
    n, hc, hh, ww = x.shape
    if n_heads <= 0 or hc % n_heads != 0:
        return None  # (the three-launch path raises the reference's error)
    if not (_plain_1x1(qkv_branch, 3 * hc, hc) and _plain_1x1(proj_branch, hc, hc)):
        return None
    if proj_branch.get("gamma") is not None or qkv_branch.get("residual") is not None:
        return None
    lib = _lib.load()
    t, c = hh * ww, hc // n_heads
    res = proj_branch.get("residual")
    wq, wp = qkv_branch["weight"], proj_branch["weight"]
    need_aux = torch.is_grad_enabled() and any(v is not None and v.requires_grad for v in (x, wq, wp, qkv_branch.get("gamma"), res))
    key = (n, t, n_heads, c, need_aux)
    rows = _STAGE_PLAN_CACHE.get(key)
    if rows is None:
        r = C.c_int(0)
        rows = _STAGE_PLAN_CACHE[key] = r.value if lib.otvae_attn_stage_plan(n, t, n_heads, c, int(need_aux), C.byref(r)) == 0 else 0
    if rows == 0:
        return None
    x = as_nhwc(x)
    if res is not None:
        res = as_nhwc(res)
    has_norm = qkv_branch.get("gamma") is not None
    mean = invstd = fold = None
    scales, shifts = [None], [None]
    if has_norm:
        bn = BNBranch(qkv_branch["gamma"], qkv_branch["beta"], qkv_branch.get("running_mean"), qkv_branch.get("running_var"),
                      qkv_branch.get("num_batches_tracked"))
        if training:
            mean, invstd, scales, shifts, fold = bn_batch_stats(x, [bn], allow_fold=True)
        else:
            mean, invstd, s0, h0 = bn_eval_affine(bn)
            scales, shifts = [s0], [h0]
    stats = (mean, invstd, scales, shifts, training, fold)
    stats_out: Optional[list] = [] if (proj_branch.get("out_stats", False) and training) else None
    meta = (n_heads, 1.0 / c, stats, (wq, None, qkv_branch.get("gamma"), qkv_branch.get("beta")), (wp, None, None, None), rows,
            need_aux, stats_out)
    y = _AttnStageFn.apply(x, meta, wq, qkv_branch.get("gamma"), qkv_branch.get("beta"), wp, res)
    if stats_out:
        y._otvae_stats = stats_out[0]
    return y


# ------------------------------------------------------------------------------------------------ prior / loss
# ------------------------------------------------------------------------------------------------ token streams (ViT)
def tokens_as_nhwc(x: Tensor) -> Tensor:
    """[N, T, D] contiguous tokens seen as the logical [N, D, T, 1] image the convolution kernels take (same memory)."""
    return x.contiguous().permute(0, 2, 1).unsqueeze(-1)


def nhwc_as_tokens(y: Tensor) -> Tensor:
    """inverse of ``tokens_as_nhwc`` for a kernel output: logical [N, D, T, 1] on NHWC memory -> [N, T, D] (a view)"""
    return as_nhwc(y).squeeze(-1).permute(0, 2, 1)


def linear_tokens(x: Tensor, weight: Tensor, bias: Optional[Tensor], relu_input: bool = False) -> Tensor:
    """``F.linear(relu(x) if relu_input else x, weight, bias)`` on [N, T, D_in] tokens through the 1x1 convolution kernels
    (the reference's ViT: patch embedding, the MultiheadAttention projections and the feed-forward pair of every
    nn.TransformerEncoderLayer, networks/vit.py:157-172).  ``weight`` is the logical [D_out, D_in] matrix; kept on
    [D_in][D_out] memory (``new_linear_weight``) it is consumed without a copy."""
    w4 = weight.unsqueeze(-1).unsqueeze(-1)
    # what a trainer attached to the parameter (its slot in the flat gradient buffer, the resident transposed copy the
    # data-gradient kernel reads) travels with the 1x1 view, so the weight gradient is written in place and reduced with
    # the step's other layers
    slot = getattr(weight, "_otvae_grad_view", None)
    if slot is not None:
        w4._otvae_grad_view = lambda: slot().unsqueeze(-1).unsqueeze(-1)
    wd = getattr(weight, "_otvae_wd", None)
    if wd is not None:
        w4._otvae_wd = wd
    y = conv_layers(tokens_as_nhwc(x), [dict(weight=w4, bias=bias, stride=1, pad=0, up=1, relu=relu_input)],
                    training=torch.is_grad_enabled())[0]
    return nhwc_as_tokens(y)


def new_linear_weight(d_out: int, d_in: int, device=None) -> Tensor:
    """logical [d_out, d_in] on [d_in][d_out] memory = the HWIO layout of a 1x1 convolution"""
    return torch.empty((d_in, d_out), device=device, dtype=torch.float32).t()


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps):
        lib = _lib.load()
        d = x.shape[-1]
        m = x.numel() // d
        x2 = x.reshape(m, d).contiguous()
        r2 = res.reshape(m, d).contiguous() if res is not None else None
        y = torch.empty_like(x2)
        s = torch.empty_like(x2) if r2 is not None else None
        mean = torch.empty(m, device=x.device, dtype=torch.float32)
        rstd = torch.empty(m, device=x.device, dtype=torch.float32)
        check(lib.otvae_layernorm_fwd(ptr(x2), ptr(r2), ptr(gamma), ptr(beta), m, d, float(eps), ptr(s), ptr(y), ptr(mean),
                                      ptr(rstd), stream()), "otvae_layernorm_fwd")
        ctx.save_for_backward(s if s is not None else x2, gamma, mean, rstd)
        ctx.has_res = res is not None
        ctx.pref = (gamma, beta)   # the parameters themselves: a trainer's flat-buffer slots hang on them (_grad_buffer)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xs, gamma, mean, rstd = ctx.saved_tensors
        m, d = xs.shape
        g2 = gy.reshape(m, d).contiguous()
        gx = torch.empty_like(xs)
        # written straight into the trainer's gradient slots when the parameters have them (one copy launch per parameter saved:
        # 28 per step of the reference's ViT); a parameter normalising twice in one step would need accumulation: not on this path
        dgamma, dbeta = _grad_buffer(ctx.pref[0], gamma), _grad_buffer(ctx.pref[1], gamma)
        ws = torch.empty(lib.otvae_layernorm_bwd_ws(m, d), device=xs.device, dtype=torch.float32)
        check(lib.otvae_layernorm_bwd(ptr(xs), ptr(g2), ptr(gamma), ptr(mean), ptr(rstd), m, d, ptr(gx), ptr(dgamma), ptr(dbeta),
                                      ptr(ws), stream()), "otvae_layernorm_bwd")
        gx = gx.reshape(gy.shape)
        return gx, (gx if ctx.has_res else None), dgamma, dbeta, None


class _LayerNormDropoutFn(torch.autograd.Function):
    """y = LayerNorm(res + dropout(x)) in one kernel; the mask is a hash the backward kernel recomputes"""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, eps, p, key, stream_id):
        lib = _lib.load()
        d = x.shape[-1]
        m = x.numel() // d
        x2, r2 = x.reshape(m, d).contiguous(), res.reshape(m, d).contiguous()
        y, s = torch.empty_like(x2), torch.empty_like(x2)
        mean = torch.empty(m, device=x.device, dtype=torch.float32)
        rstd = torch.empty(m, device=x.device, dtype=torch.float32)
        used = torch.empty(1, device=x.device, dtype=torch.int64)
        check(lib.otvae_layernorm_dropout_fwd(ptr(x2), ptr(r2), ptr(gamma), ptr(beta), m, d, float(eps), float(p), ptr(key),
                                              int(stream_id), ptr(s), ptr(y), ptr(mean), ptr(rstd), ptr(used), stream()),
              "otvae_layernorm_dropout_fwd")
        ctx.save_for_backward(s, gamma, mean, rstd, used)
        ctx.pref = (gamma, beta)
        ctx.p = float(p)
        ctx.mark_non_differentiable(used)
        return y.reshape(x.shape), used

    @staticmethod
    def backward(ctx, gy, _gused):
        lib = _lib.load()
        xs, gamma, mean, rstd, used = ctx.saved_tensors
        m, d = xs.shape
        g2 = gy.reshape(m, d).contiguous()
        gres, gx = torch.empty_like(xs), torch.empty_like(xs)
        dgamma, dbeta = _grad_buffer(ctx.pref[0], gamma), _grad_buffer(ctx.pref[1], gamma)
        ws = torch.empty(lib.otvae_layernorm_bwd_ws(m, d), device=xs.device, dtype=torch.float32)
        check(lib.otvae_layernorm_dropout_bwd(ptr(xs), ptr(g2), ptr(gamma), ptr(mean), ptr(rstd), m, d, ctx.p, ptr(used), ptr(gres),
                                              ptr(gx), ptr(dgamma), ptr(dbeta), ptr(ws), stream()), "otvae_layernorm_dropout_bwd")
        return gx.reshape(gy.shape), gres.reshape(gy.shape), dgamma, dbeta, None, None, None, None


def layer_norm_dropout_mask(used: Tensor, m: int, d: int, p: float) -> Tensor:
    """the keep mask [M, D] (bool) of the ``layer_norm_tokens(dropout_p > 0)`` call that returned ``used`` -- for tests"""
    lib = _lib.load()
    keep = torch.empty((m, d), device=used.device, dtype=torch.uint8)
    check(lib.otvae_layernorm_dropout_mask(m, d, float(p), ptr(used), ptr(keep), stream()), "otvae_layernorm_dropout_mask")
    return keep.bool()


def layer_norm_tokens(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5, residual: Optional[Tensor] = None,
                      dropout_p: float = 0.0, dropout_key: Optional[Tensor] = None, stream_id: int = 0, return_used: bool = False):
    """LayerNorm over the last dimension of ``x (+ residual)`` (torch.nn.functional.layer_norm arithmetic).  With
    ``dropout_p > 0`` the kernel computes LayerNorm(residual + dropout(x)) -- the post-norm transformer block in training
    mode -- with the mask key conventions of ``mha_attention_tokens``."""
    _lib.require_cuda(x, "layer_norm input")
    if x.dtype != torch.float32:
        raise TypeError("the MI355X LayerNorm computes in fp32")
    if residual is not None and residual.shape != x.shape:
        raise ValueError("`residual` must have the shape of `x`")
    if dropout_p > 0:
        if residual is None or dropout_key is None:
            raise ValueError("`dropout_p` > 0 needs the `residual` the thinned `x` is added to, and a `dropout_key`")
        y, used = _LayerNormDropoutFn.apply(x, residual, gamma, beta, eps, dropout_p, dropout_key, stream_id)
        return (y, used) if return_used else y
    return _LayerNormFn.apply(x, residual, gamma, beta, eps)


class _EmbeddingFn(torch.autograd.Function):
    """weight[idx] (nn.Embedding's lookup) whose backward is the library's deterministic row sum (``otvae_embedding_bwd``), written
    straight into the parameter's gradient slot when a trainer gave it one"""

    @staticmethod
    def forward(ctx, weight, idx):
        ctx.save_for_backward(idx)
        ctx.pref = weight
        ctx.shape = tuple(weight.shape)
        return weight.detach()[idx]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        k, d = ctx.shape
        g2 = g.reshape(-1, d).contiguous().float()
        gw = _grad_buffer(ctx.pref, ctx.pref)
        check(_lib.load().otvae_embedding_bwd(ptr(g2), ptr(idx.reshape(-1).contiguous()), g2.shape[0], k, d, ptr(gw), stream()),
              "otvae_embedding_bwd")
        return gw, None


def embedding(weight: Tensor, idx: Tensor) -> Tensor:
    """``F.embedding(idx, weight)`` for a 2-D fp32 weight on the device, with a native backward"""
    if not (weight.is_cuda and weight.dim() == 2 and weight.dtype == torch.float32 and weight.is_contiguous()):
        return torch.nn.functional.embedding(idx, weight)
    return _EmbeddingFn.apply(weight, idx)


class _ExpandBatchFn(torch.autograd.Function):
    """t [1 | none, *shape] -> [B, *shape] (materialised); backward: the sum over the batch by ``otvae_colsum_f32`` into the parameter's
    gradient slot when it has one (position embeddings, learned tokens of the ViT)"""

    @staticmethod
    def forward(ctx, t, b, pref):
        ctx.pref = pref if pref is not None else t
        ctx.tshape = tuple(t.shape)
        return t.reshape(1, -1).expand(b, -1).contiguous().reshape(b, *t.shape[(1 if t.dim() > 1 and t.shape[0] == 1 else 0):])

    @staticmethod
    def backward(ctx, g):
        b = g.shape[0]
        g2 = g.reshape(b, -1).contiguous().float()
        slot = _grad_buffer(ctx.pref, ctx.pref) if tuple(ctx.pref.shape) == ctx.tshape else torch.empty(ctx.tshape, device=g.device, dtype=torch.float32)
        check(_lib.load().otvae_colsum_f32(ptr(g2), b, g2.shape[1], ptr(slot), stream()), "otvae_colsum_f32")
        return slot.reshape(ctx.tshape), None, None


def expand_batch(t: Tensor, b: int, param: Optional[Tensor] = None) -> Tensor:
    """``t.expand(b, ...)`` made dense, for a tensor that is the same for every sample of the batch; ``param``: the parameter ``t`` IS
    (its gradient slot takes the batch sum directly); with a leading dimension of 1 that dimension becomes the batch"""
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        lead = t if (t.dim() > 1 and t.shape[0] == 1) else t.unsqueeze(0)
        return lead.expand(b, *lead.shape[1:]).contiguous()
    return _ExpandBatchFn.apply(t, b, param)


class _DropoutFn(torch.autograd.Function):
    """dropout(relu?(x)) over [..., D] tokens with a recomputed hash mask (no mask tensor)"""

    @staticmethod
    def forward(ctx, x, p, key, stream_id, relu):
        lib = _lib.load()
        d = x.shape[-1]
        x2 = x.reshape(-1, d).contiguous()
        y = torch.empty_like(x2)
        used = torch.empty(1, device=x.device, dtype=torch.int64)
        check(lib.otvae_dropout_fwd(ptr(x2), x2.shape[0], d, int(relu), float(p), ptr(key), int(stream_id), ptr(y), ptr(used),
                                    stream()), "otvae_dropout_fwd")
        ctx.save_for_backward(x2, used)
        ctx.cfg = (float(p), int(relu))
        ctx.mark_non_differentiable(used)
        return y.reshape(x.shape), used

    @staticmethod
    def backward(ctx, gy, _gused):
        lib = _lib.load()
        x2, used = ctx.saved_tensors
        p, relu = ctx.cfg
        g2 = gy.reshape(x2.shape).contiguous()
        gx = torch.empty_like(x2)
        check(lib.otvae_dropout_bwd(ptr(x2), ptr(g2), x2.shape[0], x2.shape[1], relu, p, ptr(used), ptr(gx), stream()),
              "otvae_dropout_bwd")
        return gx.reshape(gy.shape), None, None, None, None


def dropout_tokens(x: Tensor, p: float, dropout_key: Tensor, stream_id: int = 0, relu: bool = False, return_used: bool = False):
    """``dropout(relu(x))`` (``relu=True``) or ``dropout(x)`` over [..., D] tokens, D % 4 == 0, with the mask key conventions
    of ``mha_attention_tokens``; the mask equals ``layer_norm_dropout_mask(used, rows, D, p)``."""
    _lib.require_cuda(x, "dropout input")
    y, used = _DropoutFn.apply(x, p, dropout_key, stream_id, relu)
    return (y, used) if return_used else y


class _AttentionDropoutFn(torch.autograd.Function):
    """attention with dropout on the probabilities; the mask is a hash of (key, call site, slice, query, key token) that
    the backward kernel recomputes from the call key the forward kernel left in ``used``"""

    @staticmethod
    def forward(ctx, qkv, heads, scale, p, key, stream_id, causal):
        lib = _lib.load()
        n, width, h, w = qkv.shape
        t, c = h * w, width // (3 * heads)
        if key is None:  # nothing is dropped (causal attention only): any key will do
            key = torch.zeros(2, device=qkv.device, dtype=torch.int64)
        out = empty_nhwc(n, heads * c, h, w, qkv)
        lse = torch.empty((n, heads, t), device=qkv.device, dtype=torch.float32)
        used = torch.empty(1, device=qkv.device, dtype=torch.int64)
        check(lib.otvae_attn_dropout_fwd(ptr(qkv), n, t, heads, c, float(scale), float(p), int(causal), ptr(key), int(stream_id),
                                         ptr(out), ptr(lse), ptr(used), stream()), "otvae_attn_dropout_fwd")
        ctx.save_for_backward(qkv, out, lse, used)
        ctx.dims = (n, t, heads, c, float(scale), float(p), int(causal))
        ctx.mark_non_differentiable(used)
        return out, used

    @staticmethod
    def backward(ctx, gout, _gused):
        lib = _lib.load()
        qkv, out, lse, used = ctx.saved_tensors
        n, t, heads, c, scale, p, causal = ctx.dims
        gout = as_nhwc(gout)
        gqkv = torch.empty_strided(qkv.shape, qkv.stride(), device=qkv.device, dtype=qkv.dtype)
        check(lib.otvae_attn_dropout_bwd(ptr(qkv), ptr(out), ptr(lse), ptr(gout), n, t, heads, c, scale, p, causal, ptr(used),
                                         ptr(gqkv), stream()), "otvae_attn_dropout_bwd")
        return gqkv, None, None, None, None, None, None


class _CrossAttentionFn(torch.autograd.Function):
    """Tq queries against the keys / values of the Tk memory tokens, all three taken from ONE in-projected tensor
    ``qkv`` [N, Tq + Tk, 3 H C] (rows 0..Tq-1: the queries' projection, rows Tq..: the memory's; q | k | v head-major): q from
    the first third of the query rows, k / v from the other thirds of the memory rows.  The unused thirds get zero gradient."""

    @staticmethod
    def forward(ctx, qkv, tq, heads, scale, p, key, stream_id):
        lib = _lib.load()
        n, tall, w3 = qkv.shape
        tk, hc = tall - tq, w3 // 3
        c = hc // heads
        out = torch.empty((n, tq, hc), device=qkv.device, dtype=torch.float32)
        lse = torch.empty((n, heads, tq), device=qkv.device, dtype=torch.float32)
        used = torch.empty(1, device=qkv.device, dtype=torch.int64) if p > 0 else None
        base = qkv.data_ptr()
        kv0 = base + 4 * tq * w3  # first memory row
        check(lib.otvae_attn_cross_fwd(base, tall * w3, w3, kv0 + 4 * hc, kv0 + 8 * hc, tall * w3, w3, n, tq, tk, heads, c, float(scale),
                                       float(p), ptr(key) if p > 0 else None, int(stream_id), ptr(out), ptr(lse), ptr(used), stream()),
              "otvae_attn_cross_fwd")
        ctx.save_for_backward(qkv, out, lse, used)
        ctx.dims = (n, tq, tk, heads, c, float(scale), float(p))
        _CrossAttentionFn.last_used = used  # test aid: the call key of the latest forward (attention_cross_mask)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.load()
        qkv, out, lse, used = ctx.saved_tensors
        n, tq, tk, heads, c, scale, p = ctx.dims
        hc, w3, tall = heads * c, 3 * heads * c, tq + tk
        gout = gout.contiguous()
        gqkv = torch.zeros_like(qkv)  # the memory rows' q third and the query rows' k | v thirds took no part
        base, gbase = qkv.data_ptr(), gqkv.data_ptr()
        kv0, gkv0 = base + 4 * tq * w3, gbase + 4 * tq * w3
        check(lib.otvae_attn_cross_bwd(base, tall * w3, w3, kv0 + 4 * hc, kv0 + 8 * hc, tall * w3, w3, ptr(out), ptr(lse), ptr(gout),
                                       n, tq, tk, heads, c, scale, p, ptr(used), gbase, tall * w3, w3, gkv0 + 4 * hc, gkv0 + 8 * hc,
                                       tall * w3, w3, stream()), "otvae_attn_cross_bwd")
        return gqkv, None, None, None, None, None, None


def attention_cross_mask(used: Tensor, n: int, tq: int, tk: int, heads: int, p: float) -> Tensor:
    """the keep mask [N, H, Tq, Tk] (bool) of the cross-attention call whose forward left ``used`` -- for tests"""
    keep = torch.empty((n, heads, tq, tk), device=used.device, dtype=torch.uint8)
    check(_lib.load().otvae_attn_cross_mask(n, tq, tk, heads, float(p), ptr(used), ptr(keep), stream()), "otvae_attn_cross_mask")
    return keep.bool()


def cross_attention_tokens(x: Tensor, memory: Tensor, in_proj_weight: Tensor, in_proj_bias: Optional[Tensor], n_heads: int,
                           dropout_p: float = 0.0, dropout_key: Optional[Tensor] = None, stream_id: int = 0) -> Tensor:
    """The attention part of ``nn.MultiheadAttention(x, memory, memory)`` (before out_proj): softmax(q k^T / sqrt(C)) v with
    q = in_proj(x)[..., :D], k | v = in_proj(memory)[..., D:].  Both token sets go through ONE in-projection launch
    (concatenated along the token axis: a parameter enters the step once, so its gradient slot is written once), the
    attention kernel reads the thirds it needs through strides."""
    _lib.require_cuda(x, "x")
    if dropout_p > 0 and dropout_key is None:
        raise ValueError("`dropout_p` > 0 needs a `dropout_key` (functional.new_dropout_key)")
    tq = x.shape[1]
    qkv = linear_tokens(torch.cat((x, memory), dim=1), in_proj_weight, in_proj_bias).contiguous()
    c = qkv.shape[-1] // (3 * n_heads)
    return _CrossAttentionFn.apply(qkv, tq, n_heads, 1.0 / math.sqrt(c), float(dropout_p), dropout_key, stream_id)


def new_dropout_key(device, seed: Optional[int] = None) -> Tensor:
    """device int64[2] {seed, call counter} for ``mha_attention_tokens(dropout_p > 0)``; the seed comes from torch's default
    generator unless given, so ``torch.manual_seed`` governs the masks"""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return torch.tensor([seed, 0], dtype=torch.int64, device=device)


def attention_dropout_mask(used: Tensor, n: int, t: int, heads: int, p: float) -> Tensor:
    """the keep mask [N, H, T, T] (bool) of the call whose forward returned ``used`` -- for tests"""
    lib = _lib.load()
    keep = torch.empty((n, heads, t, t), device=used.device, dtype=torch.uint8)
    check(lib.otvae_attn_dropout_mask(n, t, heads, float(p), ptr(used), ptr(keep), stream()), "otvae_attn_dropout_mask")
    return keep.bool()


def mha_attention_tokens(qkv: Tensor, n_heads: int, dropout_p: float = 0.0, dropout_key: Optional[Tensor] = None,
                         stream_id: int = 0, return_used: bool = False, causal: bool = False):
    """softmax(q k^T / sqrt(C)) v per head on in-projected tokens [N, T, 3*H*C] (q | k | v, head-major: the layout of
    nn.MultiheadAttention's in_proj) -> [N, T, H*C]; the fused attention kernels with the 1/sqrt(C) score scale.
    ``dropout_p > 0`` (nn.MultiheadAttention's ``dropout`` in training mode) drops attention probabilities inside the
    kernel; ``dropout_key`` is the device int64[2] {seed, counter} of ``new_dropout_key`` (the caller advances the counter
    between steps), ``stream_id`` tells the layers sharing one key apart.  ``causal``: token t attends to tokens <= t."""
    _lib.require_cuda(qkv, "qkv")
    c = qkv.shape[-1] // (3 * n_heads)
    if dropout_p > 0 or causal:
        if dropout_p > 0 and dropout_key is None:
            raise ValueError("`dropout_p` > 0 needs a `dropout_key` (functional.new_dropout_key)")
        out, used = _AttentionDropoutFn.apply(tokens_as_nhwc(qkv), n_heads, 1.0 / math.sqrt(c), dropout_p, dropout_key, stream_id,
                                              causal)
        return (nhwc_as_tokens(out), used) if return_used else nhwc_as_tokens(out)
    out = _attention_op(as_nhwc(tokens_as_nhwc(qkv)), n_heads, 1.0 / math.sqrt(c))
    return nhwc_as_tokens(out)


def new_rng_key(device, seed: Optional[int] = None) -> Tensor:
    """device int64[3] {seed, call counter, block ticket} for ``normal_like``; the seed comes from torch's default generator
    unless given, so ``torch.manual_seed`` governs the draws"""
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    return torch.tensor([seed, 0, 0], dtype=torch.int64, device=device)


@torch.no_grad()
def normal_fill_(out: Tensor, key: Tensor, stream_id: int = 0, advance: bool = True) -> Tensor:
    """out ~ N(0, 1) in place from the device-side counter-based generator (``otvae_normal_fill``): capturable, and every
    replay of a captured call draws fresh values (the counter advances on the device)."""
    _lib.require_cuda(out, "out")
    if out.dtype != torch.float32 or not out.is_contiguous():
        raise TypeError("normal_fill_ fills contiguous float32 tensors")
    check(_lib.load().otvae_normal_fill(ptr(out), out.numel(), ptr(key), int(stream_id), int(advance), stream()),
          "otvae_normal_fill")
    return out


def normal_like(like: Tensor, key: Tensor, stream_id: int = 0) -> Tensor:
    return normal_fill_(torch.empty(like.shape, device=like.device, dtype=torch.float32), key, stream_id)


class _CondGaussianPriorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, eps, pm, pl, coeff):
        lib = _lib.load()
        b, n = eps.shape
        z = torch.empty_like(eps)
        loss = torch.empty(b, device=h.device, dtype=torch.float32)
        check(lib.otvae_gaussian_prior_cond_fwd(ptr(h), ptr(eps), ptr(pm), ptr(pl), b, n, float(coeff), ptr(z), ptr(loss),
                                                stream()), "otvae_gaussian_prior_cond_fwd")
        ctx.save_for_backward(h, eps, pm, pl)
        ctx.coeff = float(coeff)
        return z, loss

    @staticmethod
    def backward(ctx, gz, gloss):
        lib = _lib.load()
        h, eps, pm, pl = ctx.saved_tensors
        b, n = eps.shape
        gz = gz.contiguous() if gz is not None else None
        gloss = gloss.contiguous() if gloss is not None else None
        gh = torch.empty_like(h)
        gpm = torch.empty_like(pm) if ctx.needs_input_grad[2] else None
        gpl = torch.empty_like(pl) if ctx.needs_input_grad[3] else None
        check(lib.otvae_gaussian_prior_cond_bwd(ptr(h), ptr(eps), ptr(pm), ptr(pl), ptr(gz), ptr(gloss), b, n, ctx.coeff, ptr(gh),
                                                ptr(gpm), ptr(gpl), stream()), "otvae_gaussian_prior_cond_bwd")
        return gh, None, gpm, gpl, None


def gaussian_prior_conditional(h: Tensor, eps: Tensor, prior_mean: Tensor, prior_log_std: Tensor, coeff: float):
    """(z, coeff * KL(q || N(prior_mean, exp(prior_log_std)^2))[B]) for h [B, 2, ...] re-parametrised on dim 1, everything
    else flattened (reference prior/conditional_gaussian.py:84-93)."""
    _lib.require_cuda(h, "prior input")
    b = h.shape[0]
    out_shape = list(h.shape)
    out_shape[1] //= 2
    flat = lambda t: t.reshape(b, -1).contiguous().float()  # noqa: E731
    z, loss = _CondGaussianPriorFn.apply(flat(h), flat(eps), flat(prior_mean), flat(prior_log_std), coeff)
    return z.reshape(out_shape), loss


def gaussian_prior(h: Tensor, eps: Tensor, coeff: float) -> Tuple[Tensor, Tensor]:
    """(z, coeff*KL[B]) for the re-parametrised diagonal Gaussian (reference prior/gaussian.py:63-96)."""
    _lib.require_cuda(h, "prior input")
    if h.dim() == 2:
        z, loss = torch.ops.otvae.gaussian_prior(as_nhwc(h[:, :, None, None]), as_nhwc(eps[:, :, None, None]), float(coeff))
        return z[:, :, 0, 0], loss
    if h.dim() == 3:  # tokens [B, 2S, D] (the ViT's embed tokens): mu = the first S tokens, log_var = the last S
        b, s2, d = h.shape
        z, loss = gaussian_prior(h.reshape(b, s2 * d), eps.reshape(b, -1), coeff)
        return z.reshape(b, s2 // 2, d), loss
    if h.dim() != 4:
        raise ValueError("GaussianPrior on the MI355X path expects [B, 2D], [B, 2S, D] or [B, 2D, H, W] with reparam_dim=1")
    return torch.ops.otvae.gaussian_prior(as_nhwc(h), as_nhwc(eps), float(coeff))


class _GaussianPriorExFn(torch.autograd.Function):
    """``GaussianPrior`` with ``empirical_kl`` and / or ``fixed_var`` (and the temperature of ``encode(time=)``):
    ``otvae_gaussian_prior_ex_fwd / _bwd`` on h flattened to [B, S = 1, D (or 2D)]."""

    @staticmethod
    def forward(ctx, h, eps, temp, coeff, mode, s_):
        lib = _lib.load()
        b = h.shape[0]
        d = (h.shape[1] if mode & 2 else h.shape[1] // 2) // s_
        z = torch.empty((b, s_ * d), device=h.device, dtype=torch.float32)
        loss = torch.empty(b, device=h.device, dtype=torch.float32)
        check(lib.otvae_gaussian_prior_ex_fwd(ptr(h), ptr(eps), ptr(temp), b, s_, d, float(coeff), int(mode), ptr(z), ptr(loss), stream()),
              "otvae_gaussian_prior_ex_fwd")
        ctx.save_for_backward(h, eps, temp)
        ctx.cfg = (float(coeff), int(mode), d, int(s_))
        ctx.set_materialize_grads(False)
        return z, loss

    @staticmethod
    def backward(ctx, gz, gloss):
        if gz is None and gloss is None:
            return None, None, None, None, None, None
        h, eps, temp = ctx.saved_tensors
        coeff, mode, d, s_ = ctx.cfg
        gh = torch.empty_like(h)
        gz = gz.contiguous() if gz is not None else None
        gloss = gloss.contiguous() if gloss is not None else None
        check(_lib.load().otvae_gaussian_prior_ex_bwd(ptr(h), ptr(eps), ptr(temp), ptr(gz), ptr(gloss), h.shape[0], s_, d, coeff, mode,
                                                      ptr(gh), stream()), "otvae_gaussian_prior_ex_bwd")
        return gh, None, None, None, None, None


def _reparam_layout(h: Tensor, reparam_dim: int, fixed_var: bool):
    """(S, out_shape) of ``torch.chunk(h, 2, reparam_dim)`` on a contiguous h: within each of the S = prod(sizes between the batch and
    the chunked dimension) slices the first half holds the means and the second the log-variances, which is the [B][S][2 D] layout
    the kernels read in place (no gather of the two halves).  fixed_var: nothing is chunked."""
    r = reparam_dim if reparam_dim >= 0 else h.dim() + reparam_dim
    if fixed_var:
        return 1, list(h.shape)
    if not 1 <= r < h.dim():
        raise ValueError(f"`reparam_dim`={reparam_dim} must address a non-batch dimension of a {h.dim()}-d input")
    if h.shape[r] % 2:
        raise ValueError(f"dimension {r} of size {h.shape[r]} cannot be split into mean and log-variance halves")
    s_ = 1
    for k in range(1, r):
        s_ *= h.shape[k]
    out_shape = list(h.shape)
    out_shape[r] //= 2
    return s_, out_shape


def gaussian_prior_ex(h: Tensor, eps: Tensor, coeff: float, empirical_kl: bool = False, fixed_var: bool = False,
                      temperature: Optional[Tensor] = None, reparam_dim: int = 1) -> Tuple[Tensor, Tensor]:
    """(z, coeff * loss[B]) of ``GaussianPrior(empirical_kl=, fixed_var=, reparam_dim=)`` (reference prior/gaussian.py:63-96,
    prior/base.py:65-68) for h re-parametrised on ``reparam_dim`` (fixed_var: z = h + s eps with s = 1 or temperature[b] + 1e-8)."""
    _lib.require_cuda(h, "prior input")
    b = h.shape[0]
    s_, out_shape = _reparam_layout(h, reparam_dim, fixed_var)
    flat = h.contiguous().reshape(b, -1)
    temp = None
    if temperature is not None:
        if not fixed_var:
            raise ValueError("a temperature (`time`) is only meaningful with fixed_var=True")
        temp = temperature.reshape(b).float().contiguous()
    mode = (1 if empirical_kl else 0) | (2 if fixed_var else 0)
    z, loss = _GaussianPriorExFn.apply(flat.float(), eps.reshape(b, -1).float().contiguous(), temp, float(coeff), mode, s_)
    return z.reshape(out_shape), loss


class _CondGaussianPriorExFn(torch.autograd.Function):
    """``ConditionalGaussianPrior`` with empirical_kl / fixed_var / a re-parametrisation dimension other than 1:
    ``otvae_gaussian_prior_cond_ex_fwd / _bwd``"""

    @staticmethod
    def forward(ctx, h, eps, pm, pl, coeff, mode, s_):
        lib = _lib.load()
        b, n = eps.shape
        z = torch.empty_like(eps)
        loss = torch.empty(b, device=h.device, dtype=torch.float32)
        check(lib.otvae_gaussian_prior_cond_ex_fwd(ptr(h), ptr(eps), ptr(pm), ptr(pl), b, s_, n // s_, float(coeff), int(mode), ptr(z),
                                                   ptr(loss), stream()), "otvae_gaussian_prior_cond_ex_fwd")
        ctx.save_for_backward(h, eps, pm, pl)
        ctx.cfg = (float(coeff), int(mode), int(s_))
        ctx.set_materialize_grads(False)
        return z, loss

    @staticmethod
    def backward(ctx, gz, gloss):
        if gz is None and gloss is None:
            return (None,) * 7
        h, eps, pm, pl = ctx.saved_tensors
        coeff, mode, s_ = ctx.cfg
        b, n = eps.shape
        gz = gz.contiguous() if gz is not None else None
        gloss = gloss.contiguous() if gloss is not None else None
        gh = torch.empty_like(h)
        gpm = torch.empty_like(pm) if ctx.needs_input_grad[2] else None
        gpl = torch.empty_like(pl) if ctx.needs_input_grad[3] else None
        check(_lib.load().otvae_gaussian_prior_cond_ex_bwd(ptr(h), ptr(eps), ptr(pm), ptr(pl), ptr(gz), ptr(gloss), b, s_, n // s_, coeff,
                                                           mode, ptr(gh), ptr(gpm), ptr(gpl), stream()), "otvae_gaussian_prior_cond_ex_bwd")
        return gh, None, gpm, gpl, None, None, None


def gaussian_prior_conditional_ex(h: Tensor, eps: Tensor, prior_mean: Tensor, prior_log_std: Tensor, coeff: float,
                                  empirical_kl: bool = False, fixed_var: bool = False, reparam_dim: int = 1):
    """(z, coeff * loss[B]) of ``ConditionalGaussianPrior`` with the options it inherits (prior/conditional_gaussian.py:84-93): the prior
    rows [B, prod(dim)] are laid out like z."""
    _lib.require_cuda(h, "prior input")
    b = h.shape[0]
    s_, out_shape = _reparam_layout(h, reparam_dim, fixed_var)
    flat = lambda t: t.contiguous().reshape(b, -1).float()  # noqa: E731
    mode = (1 if empirical_kl else 0) | (2 if fixed_var else 0)
    z, loss = _CondGaussianPriorExFn.apply(flat(h), flat(eps), flat(prior_mean), flat(prior_log_std), float(coeff), mode, s_)
    return z.reshape(out_shape), loss


def nelbo_loss(pred: Tensor, target: Tensor, prior_loss: Optional[Tensor]) -> Tensor:
    """[total, recon, prior] of VAE.nelbo (reference model/vae.py:158-176): recon = mse(pred, target),
    prior = mean(prior_loss) / prod(target.shape[1:]).

    Contract inside a training engine's CAPTURED step (``engine.HipTrainer`` with the package's own ``VAE.nelbo``): the value is
    launched on the side stream with the backward pass's first fork and is complete on the launch stream only behind
    ``_PendingReduce.flush`` (before the step guard / optimizer read it) -- nothing may READ the returned vector on the launch
    stream before that join.  The engine switches the deferral off for a model class that overrides ``nelbo``."""
    _lib.require_cuda(pred, "pred")
    pred = as_nhwc(pred) if pred.dim() == 4 else pred.contiguous()
    target = as_nhwc(target) if target.dim() == 4 else target.contiguous()
    chw = 1
    for s in target.shape[1:]:
        chw *= s
    return torch.ops.otvae.nelbo_loss(pred, target, prior_loss.contiguous() if prior_loss is not None else None, float(chw))


def conv_bn_act(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, gamma: Optional[Tensor] = None,
                beta: Optional[Tensor] = None, running_mean: Optional[Tensor] = None, running_var: Optional[Tensor] = None,
                num_batches_tracked: Optional[Tensor] = None, residual: Optional[Tensor] = None, stride: int = 1, pad: int = 0,
                up: int = 1, relu: bool = True, training: bool = True) -> Tensor:
    """One ConvLayer (reference networks/cnn.py:183-192: BatchNorm -> ReLU -> nearest x``up`` -> conv (+ bias, + residual)) through
    the registered custom operators ``torch.ops.otvae.bn_batch_stats`` + ``torch.ops.otvae.conv_bn_act`` (ops.py)."""
    mean = invstd = scale = shift = None
    if gamma is not None:
        with torch.no_grad():
            mean, invstd, scale, shift = torch.ops.otvae.bn_batch_stats(x, gamma, beta, running_mean, running_var,
                                                                        num_batches_tracked, training)
    return torch.ops.otvae.conv_bn_act(x, weight, bias, gamma, beta, mean, invstd, scale, shift, residual, stride, pad, up, relu,
                                       training)
