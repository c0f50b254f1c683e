"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on MI355X,
``gloo`` in CPU tests).  The reference only ever uses stock Lightning DDP (configs/ddp.yaml:1-5) whose exchange step is
the bucketed gradient all-reduce; here every gradient already lives in ONE flat buffer, so the exchange is a single
all-reduce(SUM) (6.6 MiB for the MNIST config, 26 MiB for CIFAR: latency-bound on xGMI, so one collective beats
buckets) issued on a side stream, followed by Adam consuming ``grad / world_size``."""
from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .. import _lib

__all__ = ["FlatGradReducer", "world_size", "broadcast_module"]


def world_size(group=None) -> int:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group)
    return 1


class FlatGradReducer:
    """all-reduce(SUM) of a flat gradient buffer.  On GPU tensors the collective runs on its own stream so that it can
    overlap with whatever the main stream still has queued (the latent-statistics kernels at the end of backward)."""

    def __init__(self, flat: Tensor, group=None, enabled: bool = True):
        self.flat, self.group = flat, group
        # enabled=False: a rank-local engine inside a distributed job (measurement helpers of bench.py): no collective at all
        self.world = world_size(group) if enabled else 1
        # a 1-rank process group still runs the collective (used to rehearse the multi-GPU call sequence on one GPU)
        self.active = enabled and dist.is_available() and dist.is_initialized()
        self.stream = _lib.fresh_stream(flat.device) if (flat.is_cuda and self.active) else None

    @property
    def grad_scale(self) -> float:
        """what Adam multiplies the summed gradient with: DDP averages over ranks"""
        return 1.0 / self.world

    def allreduce_range(self, lo: int, hi: int, wait: bool = True) -> None:
        """all-reduce(SUM) of flat[lo:hi] on the reducer's stream, ordered after everything queued so far on the current
        stream; with ``wait=False`` the current stream is NOT made to wait (call ``join()`` before consuming)."""
        if hi <= lo or not self.active:
            return
        part = self.flat[lo:hi]
        if self.stream is None:
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group)
        if wait:
            self.join()

    def join(self) -> None:
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def allreduce(self) -> None:
        if self.world == 1:
            return
        if self.stream is None:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        torch.cuda.current_stream().wait_stream(self.stream)


@torch.no_grad()
def broadcast_module(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Identical replicas: every parameter and buffer takes rank ``src``'s value (what DDP does at construction)."""
    if world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_contiguous():
            dist.broadcast(t.data, src=src, group=group)
        else:  # strided views (HWIO conv weights): go through a dense temporary
            tmp = t.data.contiguous()
            dist.broadcast(tmp, src=src, group=group)
            t.data.copy_(tmp)
