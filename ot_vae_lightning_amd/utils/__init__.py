"""Host-side helpers mirroring the subset of the reference's ``ot_vae_lightning/utils/__init__.py`` that the hot path
touches (DDPMixin collectives hooks :37-46, FilterKwargs/hasarg :78-109,221-230, replicate/mean_replicated_batch
:154-175, ema :204-206, permute_and_flatten/unflatten_and_unpermute :233-311, unsqueeze_like :314-328).
Pure host logic on tensors of any device; no arithmetic hot loops live here."""
import contextlib
import inspect
import warnings
from copy import copy
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from torch import Tensor

__all__ = ["DDPMixin", "FilterKwargs", "hasarg", "replicate_batch", "mean_replicated_batch", "std_replicated_batch",
           "ema", "laplace_smoothing", "permute_and_flatten", "unflatten_and_unpermute", "unsqueeze_like", "ddp_reduce_sum",
           "ddp_gather_all", "apply_to_collection", "ema_inplace", "VectorLayout", "PartialCheckpoint", "human_format"]


def _dist_on() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def ddp_reduce_sum(t: Tensor) -> Tensor:
    """All-reduce(SUM) over the default process group (RCCL on GPUs); identity in a single process -- the
    behaviour of Lightning's ``sync_ddp_if_available(reduce_op='sum')`` the reference binds (utils/__init__.py:32)."""
    if _dist_on():
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def ddp_gather_all(t: Tensor) -> List[Tensor]:
    if _dist_on():
        out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(out, t.contiguous())
        return out
    return [t]


def _rank_zero_warn(msg: str) -> None:
    if not _dist_on() or dist.get_rank() == 0:
        warnings.warn(msg)


class DDPMixin(object):
    """Collectives injected as callables (reference utils/__init__.py:37-46)."""

    def __init__(self, ddp_reduce_func: Optional[Callable] = ddp_reduce_sum,
                 ddp_gather_func: Optional[Callable] = ddp_gather_all,
                 ddp_warn_func: Optional[Callable] = _rank_zero_warn):
        self.reduce = ddp_reduce_func or (lambda x: x)
        self.gather = ddp_gather_func or (lambda x: [x])
        self.warn = ddp_warn_func or warnings.warn


def hasarg(callee, arg_name: str) -> bool:
    func = getattr(callee, "forward") if isinstance(callee, nn.Module) else callee
    return arg_name in inspect.signature(func).parameters.keys()


class FilterKwargs(contextlib.AbstractContextManager):
    """``with FilterKwargs(f, 'z') as g: g(x, z=z)`` drops ``z`` when ``f`` does not accept it."""

    def __init__(self, callee: Callable, arg_keys: Union[str, List[str]] = ()):
        self.callee = callee
        self.arg_keys = arg_keys if isinstance(arg_keys, list) else [arg_keys]

    def __call__(self, *args, **kwargs):
        kept = copy(kwargs)
        for key in self.arg_keys:
            if key in kwargs and not hasarg(self.callee, key):
                kept.pop(key)
        return self.callee(*args, **kept)

    def __exit__(self, exc_type, exc_val, exc_tb):
        return False


def apply_to_collection(data, dtype, function, *args, **kwargs):
    if isinstance(data, dtype):
        return function(data, *args, **kwargs)
    if isinstance(data, dict):
        return {k: apply_to_collection(v, dtype, function, *args, **kwargs) for k, v in data.items()}
    if isinstance(data, (list, tuple)):
        return type(data)(apply_to_collection(v, dtype, function, *args, **kwargs) for v in data)
    return data


def _replicate_tensor(t: Tensor, n: int) -> Tensor:
    return t.unsqueeze(0).expand(n, *t.shape).reshape(n * t.shape[0], *t.shape[1:])


def replicate_batch(batch, n: int):
    if n in (0, 1) or batch is None:
        return batch
    return apply_to_collection(batch, Tensor, _replicate_tensor, n)


def _unsqueeze_first(batch: Tensor, n: int) -> Tensor:
    return batch.reshape(n, batch.shape[0] // n, *batch.shape[1:])


def mean_replicated_batch(expanded: Tensor, n: int) -> Tensor:
    if n in (0, 1):
        return expanded
    return _unsqueeze_first(expanded, n).mean(0)


def std_replicated_batch(expanded: Tensor, n: int) -> Tensor:
    if n in (0, 1):
        return expanded
    return _unsqueeze_first(expanded, n).std(0)


def ema(moving_avg, new, decay):
    if decay is None:
        return moving_avg + new
    return moving_avg * decay + new * (1 - decay)


def ema_inplace(moving_avg, new, decay):
    """in-place counterpart of ``ema`` (reference utils/__init__.py:190-201)"""
    if decay is None:
        moving_avg.data.add_(new)
    else:
        moving_avg.data.mul_(decay).add_(new, alpha=(1 - decay))


def laplace_smoothing(x, n_categories, eps=1e-5):
    """Additive smoothing of a count vector that keeps its total (reference utils/__init__.py:209-218)."""
    if eps is None:
        return x
    total = x.sum(-1, keepdim=True)
    return (x + eps) / (total + n_categories * eps) * total


def permute_and_flatten(x: Tensor, permute_dims: Sequence[int], batch_first: bool = True,
                        flatten_batch: bool = False) -> Tensor:
    """[B, ...] -> [B, prod(other dims), prod(permute dims)] (or [prod(other), B, .] / [B*prod(other), .])."""
    all_dims = set(range(1, x.dim()))
    if len(all_dims) == 0:
        raise ValueError("`input` is expected to have at least 2 dimensions")
    if len(permute_dims) == 0:
        raise ValueError("`permute_dims` is expected to contain at least one dimension")
    if not set(permute_dims).issubset(all_dims):
        raise ValueError("`permute_dims` is expected to be a subset of the `input` dimensions")
    rest = sorted(all_dims.difference(set(permute_dims)))
    if len(rest) == 0:
        return x.flatten(int(not flatten_batch))
    order = (0, *rest, *permute_dims) if batch_first else (*rest, 0, *permute_dims)
    y = x.permute(*order).contiguous()
    y = y.flatten(int(batch_first and not flatten_batch), len(rest) - int(not batch_first and not flatten_batch))
    return y.flatten(-len(permute_dims))


def unflatten_and_unpermute(xr: Tensor, orig_shape: Sequence[int], permute_dims: Sequence[int],
                            batch_first: bool = True, flatten_batch: bool = False) -> Tensor:
    rest = sorted(set(range(1, len(orig_shape))).difference(set(permute_dims)))
    if len(rest) == 0:
        return xr.view(*orig_shape)
    pshape = [orig_shape[d] for d in permute_dims]
    rshape = [orig_shape[d] for d in rest]
    x = xr
    if flatten_batch:
        bs, nrest = orig_shape[0], int(np.prod(rshape))
        x = x.unflatten(0, [bs, nrest] if batch_first else [nrest, bs])
    x = x.unflatten(-1, pshape)
    x = x.unflatten(int(batch_first), rshape)
    pmap = list(range(len(orig_shape)))
    if not batch_first:
        pmap[0] = len(rest)
    for dim in range(1, len(orig_shape)):
        if dim in rest:
            pmap[dim] = rest.index(dim) + int(batch_first)
        else:
            pmap[dim] = len(rest) + 1 + list(permute_dims).index(dim)
    return x.permute(*pmap).contiguous()


class VectorLayout:
    """How a latent tensor [B, *size] is cut into vectors: the axes in ``vector_dims`` (1-based, the batch axis is 0) are
    flattened into the vector, the remaining ones enumerate *positions*.  Shared by everything that runs a per-vector
    model over latents (``prior.CodebookPrior``, ``ot.LatentTransport``):

        batch_first=False                      [B, *size] -> [positions, B, dim]    one model per position
        batch_first=True, flatten_batch=True   [B, *size] -> [B * positions, dim]   one model for all positions
    """

    def __init__(self, size: Sequence[int], vector_dims: Sequence[int], what: str = "vector_dims", *,
                 batch_first: bool = False, flatten_batch: bool = False):
        self.size = tuple(int(s) for s in size)
        every = list(range(1, len(self.size) + 1))
        if not set(vector_dims).issubset(every):
            raise ValueError(f"latents of size {self.size} have {len(self.size) + 1} dimensions with the batch: `{what}` "
                             f"must be a subset of {every}, given {tuple(vector_dims)}")
        self.vector_dims = tuple(vector_dims)
        self.position_dims = tuple(d for d in every if d not in self.vector_dims)
        self.event_shape = torch.Size([self.size[d - 1] for d in self.vector_dims])
        self.batch_shape = torch.Size([self.size[d - 1] for d in self.position_dims])
        self.dim = int(np.prod(self.event_shape))
        self.positions = int(np.prod(self.batch_shape))
        self._how = dict(permute_dims=self.vector_dims, batch_first=batch_first,
                         flatten_batch=flatten_batch and len(self.position_dims) > 0)

    def split(self, latents: Tensor) -> Tensor:
        return permute_and_flatten(latents, **self._how)

    def join(self, vectors: Tensor) -> Tensor:
        return unflatten_and_unpermute(vectors, orig_shape=torch.Size([-1, *self.size]), **self._how)


def unsqueeze_like(tensor: Tensor, like: Tensor) -> Tensor:
    n = like.ndim - tensor.ndim
    if n < 0:
        raise ValueError(f"tensor.ndim={tensor.ndim} > like.ndim={like.ndim}")
    return tensor if n == 0 else tensor[(...,) + (None,) * n]


from .partial_checkpoint import PartialCheckpoint, human_format  # noqa: E402
