"""``FilterSequential`` and ``QKVAttention`` with the reference's interface (networks/nets_utils.py:10-19,55-82),
executing on the MI355X attention kernel."""
import inspect

import torch
import torch.nn as nn

from .. import functional as HF

__all__ = ["FilterSequential", "QKVAttention"]


def _accepts(module: nn.Module, name: str) -> bool:
    return name in inspect.signature(module.forward).parameters


class FilterSequential(nn.Sequential):
    """Sequential that forwards a keyword argument only to the layers whose ``forward`` declares it.
    The reference re-inspects every layer's signature on every call (nets_utils.py:15-19, utils/__init__.py:78-109);
    here the per-layer decision is cached the first time a keyword set is seen."""

    def forward(self, x, **kwargs):
        keys = tuple(sorted(kwargs))
        cache = self.__dict__.setdefault("_kw_cache", {})
        plan = cache.get(keys)
        if plan is None or len(plan) != len(self):
            plan = [tuple(k for k in keys if _accepts(layer, k)) for layer in self]
            cache[keys] = plan
        for layer, ks in zip(self, plan):
            x = layer(x, **{k: kwargs[k] for k in ks})
        return x


class QKVAttention(nn.Module):
    """QKV attention over ``[N, (G,) 3*H*C, T]`` -> ``[N, G*H*C, T]`` (no mask, no residual, q and k both scaled by
    C**-0.5).  A 4-D channels-last ``[N, 3*H*C, Hs, Ws]`` tensor is accepted as well (the AttentionBlock fast path,
    no layout change)."""

    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = int(n_heads)

    def forward(self, qkv):
        if qkv.dim() == 4 and HF.is_nhwc(qkv) and qkv.shape[1] % (3 * self.n_heads) == 0 and not getattr(self, "_grouped", False):
            return HF.qkv_attention(qkv, self.n_heads)
        if qkv.dim() == 3:
            qkv = qkv.unsqueeze(1)
        bs, groups, width, length = qkv.shape
        assert width % (3 * self.n_heads) == 0, \
            f"tensor width: {width} must be divisible by (3 * n_heads): {3 * self.n_heads}"
        a = HF.qkv_attention(qkv.reshape(bs * groups, width, length), self.n_heads)
        return a.reshape(bs, -1, length)
