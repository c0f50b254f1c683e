"""Nearest-atom assignment of ``CodebookModel.energy/assign/predict`` in 'argmax' mode (reference
ot/distribution_models/codebook_model.py:150-160, base.py:216-233): the source of the VAE's discrete latent indices."""
import torch
from torch import Tensor

from .. import _lib
from .._lib import check, ptr, stream

__all__ = ["codebook_assign"]


def codebook_assign(x: Tensor, codebook: Tensor, temperature: float = 1.0):
    """x [*, B, d], codebook [*, K, d] -> (encodings [*, B, d] = codebook[idx], idx [*, B] int64) with
    idx = argmax_k softmax((1 / (|x - c_k|_2 + 1e-8)) / temperature)."""
    lib = _lib.load()
    _lib.require_cuda(x, "x")
    lead, bsz, d = x.shape[:-2], x.shape[-2], x.shape[-1]
    k = codebook.shape[-2]
    x3 = x.float().reshape(-1, bsz, d).contiguous()
    c3 = codebook.float().expand(*lead, k, d).reshape(-1, k, d).contiguous()
    nb = x3.shape[0]
    idx = torch.empty((nb, bsz), device=x.device, dtype=torch.int64)
    enc = torch.empty_like(x3)
    check(lib.otvae_codebook_assign(ptr(x3), ptr(c3), nb, bsz, k, d, float(temperature), ptr(idx), ptr(enc), stream()),
          "otvae_codebook_assign")
    return enc.reshape(*lead, bsz, d), idx.reshape(*lead, bsz)
