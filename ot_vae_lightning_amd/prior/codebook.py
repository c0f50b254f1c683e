"""``CodebookPrior``: vector-quantised latents (reference prior/codebook.py:20-117).  The latent tensor is cut into
vectors along ``embed_dims`` ([B, C, H, W] -> [positions, B, dim], one shared codebook for all positions), every vector is
replaced by its ``CodebookModel`` prediction (HIP assignment kernels; in training mode the model first takes its
streaming k-means update), and the one-hot modes pass the gradient straight through (``x + (e - x).detach()``).

What is differentiable here: the straight-through path and the 'l2' commitment term, i.e. everything the one-hot
modes need.  The assignment probabilities come from a HIP kernel without a backward pass, so the entropy losses
('kl', 'first_kl') and the soft 'mean' mode are available for evaluation only and raise ``NotImplementedError`` when
a gradient with respect to the input is requested."""
from math import cos, log, pi
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from .. import utils
from ..ot.distribution_models.codebook_model import CodebookModel
from .base import Prior

__all__ = ["CodebookPrior"]

_HARD = ("sample", "argmax")


class CodebookPrior(Prior):
    def __init__(self, latent_size: Sequence[int], embed_dims: Sequence[int], loss: Optional[str] = None,
                 temperature_annealing: Optional[int] = None, loss_coeff: float = 1., annealing_steps: int = 0,
                 **codebook_kwargs):
        super().__init__(loss_coeff=loss_coeff, annealing_steps=annealing_steps)
        latent_size = tuple(int(s) for s in latent_size)
        every = list(range(1, len(latent_size) + 1))
        if not set(embed_dims).issubset(every):
            raise ValueError(f"`latent_size`={latent_size}: inputs have {len(latent_size) + 1} dimensions with the batch; "
                             f"`embed_dims` must be a subset of {every}, given {tuple(embed_dims)}")
        if loss is not None and loss.lower() not in ("l2", "kl", "first_kl"):
            raise NotImplementedError(f"loss must be 'l2', 'kl' or 'first_kl'. Given: {loss}")
        self.size, self.embed_dims = torch.Size(latent_size), tuple(embed_dims)
        self.batch_dims = torch.Size([d for d in every if d not in self.embed_dims])
        self.event_shape = torch.Size([latent_size[d - 1] for d in self.embed_dims])
        self.batch_shape = torch.Size([latent_size[d - 1] for d in self.batch_dims])
        self.dimensionality = int(np.prod(self.event_shape))
        self._layout = dict(permute_dims=self.embed_dims, batch_first=False, flatten_batch=False)
        self.loss = loss
        self.codebook_model = CodebookModel(1, self.dimensionality, **codebook_kwargs)
        self.commitment_cost = 0. if self.codebook_model.training_mode in _HARD else 0.1
        self.temperature_annealing = temperature_annealing
        self.original_temperature = self.codebook_model.temperature

    @property
    def num_embeddings(self) -> int:
        return self.codebook_model.n_components

    def out_size(self, size):
        return size

    def permute_and_flatten(self, x: Tensor) -> Tensor:
        return utils.permute_and_flatten(x, **self._layout)

    def unflatten_and_unpermute(self, x: Tensor) -> Tensor:
        return utils.unflatten_and_unpermute(x, orig_shape=torch.Size([-1, *self.size]), **self._layout)

    def _compute_loss(self, x: Tensor, encodings: Tensor, dist) -> Tensor:
        kind = self.loss.lower() if self.loss is not None else None
        if x.dim() < encodings.dim():  # a latent embedded as a whole has no position axis; the shared codebook adds one
            x = x.expand_as(encodings)
        if kind is None:
            prior_loss = torch.zeros(x.size(-2), device=x.device).type_as(x)
        elif kind == "l2":
            prior_loss = F.mse_loss(x, encodings.detach(), reduction="none").mean(-1).sum(0)
        else:
            gap = log(self.num_embeddings) - dist.entropy()           # [positions, B]
            prior_loss = gap.sum(0) if kind == "kl" else gap[0]
        if self.commitment_cost > 0:
            prior_loss = prior_loss + self.commitment_cost * F.mse_loss(encodings, x.detach(), reduction="none").mean(-1).sum(0)
        return prior_loss

    def encode(self, x: Tensor) -> Prior.EncodingResults:
        model = self.codebook_model
        soft = model.mode not in _HARD
        if x.requires_grad and torch.is_grad_enabled() and (soft or (self.loss or "").lower() in ("kl", "first_kl")):
            raise NotImplementedError("the assignment probabilities have no backward pass on the MI355X path: the 'mean' mode "
                                      "and the 'kl' / 'first_kl' losses are evaluation-only (use a one-hot mode with "
                                      "loss=None or 'l2' for training)")
        x = self.permute_and_flatten(x)                               # [positions, B, dim]
        encodings, indices, dist = model(x.detach())                  # training: streaming k-means update, then predict
        encodings = encodings.type_as(x)
        prior_loss = self._compute_loss(x, encodings, dist)
        if model.training_mode in _HARD:
            encodings = x + (encodings - x).detach()                  # straight-through estimator
        encodings = self.unflatten_and_unpermute(encodings)
        dist.probs = dist.probs.transpose(0, 1)                       # [B, positions, K], as the reference hands them out
        return encodings, prior_loss, {"distribution": dist, "indices": indices.transpose(0, 1)}

    def sample(self, shape, device, mode: str = "sample") -> Tensor:
        index_shape = self.permute_and_flatten(torch.empty(*shape)).shape[:-1]
        atoms = self.codebook_model.distribution.sample(index_shape).squeeze(-2)
        return self.unflatten_and_unpermute(atoms).to(device)

    def forward(self, x: Tensor, step: int, **kwargs) -> Prior.EncodingResults:
        if self.temperature_annealing is not None and self.training:
            # the reference's expression (prior/codebook.py:115), evaluated as written there
            self.codebook_model.temperature = self.original_temperature * 0.5 * cos(pi * step / self.temperature_annealing) + 0.5
        return super().forward(x, step, **kwargs)
