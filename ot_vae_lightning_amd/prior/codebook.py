"""``CodebookPrior``: vector-quantised latents (reference prior/codebook.py:20-117).

The latent [B, *latent_size] is cut into vectors along ``embed_dims`` (``utils.VectorLayout``: [positions, B, dim], one
codebook shared by all positions) and every vector is replaced by what ``CodebookModel`` predicts for it -- the HIP
assignment kernels (``otvae_codebook_assign / _probs``), preceded in training mode by the model's streaming k-means
update (``otvae_codebook_kmeans``).  One-hot modes ('sample', 'argmax') pass the gradient straight through,
``x + (e - x).detach()``.

Regularisers, all per sample ([B]):

    loss=None        0
    loss='l2'        sum_positions mean_dim (x - sg[e])^2                       (pulls the encoder towards its atoms)
    loss='kl'        sum_positions ( log K - H[assignment distribution] )
    loss='first_kl'  the same for position 0 only
    + 0.1 * sum_positions mean_dim (sg[x] - e)^2   in the soft training mode    (commitment of the atoms)

Differentiable on this path: the straight-through estimator, the two squared-error terms, and -- through
``otvae_codebook_probs_bwd`` -- the assignment probabilities, so the soft 'mean' mode and the entropy losses train the
encoder as in the reference (the codebook itself moves by its streaming k-means update, not by gradient).
"""
import math
from typing import Optional, Sequence

import torch
from torch import Tensor

from .. import utils
from ..ot.distribution_models.codebook_model import CodebookModel
from .base import Prior

__all__ = ["CodebookPrior"]

_ONE_HOT_MODES = ("sample", "argmax")
_ENTROPY_LOSSES = ("kl", "first_kl")
_LOSSES = ("l2",) + _ENTROPY_LOSSES


def _sq_err(moving: Tensor, fixed: Tensor) -> Tensor:
    """[positions, B, dim] x2 -> [B]: squared error averaged over the vector, summed over positions; ``fixed`` gets no gradient"""
    return (moving - fixed.detach()).square().mean(-1).sum(0)


class CodebookPrior(Prior):
    def __init__(self, latent_size: Sequence[int], embed_dims: Sequence[int], loss: Optional[str] = None,
                 temperature_annealing: Optional[int] = None, loss_coeff: float = 1., annealing_steps: int = 0,
                 **codebook_kwargs):
        super().__init__(loss_coeff=loss_coeff, annealing_steps=annealing_steps)
        if loss is not None and loss.lower() not in _LOSSES:
            raise NotImplementedError(f"loss must be 'l2', 'kl' or 'first_kl'. Given: {loss}")
        self.layout = utils.VectorLayout(latent_size, embed_dims, "embed_dims")
        self.loss = loss
        self.codebook_model = CodebookModel(1, self.layout.dim, **codebook_kwargs)
        self.commitment_cost = 0. if self.codebook_model.training_mode in _ONE_HOT_MODES else 0.1
        self.temperature_annealing = temperature_annealing
        self.original_temperature = self.codebook_model.temperature

    # ---- the reference's public attributes, read off the layout
    size = property(lambda self: torch.Size(self.layout.size))
    embed_dims = property(lambda self: self.layout.vector_dims)
    batch_dims = property(lambda self: torch.Size(self.layout.position_dims))
    event_shape = property(lambda self: self.layout.event_shape)
    batch_shape = property(lambda self: self.layout.batch_shape)
    dimensionality = property(lambda self: self.layout.dim)
    num_embeddings = property(lambda self: self.codebook_model.n_components)

    def permute_and_flatten(self, x: Tensor) -> Tensor:
        return self.layout.split(x)

    def unflatten_and_unpermute(self, x: Tensor) -> Tensor:
        return self.layout.join(x)

    def out_size(self, size):
        return size

    def _loss_kind(self) -> Optional[str]:
        return None if self.loss is None else self.loss.lower()

    def _compute_loss(self, x: Tensor, encodings: Tensor, dist) -> Tensor:
        if x.dim() < encodings.dim():       # a latent embedded as a whole has no position axis; the shared codebook adds one
            x = x.expand_as(encodings)
        kind = self._loss_kind()
        if kind == "l2":
            total = _sq_err(x, encodings)
        elif kind in _ENTROPY_LOSSES:
            information = math.log(self.num_embeddings) - dist.entropy()        # [positions, B]
            total = information[0] if kind == "first_kl" else information.sum(0)
        else:
            total = x.new_zeros(x.size(-2))
        if self.commitment_cost > 0:
            total = total + self.commitment_cost * _sq_err(encodings, x)
        return total

    def encode(self, x: Tensor) -> Prior.EncodingResults:
        vectors = self.layout.split(x)                                           # [positions, B, dim]
        model = self.codebook_model
        if model.training and not model.update_with_autograd:
            model.update(vectors.detach())                                       # streaming k-means, no gradient
        # the probabilities carry a gradient to the encoder where something differentiable is made of them
        soft = model.mode not in _ONE_HOT_MODES or self._loss_kind() in _ENTROPY_LOSSES
        atoms, indices, dist = model.predict(vectors if soft else vectors.detach())
        atoms = atoms.type_as(vectors)
        prior_loss = self._compute_loss(vectors, atoms, dist)
        if self.codebook_model.training_mode in _ONE_HOT_MODES:
            atoms = vectors + (atoms - vectors).detach()
        dist.probs = dist.probs.transpose(0, 1)                                  # handed out as [B, positions, K]
        return self.layout.join(atoms), prior_loss, {"distribution": dist, "indices": indices.transpose(0, 1)}

    def sample(self, shape, device, mode: str = "sample") -> Tensor:
        # one draw per (position, batch entry); only the index shape matters, so a meta tensor stands in for the latent
        index_shape = self.layout.split(torch.empty(*shape, device="meta")).shape[:-1]
        atoms = self.codebook_model.distribution.sample(index_shape).squeeze(-2)
        return self.layout.join(atoms).to(device)

    def forward(self, x: Tensor, step: int, **kwargs) -> Prior.EncodingResults:
        if self.training and self.temperature_annealing is not None:
            # the schedule exactly as the reference evaluates it (prior/codebook.py:115-116): T0/2 * cos(pi*t/T) + 1/2
            phase = math.cos(math.pi * step / self.temperature_annealing)
            self.codebook_model.temperature = self.original_temperature * 0.5 * phase + 0.5
        return super().forward(x, step, **kwargs)
