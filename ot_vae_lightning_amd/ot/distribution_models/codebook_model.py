"""``CodebookModel`` / ``CategoricalEmbeddings`` with the reference's contract
(ot/distribution_models/codebook_model.py:25-214, mixture behaviour from base.py:165-262): a set of K atoms per
leading index, fitted by (EMA) k-means on streaming batches, queried by nearest-atom assignment.

MI355X path: the O(B*K*d) work -- energies, arg-max assignment, the assignment distribution and the k-means sufficient
statistics -- runs in HIP kernels (``otvae_codebook_assign / _probs / _kmeans``); the O(K*d) buffer arithmetic that
follows is a handful of tiny tensor expressions kept in the reference's own order (boolean-mask updates, Laplace
smoothing over the observed atoms), and ``w2`` composes the HIP Sinkhorn solver exactly as the reference does.

Supported: ``metric='euclidean'`` (the hot path: p = 2, energies never materialised) and ``'cosine'`` / any ``p > 0`` / ``topk``
(energies materialised by ``otvae_codebook_energy``, then ``MixtureMixin.assign``'s top-k restriction and soft-max), the assignment modes ``'argmax'``, ``'sample'`` (one-hot) and
``'mean'`` (soft: the assignment distribution itself, as ``DiscreteTransport`` uses it in the reference's
tests/test_latent_transport.py:92-101; its weighted sums ``probs^T @ samples`` go through ``matrix_utils.mm`` =
``otvae_gemm_f32``), the Gumbel modes (``gumbel_weights``: ``otvae_softmax_rows`` with injectable draws), and
``update_with_autograd=True`` (the codebook as a trained parameter: gradients reach it through ``weights @ codebook`` and,
in the soft modes and entropy losses, through the assignment probabilities, ``otvae_codebook_probs_bwd_atoms``).  ``energy`` (atoms against atoms, K x K, inside ``w2`` only) keeps ``torch.cdist``: its exact
zero diagonal is what the reference's 1 / (dist + 1e-8) cost sees, which the |x|^2 + |y|^2 - 2 x.y kernel does not reproduce."""
from functools import partial
from typing import Optional, Tuple

import torch
import torch.distributions as D
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from ... import _lib, utils
from ..._lib import check, ptr, stream
from ..matrix_utils import mm, softmax_rows
from ..w2_utils import sinkhorn_log
from .base import MIXTURE_MODES, DistributionModel, gumbel_weights

__all__ = ["CodebookModel", "CategoricalEmbeddings"]


class CategoricalEmbeddings(D.Categorical):
    """Categorical over K atoms that samples / averages the atoms themselves (codebook_model.py:25-63)."""

    def __init__(self, embeddings: Tensor, probs: Optional[Tensor] = None, logits: Optional[Tensor] = None) -> None:
        super().__init__(probs, logits)
        self.embeddings = embeddings
        if self.probs.shape != self.embeddings.shape[:-1]:
            raise ValueError("`probs` and `embeddings` should have the same leading dimensions")

    def _select(self, weights: Tensor) -> Tensor:
        return mm(weights.unsqueeze(-2).type_as(self.embeddings), self.embeddings).squeeze(-2)

    def _select_one_hot(self, index_list: Tensor) -> Tensor:
        return self._select(F.one_hot(index_list, self._num_events).type_as(index_list))

    @property
    def mean(self):
        return self._select(self.probs)

    @property
    def mode(self):
        return self._select_one_hot(self.probs.argmax(-1))

    def sample(self, sample_shape=torch.Size()) -> Tensor:
        return self._select_one_hot(super().sample(sample_shape))




class _AssignmentProbsFn(torch.autograd.Function):
    """softmax((1 / (|x - c_k| + 1e-8)) / T) (+ its entropy) with a backward pass to the samples (``otvae_codebook_probs_bwd``):
    what lets CodebookPrior's soft 'mean' mode and its entropy losses train.  The codebook gets a gradient only when it is a
    trained parameter (requires_grad = update_with_autograd, reference codebook_model.py:89): ``otvae_codebook_probs_bwd_atoms``."""

    @staticmethod
    def forward(ctx, x3, c3, temperature, with_entropy):
        lib = _lib.load()
        nb, bsz, d = x3.shape
        K = c3.shape[1]
        probs = torch.empty((nb, bsz, K), device=x3.device, dtype=torch.float32)
        ent = torch.empty((nb, bsz), device=x3.device, dtype=torch.float32) if with_entropy else None
        check(lib.otvae_codebook_probs(ptr(x3), ptr(c3), nb, bsz, K, d, float(temperature), ptr(probs), ptr(ent), stream()),
              "otvae_codebook_probs")
        ctx.save_for_backward(x3, c3, probs)
        ctx.temperature = float(temperature)
        ctx.set_materialize_grads(False)
        return (probs, ent) if with_entropy else probs

    @staticmethod
    def backward(ctx, gprobs, gent=None):
        x3, c3, probs = ctx.saved_tensors
        if gprobs is None and gent is None:
            return None, None, None, None
        nb, bsz, d = x3.shape
        gx = torch.empty_like(x3)
        gp = gprobs.contiguous().float() if gprobs is not None else None
        ge = gent.contiguous().float() if gent is not None else None
        if ctx.needs_input_grad[1]:  # update_with_autograd: the atoms are trained too
            K = c3.shape[1]
            coef = torch.empty((nb, bsz, K), device=x3.device, dtype=torch.float32)
            gc = torch.empty_like(c3)
            check(_lib.load().otvae_codebook_probs_bwd_atoms(ptr(x3), ptr(c3), ptr(probs), ptr(gp), ptr(ge), nb, bsz, K, d,
                                                            ctx.temperature, ptr(gx), ptr(coef), ptr(gc), stream()),
                  "otvae_codebook_probs_bwd_atoms")
            return gx, gc, None, None
        check(_lib.load().otvae_codebook_probs_bwd(ptr(x3), ptr(c3), ptr(probs), ptr(gp), ptr(ge), nb, bsz, c3.shape[1], d,
                                                  ctx.temperature, ptr(gx), stream()), "otvae_codebook_probs_bwd")
        return gx, None, None, None


class _CodebookEnergyFn(torch.autograd.Function):
    """``CodebookModel.energy`` for the metrics besides the hot path's euclidean p = 2 (codebook_model.py:155-168) on
    ``otvae_codebook_energy`` / ``_bwd``: raw energies [nb, B, K], differentiable in the samples and the atoms."""

    @staticmethod
    def forward(ctx, x3, c3, metric, p):
        nb, bsz, d = x3.shape
        K = c3.shape[1]
        e = torch.empty((nb, bsz, K), device=x3.device, dtype=torch.float32)
        check(_lib.load().otvae_codebook_energy(ptr(x3), ptr(c3), nb, bsz, K, d, int(metric), float(p), ptr(e), stream()),
              "otvae_codebook_energy")
        ctx.save_for_backward(x3, c3)
        ctx.cfg = (int(metric), float(p))
        return e

    @staticmethod
    def backward(ctx, ge):
        x3, c3 = ctx.saved_tensors
        metric, p = ctx.cfg
        nb, bsz, d = x3.shape
        gx = torch.empty_like(x3) if ctx.needs_input_grad[0] else None
        gc = torch.empty_like(c3) if ctx.needs_input_grad[1] else None
        if gx is None and gc is None:
            return None, None, None, None
        check(_lib.load().otvae_codebook_energy_bwd(ptr(x3), ptr(c3), ptr(ge.contiguous().float()), nb, bsz, c3.shape[1], d, metric, p,
                                                   ptr(gx), ptr(gc), stream()), "otvae_codebook_energy_bwd")
        return gx, gc, None, None


def topk_energy(energy: Tensor, topk: Optional[int]) -> Tensor:
    """``MixtureMixin.assign``'s restriction to the k largest energies per sample (base.py:217-220): everything else becomes -inf,
    i.e. weight 0 after the soft-max; gradients reach the kept entries only.  (Index glue on a [*, B, K] tensor, as the reference.)"""
    if topk is None or topk <= 0:
        return energy
    val, idx = torch.topk(energy, int(topk), dim=-1)
    return torch.full_like(energy, float("-inf")).scatter(-1, idx, val)


class CodebookModel(DistributionModel):
    Distribution = CategoricalEmbeddings

    def __init__(self, *size: int, mixture_cfg={}, **kwargs) -> None:
        cfg = dict(n_components=None, metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax",
                   inference_mode="argmax", kmeans_iter=100, laplace_eps=1e-5)
        unknown = set(mixture_cfg) - set(cfg)
        if unknown:
            raise TypeError(f"unexpected mixture_cfg keys: {sorted(unknown)}")
        cfg.update(mixture_cfg)
        if cfg["n_components"] is None:
            raise TypeError("mixture_cfg must give `n_components`")
        if cfg["metric"] not in ("euclidean", "cosine"):   # (the reference raises from `energy`, codebook_model.py:168)
            raise NotImplementedError(f"Supported `metric`: 'cosine', 'euclidean'. Got `metric`={cfg['metric']}")
        if not float(cfg["p"]) > 0:
            raise ValueError("`p` must be positive")
        for m in (cfg["training_mode"], cfg["inference_mode"]):
            if m not in MIXTURE_MODES:
                raise NotImplementedError(f"assignment mode {m!r}: expected one of {MIXTURE_MODES}")
        self.n_components = int(cfg["n_components"])
        self.metric, self.p, self.topk = cfg["metric"], float(cfg["p"]), cfg["topk"]
        self.temperature = float(cfg["temperature"])
        self.training_mode, self.inference_mode = cfg["training_mode"], cfg["inference_mode"]
        self.kmeans_iter = int(cfg["kmeans_iter"])
        self.laplace_smoothing = partial(utils.laplace_smoothing, n_categories=self.n_components, eps=cfg["laplace_eps"])
        DistributionModel.__init__(self, *size, **kwargs)
        w = torch.ones(*self.leading_shape, self.n_components)
        self.register_buffer("weight_init", (w / w.sum(-1, keepdim=True)).type_as(self.vec_init))
        # update_with_autograd: the atoms are a trained parameter and the k-means buffers are not created (codebook_model.py:89-93)
        self.codebook = nn.Parameter(self.vec_init.clone(), requires_grad=self.update_with_autograd)
        if not self.update_with_autograd:
            self.register_buffer("_running_sum", torch.zeros_like(self.vec_init))
            self.register_buffer("_n_obs", torch.zeros(*self.leading_shape, self.n_components).type_as(self.vec_init))

    # ---- shapes / distributions
    @property
    def vec_shape(self):
        return (*self.leading_shape, self.n_components, self.dim)

    @property
    def weights(self) -> Tensor:
        if not hasattr(self, "_n_obs") or torch.allclose(self._n_obs, torch.zeros_like(self._n_obs)):
            return self.weight_init.type_as(self.codebook)
        return self._n_obs.type_as(self.codebook) / self._n_obs.sum(-1, keepdim=True)

    @property
    def distribution(self) -> CategoricalEmbeddings:
        return CategoricalEmbeddings(self.codebook, probs=self.weights)

    @property
    def batched_distribution(self) -> CategoricalEmbeddings:
        return CategoricalEmbeddings(self.codebook.unsqueeze(-3), probs=self.weights.unsqueeze(-2))

    @torch.no_grad()
    def reset(self) -> None:
        self.codebook.copy_(self.vec_init)
        if self.update_with_autograd:
            return
        self._running_sum.zero_()
        self._n_obs.zero_()

    # ---- HIP-side pieces
    def _flat(self, samples: Tensor):
        """samples [*lead', B, d] (lead' broadcastable to leading_shape) -> ([nb, B, d] fp32, [nb, K, d] fp32, lead)"""
        _lib.require_cuda(samples, "samples")
        lead = torch.broadcast_shapes(samples.shape[:-2], self.leading_shape)
        bsz = samples.shape[-2]
        x3 = samples.float().expand(*lead, bsz, self.dim).reshape(-1, bsz, self.dim).contiguous()
        cb = self.codebook if self.update_with_autograd else self.codebook.detach()   # a trained codebook stays in the graph
        c3 = cb.float().expand(*lead, self.n_components, self.dim).reshape(-1, self.n_components, self.dim).contiguous()
        return x3, c3, lead

    @property
    def _general(self) -> bool:
        """anything but the hot path's euclidean p = 2 energies without topk: the energies are materialised
        (``otvae_codebook_energy``), restricted to the top k, and normalised by ``otvae_softmax_rows``"""
        return self.metric != "euclidean" or self.p != 2.0 or (self.topk is not None and self.topk > 0)

    def _energy3(self, samples: Tensor):
        x3, c3, lead = self._flat(samples)
        return _CodebookEnergyFn.apply(x3, c3, 0 if self.metric == "euclidean" else 1, self.p), x3, c3, lead

    def _argmax(self, samples: Tensor) -> Tuple[Tensor, Tensor]:
        lib = _lib.load()
        if self._general:
            e, x3, c3, lead = self._energy3(samples)
            idx = topk_energy(e, self.topk).argmax(-1)                                  # [nb, B]
            enc = torch.gather(c3, 1, idx.unsqueeze(-1).expand(-1, -1, self.dim))
            return enc.reshape(*lead, x3.shape[1], self.dim), idx.reshape(*lead, x3.shape[1])
        x3, c3, lead = self._flat(samples)
        nb, bsz = x3.shape[0], x3.shape[1]
        idx = torch.empty((nb, bsz), device=x3.device, dtype=torch.int64)
        enc = torch.empty_like(x3)
        check(lib.otvae_codebook_assign(ptr(x3), ptr(c3), nb, bsz, self.n_components, self.dim, self.temperature, ptr(idx),
                                        ptr(enc), stream()), "otvae_codebook_assign")
        return enc.reshape(*lead, bsz, self.dim), idx.reshape(*lead, bsz)

    def assignment_probs(self, samples: Tensor, with_entropy: bool = False):
        """softmax(energy / temperature) [*, B, K] (and its entropy [*, B]); differentiable with respect to ``samples``"""
        if self._general:
            e, x3, c3, lead = self._energy3(samples)
            probs = softmax_rows(topk_energy(e, self.topk), 1.0 / self.temperature).reshape(*lead, x3.shape[1], self.n_components)
            return (probs, D.Categorical(probs).entropy()) if with_entropy else probs
        x3, c3, lead = self._flat(samples)
        bsz = x3.shape[1]
        out = _AssignmentProbsFn.apply(x3, c3, self.temperature, with_entropy)
        if with_entropy:
            return out[0].reshape(*lead, bsz, self.n_components), out[1].reshape(*lead, bsz)
        return out.reshape(*lead, bsz, self.n_components)

    def energy(self, samples: Tensor) -> Tensor:
        """1 / (|x - c_k|_2 + 1e-8) [*, B, K] (codebook_model.py:150-156); only the tiny atoms-vs-atoms case of ``w2``
        goes through here, the assignment kernels never materialise it."""
        self._validate_samples(samples)
        # every metric / p, the default euclidean p = 2 included, on otvae_codebook_energy (differentiable: the Gumbel assignment modes
        # train through this value); rounds 1-3 kept the default case on torch.cdist.  The kernel computes in fp32: a double-precision
        # codebook (the atoms-vs-atoms distances of a float64 DiscreteTransport) keeps its precision on cdist
        if self.codebook.dtype != torch.float32 and self.metric == "euclidean" and self.p == 2.0:
            return 1 / (torch.cdist(samples.type_as(self.codebook), self.codebook, self.p) + 1e-8)
        e, x3, _, lead = self._energy3(samples)
        return e.reshape(*lead, x3.shape[1], self.n_components).type_as(self.codebook)

    @property
    def mode(self) -> str:
        return self.training_mode if self.training else self.inference_mode

    def assign(self, samples: Tensor):
        """(assignment weights [*, B, K], sampled indices [*, B], Categorical(softmax weights)) -- base.py:206-239.
        'argmax' / 'sample': one-hot weights of the nearest / the sampled atom; 'mean': the softmax weights themselves.
        As in the reference the returned ``indices`` are a draw from the distribution (they come from this device's
        generator, so they are not comparable across devices); the deterministic nearest-atom indices are
        ``nearest(samples)[1]``."""
        probs = self.assignment_probs(samples)
        distribution = D.Categorical(probs)
        indices = distribution.sample()
        mode = self.mode
        if mode == "mean" or self.topk == 1:   # (base.py:228: with one atom left the soft weights are the one-hot weights)
            weights = probs
        elif mode == "sample":
            weights = F.one_hot(indices, self.n_components).type_as(probs)
        elif "gumbel" in mode:
            noise, self.gumbel_noise = getattr(self, "gumbel_noise", None), None   # injected draws are used once
            weights = gumbel_weights(topk_energy(self.energy(samples), self.topk), self.temperature, "hard" in mode, noise).type_as(probs)
        else:
            _, idx = self._argmax(samples)
            weights = F.one_hot(idx, self.n_components).type_as(probs)
        return weights, indices, distribution

    def predict(self, features: Tensor):
        """(weights @ codebook, sampled indices, assignment distribution) -- codebook_model.py:145-148.  In 'argmax'
        mode the product with a one-hot matrix is the gather the assignment kernel already did."""
        self._validate_samples(features)
        if self.mode == "argmax" and not self.update_with_autograd and self.topk != 1:
            preds, _ = self._argmax(features)
            distribution = D.Categorical(self.assignment_probs(features))
            return preds.type_as(self.codebook), distribution.sample(), distribution
        weights, indices, distribution = self.assign(features)
        return mm(weights.type_as(self.codebook), self.codebook), indices, distribution

    def nearest(self, features: Tensor) -> Tuple[Tensor, Tensor]:
        """(codebook[argmax], argmax indices): the deterministic part of ``predict``"""
        return self._argmax(features)

    def kmean_iteration(self, samples: Optional[Tensor]):
        if samples is None:
            return self._n_obs, self._running_sum
        lib = _lib.load()
        x3, c3, lead = self._flat(samples)
        nb, bsz = x3.shape[0], x3.shape[1]
        mode = self.mode
        if mode == "mean":  # soft assignment: sums of the probabilities and probs^T @ samples (base.py:241-251)
            probs = self.assignment_probs(samples).reshape(nb, bsz, self.n_components)
            counts, sums = probs.sum(-2), mm(probs.transpose(-1, -2).contiguous(), x3)
            return (counts.reshape(*lead, self.n_components).type_as(self._n_obs),
                    sums.reshape(*lead, self.n_components, self.dim).type_as(self._running_sum))
        if mode == "sample":
            idx = D.Categorical(self.assignment_probs(samples)).sample()
        else:
            _, idx = self._argmax(samples)
        idx = idx.reshape(nb, bsz).contiguous()
        counts = torch.empty((nb, self.n_components), device=x3.device, dtype=torch.float32)
        sums = torch.empty((nb, self.n_components, self.dim), device=x3.device, dtype=torch.float32)
        check(lib.otvae_codebook_kmeans(ptr(x3), ptr(idx), nb, bsz, self.n_components, self.dim, ptr(counts), ptr(sums),
                                        stream()), "otvae_codebook_kmeans")
        return (counts.reshape(*lead, self.n_components).type_as(self._n_obs),
                sums.reshape(*lead, self.n_components, self.dim).type_as(self._running_sum))

    # ---- fitting (codebook_model.py:121-143, 189-214)
    def _no_buffers(self, what: str):
        raise RuntimeError(f"`update_with_autograd` is True: the codebook is trained with autograd; the k-means buffers `{what}` feeds "
                           "were not created (the reference warns, base.py:82-90, then fails on the missing buffers)")

    @torch.no_grad()
    def update(self, samples: Tensor) -> None:
        self._validate_samples(samples)
        if self.update_with_autograd:
            self._no_buffers("update")
        samples = samples.detach().type_as(self._running_sum)
        self._init_parameters(samples)
        res = self.kmean_iteration(samples)
        if self.reduce_on_update:
            res = [self.reduce(r) for r in res]
        buffers = self._update_buffers(*res, decay=True)
        self._update_parameters(*buffers)

    @torch.no_grad()
    def fit(self, samples: Optional[Tensor] = None) -> None:
        if self.update_with_autograd:
            self._no_buffers("fit")
        if samples is not None:
            self._validate_samples(samples)
            samples = samples.detach().type_as(self._running_sum)
            self._init_parameters(samples)
        res = None
        for _ in range(self.kmeans_iter):
            res = self.kmean_iteration(samples)
            self._update_parameters(*[self.reduce(r) for r in res])
            if samples is None:
                break  # the statistics are the stored buffers: every further iteration recomputes the same codebook
        if self.kmeans_iter > 0:
            self._update_buffers(*res, decay=False)

    def _update_parameters(self, weights_sum: Tensor, samples_sum: Tensor) -> None:
        hit = weights_sum > 1e-8  # only the observed atoms move
        self.codebook.data[hit] = (samples_sum[hit] / self.laplace_smoothing(weights_sum[hit]).unsqueeze(-1)) \
            .type_as(self.codebook)

    def _update_buffers(self, weights_sum: Tensor, samples_sum: Tensor, decay: bool = False):
        hit = weights_sum > 1e-8
        if decay:
            self._n_obs[hit] = self.ema_update(self._n_obs[hit], weights_sum[hit])
            self._running_sum[hit] = self.ema_update(self._running_sum[hit], samples_sum[hit])
        else:
            self._n_obs[hit] = weights_sum[hit]
            self._running_sum[hit] = samples_sum[hit]
        return self._n_obs, self._running_sum

    def _init_parameters(self, samples: Tensor) -> None:
        if torch.allclose(self.codebook, self.vec_init):
            rand_indices = torch.randperm(samples.size(-2))[:self.n_components]  # host generator, as the reference
            self.codebook.copy_(samples[..., rand_indices.to(samples.device), :])
            self._n_obs += 1

    def w2(self, other: CategoricalEmbeddings) -> Tensor:
        """entropic OT between the two atom sets (codebook_model.py:175-182)"""
        cost = 1 / (self.energy(other.embeddings) + 1e-8)
        plan = sinkhorn_log(self.distribution.probs, other.probs, cost, reg=1e-5, max_iter=100, threshold=1e-3)
        return torch.sum(cost * plan, dim=(-2, -1))

    def extra_repr(self) -> str:
        return (DistributionModel.extra_repr(self) + f", num_components={self.n_components}, metric={self.metric}, "
                f"p={self.p}, temperature={self.temperature}, training_mode={self.training_mode}, "
                f"inference_mode={self.inference_mode}")
