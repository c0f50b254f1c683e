"""``VAE(encoder=, decoder=, prior=)`` / ``VAE(autoencoder=, prior=)`` with the reference's plug-in contract
(model/vae.py:38-270): ``nelbo`` (the training loss), ``encode``, ``decode``, ``forward``, ``sample``,
``latent_size``, ``configure_optimizers``.  The reduction mse + mean(prior)/(C*H*W) is one fused kernel."""
import functools
import itertools
from typing import Dict, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from .. import functional as HF
from .. import utils
from ..prior import Prior
from ..utils import FilterKwargs
from .base import VisionModule

__all__ = ["VAE"]


class VAE(VisionModule):
    Batch = Dict[str, Union[Tensor, Dict]]

    def __init__(self, *base_args, monitor: str = "psnr", mode: str = "max", prior: Optional[Prior] = None,
                 autoencoder: Optional[nn.Module] = None, encoder: Optional[nn.Module] = None,
                 decoder: Optional[nn.Module] = None, conditional: bool = False, expansion: int = 1,
                 **base_kwargs) -> None:
        super().__init__(*base_args, monitor=monitor, mode=mode, **base_kwargs)
        if autoencoder is None and (encoder is None or decoder is None):
            raise ValueError("At least one of `autoencoder` or (`encoder`, `decoder`) parameters must be set")
        if autoencoder is not None and (encoder is not None or decoder is not None):
            raise ValueError("Setting both `autoencoder` and `encoder` or `decoder` is ambiguous")
        self.conditional, self.expansion = conditional, expansion
        self.hparams.conditional, self.hparams.expansion = conditional, expansion
        self.loss = self.nelbo
        self.prior = prior
        if autoencoder is not None:
            assert isinstance(autoencoder, nn.Module) and hasattr(autoencoder, "encode") and hasattr(autoencoder, "decode"), \
                "Parameter `autoencoder` should be a nn.Module and implement the methods `encode` and `decode`"
            self.autoencoder = autoencoder
            self._encode_func, self._decode_func = autoencoder.encode, autoencoder.decode
        else:
            self.encoder, self.decoder = encoder, decoder
            self._encode_func, self._decode_func = encoder, decoder
        self._expand = functools.partial(utils.replicate_batch, n=expansion)
        self._reduce_mean = functools.partial(utils.mean_replicated_batch, n=expansion)
        self._reduce_std = functools.partial(utils.std_replicated_batch, n=expansion)
        # forward a keyword only to the plug-ins whose signature declares it (reference: FilterKwargs on 'labels',
        # model/vae.py:50-51,209; generalised to every keyword so that priors can take e.g. `eps` / `prior_samples`)
        self._filter = lambda callee, keys=("labels", "eps"): FilterKwargs(callee, arg_keys=list(keys))

    # -- reference API -------------------------------------------------------------------------------------------
    def batch_preprocess(self, batch) -> Batch:
        samples, labels = batch
        return {"samples": samples, "target": samples, "kwargs": {"labels": labels} if self.conditional else {}}

    @VisionModule.postprocess
    @VisionModule.preprocess
    def forward(self, samples: Tensor, expand: bool = False, **kwargs) -> Tensor:
        latents = self.encode(samples, expand=expand, no_preprocess_override=True, **kwargs)
        return self.decode(latents, expand_kwargs=expand, no_postprocess_override=True, **kwargs)

    def optim_parameters(self):
        groups = [self.autoencoder.parameters()] if hasattr(self, "autoencoder") else \
            [self.encoder.parameters(), self.decoder.parameters()]
        if self.prior is not None:
            groups.append(self.prior.parameters())
        return filter(lambda p: p.requires_grad, itertools.chain(*groups))

    def configure_optimizers(self):
        """Reference: Adam(lr 1e-3, betas (.9,.999)) + ReduceLROnPlateau (model/vae.py:148-156).  The MI355X
        training engine (``engine.HipTrainer``) runs the same update as one fused kernel over a flat buffer."""
        opt = torch.optim.Adam(self.optim_parameters(), lr=1e-3, betas=(0.9, 0.999))
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode=self.mode, factor=0.75, patience=8, threshold=1e-1,
                                                           min_lr=1e-6)
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "monitor": self.monitor}}

    def enable_graphed_step(self, warmup: int = 2) -> "VAE":
        """``training_step`` -> ``loss.backward()`` -> any optimizer at hipGraph speed: ``self.loss`` (what the reference's
        ``training_step`` calls, model/base.py:122-129) becomes an ``engine.GraphedNelbo`` over ``self.nelbo`` -- one autograd node
        that replays a captured forward graph and, in ``backward``, a captured backward graph; gradients land in ``p.grad``."""
        from ..engine.graphed import GraphedNelbo
        self.disable_graphed_step()
        self.loss = GraphedNelbo(self, warmup=warmup)
        return self

    def disable_graphed_step(self) -> "VAE":
        """Back to the eagerly issued ``nelbo``; the captured graphs are released now (``GraphedNelbo.close``).  Not required before
        dropping the model: a model that simply goes out of scope releases them the same way (engine/lifetime.py)."""
        cur = self.__dict__.get("loss")
        if cur is not None and hasattr(cur, "close"):
            cur.close()
            self.loss = self.nelbo
        return self

    def recon_loss(self, reconstructions: Tensor, target: Tensor, **kwargs) -> Tensor:
        return HF.nelbo_loss(reconstructions, target, None)[1]

    def prior_loss(self, prior_loss: Tensor, prior_artifacts, **kwargs) -> Tensor:
        return prior_loss.mean()

    def nelbo(self, batch: Batch, batch_idx: int) -> Tuple[Tensor, Dict[str, Tensor], Batch]:
        samples, target, kwargs = batch["samples"], batch["target"], batch["kwargs"]
        batch_size = samples.size(0)
        latents, prior_loss, prior_artifacts = self.encode(samples, expand=True, return_prior_artifacts=True, **kwargs)
        reconstructions = self.decode(latents, expand_kwargs=True, **kwargs)
        reconstructions_mean = self._reduce_mean(reconstructions)
        out3 = HF.nelbo_loss(reconstructions_mean, target, prior_loss)   # [total, recon, prior/(C*H*W)]
        self._last_out3 = out3.detach()
        self._last_nelbo = out3 if out3.requires_grad else None  # engine.HipTrainer seeds the backward pass here
        loss = out3[0]
        logs = {"train/loss/total": loss, "train/loss/recon": out3[1], "train/loss/prior": out3[2]}
        artifacts = {"preds": reconstructions[:batch_size], "latents": latents[:batch_size],
                     "preds_mean": reconstructions_mean}
        return loss, logs, {**batch, **artifacts, **prior_artifacts}

    @property
    def latent_size(self):
        enc_out = self.autoencoder.latent_size if hasattr(self, "autoencoder") else self.encoder.out_size
        return enc_out if self.prior is None else self.prior.out_size(enc_out)

    @VisionModule.preprocess
    def encode(self, samples: Tensor, return_prior_artifacts: bool = False, expand: bool = False, **kwargs):
        with self._filter(self._encode_func, kwargs.keys()) as encode:
            encodings = encode(samples, **kwargs)
        # handle for a two-phase backward (engine.HipTrainer overlaps the gradient all-reduce of everything downstream
        # of the encoder with the encoder's own backward): the ONE tensor through which the loss depends on the encoder
        self._last_cut = encodings if encodings.requires_grad else None
        if expand:
            # an explicit noise tensor (`eps`, this package's extension of GaussianPrior.forward) holds one draw per REPLICATED latent --
            # the reference draws inside the prior, after the replication -- so it is not replicated with the other keywords
            eps = kwargs.pop("eps", None) if self.expansion and self.expansion > 1 else None
            encodings, kwargs = self._expand(encodings), self._expand(kwargs)
            if eps is not None:
                if eps.shape[0] != encodings.shape[0]:
                    raise ValueError(f"`eps` must hold one draw per replicated latent: {encodings.shape[0]} = expansion * batch, "
                                     f"got {eps.shape[0]}")
                kwargs["eps"] = eps
        if self.prior is None:
            results = encodings, torch.zeros(encodings.size(0), device=encodings.device, dtype=encodings.dtype), {}
        else:
            with self._filter(self.prior, kwargs.keys()) as prior:
                results = prior(encodings, **kwargs, step=self.global_step)
        return results if return_prior_artifacts else results[0]

    @VisionModule.postprocess
    def decode(self, latents: Tensor, expand_kwargs: bool = False, **kwargs) -> Tensor:
        if expand_kwargs:
            kwargs = self._expand({k: v for k, v in kwargs.items() if k != "eps"})   # (the prior's noise is no decoder input)
        with self._filter(self._decode_func, kwargs.keys()) as decode:
            return decode(latents, **kwargs)

    @VisionModule.postprocess
    def sample(self, batch_size: int, **kwargs) -> Tensor:
        if self.prior is not None:
            with self._filter(self.prior.sample, kwargs.keys()) as sample:
                latents = sample((batch_size, *self.latent_size), device=self.device, **kwargs)
        else:
            latents = torch.randn((batch_size, *self.latent_size), device=self.device)
        return self.decode(latents, **kwargs, no_postprocess_override=True)
