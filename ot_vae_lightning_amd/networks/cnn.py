"""Modular CNN with the reference's plug-in API (``ConvLayer, Conv1x1, AttentionBlock, ConvBlock, CNN, AutoEncoder``;
reference networks/cnn.py) whose arithmetic runs on the gfx950 kernels behind ``include/otvae.h``.

Same class names, constructor signatures, ``state_dict`` keys and initialisation draws as the reference, so its
checkpoints load and ``VAE(encoder=CNN(...), decoder=CNN(...), prior=...)`` is unchanged for the user.  What differs
is execution: a ConvLayer is ONE fused kernel (BatchNorm-apply + ReLU + nearest up-sampling + convolution + bias
[+ residual add]) fed by a statistics pass shared between a block's first layer and its skip branch, activations
stay channels-last between layers, and attention never materialises the TxT matrix.

Activations other than ReLU (LeakyReLU(0.2), SELU, GELU, SiLU) and ``equalized_lr`` -- what the reference's
configs/vae/defaults_imagenet.yaml trains with -- run unfused around the same kernels (functional._conv_layer_general,
csrc/activation.hip), and so do GroupNorm / InstanceNorm2d (csrc/groupnorm.hip), FiLM conditioning (`additional_embed`) and
Dropout2d (csrc/film_dropout2d.hip).  Grouped and dilated layers keep nn.Conv2d's parameter shape and expand it per call into the
dense weight the kernels take (csrc/weight_expand.hip: zeros between groups and in the holes of the dilation; up to 7 x 7 taps).
A module-valued ``up_sample`` / ``down_sample`` (the user's own plug-in) is called as given, around the layer's kernels.
"""
import math
import warnings
from math import log2, sqrt
from typing import List, Optional, Union

import torch
import torch.nn as nn
from torch import Tensor

from .. import functional as HF
from .nets_utils import FilterSequential, QKVAttention

__all__ = ["ConvLayer", "Conv1x1", "ConvBlock", "AttentionBlock", "CNN", "AutoEncoder"]


def _is_none(s: Optional[str]) -> bool:
    return s is None or "none" in s.lower() or "null" in s.lower()


class ConvLayer(nn.Module):
    """BatchNorm -> ReLU -> (nearest x2 up-sampling) -> convolution (4x4 stride 2 when down-sampling), pre-activation
    order as in the reference (networks/cnn.py:183-192).  ``weight`` has the logical shape [out, in, kh, kw]."""

    enable_warnings = False

    def __init__(self, in_features: int, out_features: int,
                 down_sample: Union[bool, int, nn.Module] = False, up_sample: Union[bool, int, nn.Module] = False,
                 additional_embed: Optional[int] = None, normalization: Optional[str] = None,
                 activation: Optional[str] = None, equalized_lr: Optional[float] = None, dropout: float = 0.,
                 kernel_size=3, stride=1, padding=1, dilation=1, groups: int = 1, bias: bool = True) -> None:
        super().__init__()
        # a user-supplied resampling MODULE is the user's plug-in: it is called as given (cnn.py:97,106), around the layer's own
        # kernels (functional._conv_layer_general); only the reference's built-in nearest x2 / strided variants are fused
        down_module = down_sample if isinstance(down_sample, nn.Module) else None
        up_module = up_sample if isinstance(up_sample, nn.Module) else None
        if down_module is not None:
            down_sample = False
        if up_module is not None:
            up_sample = False
        kernel_size = kernel_size if isinstance(kernel_size, int) else kernel_size[0]
        stride = stride if isinstance(stride, int) else stride[0]
        padding = padding if isinstance(padding, int) else padding[0]
        dilation = dilation if isinstance(dilation, int) else dilation[0]
        if bool(down_sample):  # reference cnn.py:98-101
            kernel_size = max(2 * int(down_sample), kernel_size)
            stride = 2 if isinstance(down_sample, bool) else int(down_sample)
            padding = (kernel_size - 1) // 2
        groups = groups if in_features % groups == 0 and out_features % groups == 0 else 1
        # grouped / dilated layers keep nn.Conv2d's parameter shape [out, in / groups, k, k]; the forward pass expands it into the
        # dense weight the kernels take (functional._WeightExpandFn), whose taps may span 7 x 7 at most
        self.groups, self.dilation = int(groups), (int(dilation), int(dilation))
        self._expand = (int(groups), int(dilation)) if (groups != 1 or dilation != 1) else None
        # The tuned kernels take strides 1 / 2 and footprints up to 7 x 7 taps.  Anything else the reference's constructor can make --
        # `down_sample = s >= 4` is a (2 s) x (2 s) kernel with stride s (cnn.py:98-101; CNN(scaling_factor=4): 8 x 8, stride 4) -- runs
        # on the direct-convolution fallback (csrc/conv_generic.hip) with the normalisation / activation unfused around it
        footprint = (kernel_size - 1) * dilation + 1
        self._generic = footprint > 7 or stride not in (1, 2)
        if footprint > 32:
            raise NotImplementedError(f"a {kernel_size} x {kernel_size} kernel with dilation {dilation} spans more than the 32 x 32 taps "
                                      "of the MI355X convolution kernels")
        if stride < 1:
            raise ValueError(f"stride must be positive, got {stride}")
        up = 1
        if isinstance(up_sample, bool):
            up = 2 if up_sample else 1
        elif isinstance(up_sample, int) and up_sample > 0:
            up = int(up_sample)
        if up not in (1, 2) or (up == 2 and self._generic):
            # the kernels fuse the nearest x2 up-sampling only (and the direct-convolution fallback none); any other factor runs as the
            # reference's own nn.Upsample module around them (the same route as a user-supplied module)
            up_module, up = nn.Upsample(scale_factor=up), 1
        self.in_channels, self.out_channels = in_features, out_features
        self.kernel_size, self.stride, self.padding = (kernel_size, kernel_size), (stride, stride), (padding, padding)
        self._up = up

        # parameters: same creation order and the same RNG draws as nn.Conv2d.reset_parameters (reference: ConvLayer
        # subclasses nn.Conv2d), written into HWIO memory
        w0 = torch.empty(out_features, in_features // groups, kernel_size, kernel_size)
        nn.init.kaiming_uniform_(w0, a=math.sqrt(5))
        if self._expand is None:
            self.weight = nn.Parameter(HF.new_hwio(out_features, in_features, kernel_size, kernel_size))
        else:
            self.weight = nn.Parameter(torch.empty_like(w0))
        if bias:
            fan_in = (in_features // groups) * kernel_size * kernel_size
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            self.bias = nn.Parameter(torch.empty(out_features).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._down_sample = down_module if down_module is not None else nn.Identity()
        self._up_sample = up_module if up_module is not None else (nn.Upsample(scale_factor=up) if up > 1 else nn.Identity())
        self._resample_modules = (up_module is not None, down_module is not None)
        self._dropout = nn.Dropout2d(dropout) if dropout and dropout > 0 else nn.Identity()   # cnn.py:112 (container: p)
        # FiLM conditioning (cnn.py:114-116): two Linear projections of the (activated) embedding, created in the reference's order
        self._embed_proj_scale = nn.Linear(additional_embed, in_features) if bool(additional_embed) else None
        self._embed_proj_bias = nn.Linear(additional_embed, in_features) if bool(additional_embed) else None

        if _is_none(normalization):
            self._normalization = nn.Identity()
        elif "batch" in normalization.lower():
            self._normalization = nn.BatchNorm2d(in_features)  # parameter/buffer container; its forward is never called
        elif "group" in normalization.lower():     # cnn.py:123
            self._normalization = nn.GroupNorm(div_sqrt(in_features // groups), in_features)
        elif "instance" in normalization.lower():  # cnn.py:124: no affine parameters, no running statistics
            self._normalization = nn.InstanceNorm2d(in_features)
        else:
            raise NotImplementedError(f"normalization={normalization} not supported")

        # activation + its weight initialisation, in the reference's order of tests (cnn.py:128-147)
        self._act_kind = 0
        if _is_none(activation):
            self._activation = nn.Identity()
        elif "leaky" in activation.lower():
            self._activation, self._act_kind = nn.LeakyReLU(0.2), HF.ACT_KINDS["leaky"]
            nn.init.kaiming_uniform_(w0, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        elif "relu" in activation.lower():
            self._activation, self._act_kind = nn.ReLU(), HF.ACT_KINDS["relu"]
            nn.init.kaiming_uniform_(w0, mode="fan_out", nonlinearity="relu")
        elif "selu" in activation.lower():
            self._activation, self._act_kind = nn.SELU(), HF.ACT_KINDS["selu"]
            nn.init.xavier_uniform_(w0, gain=nn.init.calculate_gain("selu"))
        elif "gelu" in activation.lower():
            self._activation, self._act_kind = nn.GELU(), HF.ACT_KINDS["gelu"]
            nn.init.xavier_uniform_(w0, gain=nn.init.calculate_gain("linear"))
        elif "silu" in activation.lower() or "swish" in activation.lower():
            self._activation, self._act_kind = nn.SiLU(), HF.ACT_KINDS["silu"]
            nn.init.xavier_uniform_(w0, gain=nn.init.calculate_gain("linear"))
        else:
            raise NotImplementedError(f"activation={activation} not supported")
        # equalized learning rate (cnn.py:114-118,149-158,186-188): N(0, 1/lr_mult) weights, the forward pass multiplies the
        # weight by gain / sqrt(fan_in) * lr_mult and the bias by lr_mult
        self._lr_mult, self._gain = equalized_lr or 1, 1
        self._conv_scale = self._gain / math.sqrt((in_features // groups) * kernel_size * kernel_size) if equalized_lr else 1
        self._linear_scale = self._gain / math.sqrt(in_features) if equalized_lr else 1
        if equalized_lr:
            nn.init.normal_(w0, std=1 / self._lr_mult)
            if self._embed_proj_scale is not None:   # cnn.py:152-157
                nn.init.normal_(self._embed_proj_scale.weight, std=1 / self._lr_mult)
                nn.init.normal_(self._embed_proj_bias.weight, std=1 / self._lr_mult)
        with torch.no_grad():
            self.weight.copy_(w0)

    # -- helpers ----------------------------------------------------------------------------------------------
    @property
    def _has_norm(self) -> bool:
        return isinstance(self._normalization, nn.BatchNorm2d)

    def _film(self, embed: Optional[Tensor]):
        """(scale, bias) [N, C] of the FiLM conditioning from ``embed`` [N, E] (cnn.py:160-181), or None"""
        if self._embed_proj_scale is None:
            if embed is not None and self.enable_warnings:
                warnings.warn("given conditional argument `embed` but the layer has no embedding projection")
            return None
        if embed is None:
            raise ValueError("`additional_embed` specified in the ConvLayer constructor but `embed` is None")
        from ..ot.matrix_utils import mm
        e = embed.float()
        if self._act_kind:  # the layer's own activation on the embedding (cnn.py:170,175)
            e = HF._BnActFn.apply(HF.as_nhwc(e[:, :, None, None]), None, None, None, self._act_kind, (None, None))[:, :, 0, 0]
        c = float(self._linear_scale * self._lr_mult)
        lr = float(self._lr_mult)
        out = []
        for proj in (self._embed_proj_scale, self._embed_proj_bias):
            w, b = proj.weight, proj.bias
            if c != 1.0:
                w = HF._ScaleFn.apply(w, c)
            if lr != 1.0:
                b = HF._ScaleFn.apply(b, lr)
            out.append(mm(e.contiguous(), w.t().contiguous()) + b)
        return tuple(out)

    def _dropout2d(self):
        """(p, device key {seed, counter}, stream id) of the layer's Dropout2d for this training-mode forward, or None"""
        if not (self.training and isinstance(self._dropout, nn.Dropout2d)):
            return None
        key = self.__dict__.get("_dropout_key")
        if key is None or key.device != self.weight.device:
            key = self.__dict__["_dropout_key"] = HF.new_dropout_key(self.weight.device)
        key[1:].add_(1)   # device-side: a captured step draws a fresh mask on every replay
        return float(self._dropout.p), key, 0

    def branch(self, residual: Optional[Tensor] = None, out_stats: bool = True, embed: Optional[Tensor] = None) -> dict:
        bn = self._normalization if self._has_norm else None
        gn = None
        if isinstance(self._normalization, nn.GroupNorm):
            gn = (self._normalization.num_groups, self._normalization.weight, self._normalization.bias)
        elif isinstance(self._normalization, nn.InstanceNorm2d):
            gn = (self.in_channels, None, None)
        return dict(group_norm=gn, film=self._film(embed), dropout2d=self._dropout2d(), weight=self.weight, bias=self.bias,
                    expand=self._expand,
                    up_module=self._up_sample if self._resample_modules[0] else None,
                    down_module=self._down_sample if self._resample_modules[1] else None,
                    gamma=bn.weight if bn is not None else None, beta=bn.bias if bn is not None else None,
                    running_mean=bn.running_mean if bn is not None else None,
                    running_var=bn.running_var if bn is not None else None,
                    num_batches_tracked=bn.num_batches_tracked if bn is not None else None,
                    residual=residual, stride=self.stride[0], pad=self.padding[0], up=self._up,
                    relu=isinstance(self._activation, nn.ReLU), act=self._act_kind, generic=self._generic,
                    wscale=float(self._conv_scale * self._lr_mult), bscale=float(self._lr_mult), out_stats=out_stats)

    def forward(self, x: Tensor, embed: Optional[Tensor] = None, *, residual: Optional[Tensor] = None,
                out_stats: bool = True) -> Tensor:
        """``out_stats``: let the kernel's epilogue also emit the per-channel sums of its output, which the next layer's
        BatchNorm picks up instead of re-reading the tensor (wasted only if the consumer has no BatchNorm)."""
        return HF.conv_layers(x, [self.branch(residual, out_stats, embed)], training=self.training)[0]

    def extra_repr(self) -> str:
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, up={self._up}, bias={self.bias is not None}")

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # nothing special: copy_() maps the checkpoint's OIHW values onto the HWIO memory
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class Conv1x1(ConvLayer):
    """Bias-free, activation-free 1x1 ConvLayer (4x4 stride 2 when down-sampling), reference cnn.py:195-206."""

    def __init__(self, in_features: int, out_features: int, **kwargs):
        defaults = dict(down_sample=False, up_sample=False, additional_embed=None, normalization=None, activation=None,
                        equalized_lr=False, dropout=0., stride=1, kernel_size=1, padding=0, dilation=1, groups=1,
                        bias=False)
        super().__init__(in_features, out_features, **{**defaults, **kwargs})


class AttentionBlock(nn.Module):
    """proj_out(attention(qkv(BN(x)))) -- not residual (reference cnn.py:212-240)."""

    def __init__(self, channels: int, heads: int = 1, additional_embed: Optional[int] = None,
                 normalization: Optional[str] = None, equalized_lr: Optional[float] = None, groups: int = 1):
        super().__init__()
        if channels % heads != 0:
            raise ValueError(f"q,k,v channels: {channels} is not divisible by heads: {heads}")
        self.qkv = Conv1x1(channels, channels * 3, additional_embed=additional_embed, normalization=normalization,
                           equalized_lr=equalized_lr, groups=groups)
        self.attention = QKVAttention(heads)
        self.proj_out = Conv1x1(channels, channels, equalized_lr=equalized_lr, groups=groups)

    def forward(self, x: Tensor, embed: Optional[Tensor] = None, *, residual: Optional[Tensor] = None) -> Tensor:
        qb, pb = self.qkv.branch(None, False, embed), self.proj_out.branch(residual, True)
        y = HF.attention_stage(x, qb, self.attention.n_heads, pb, training=self.qkv.training)   # one launch where the shapes allow
        if y is not None:
            return y
        qkv = HF.conv_layers(x, [qb], training=self.qkv.training)[0]          # [N, 3*C, H, W] channels-last; feeds attention, no BN
        h = HF.qkv_attention(qkv, self.attention.n_heads)                     # [N, C, H, W]
        return HF.conv_layers(h, [pb], training=self.proj_out.training)[0]


class ConvBlock(nn.Module):
    """[ConvLayer(down/up), ConvLayer x (n_layers-1), AttentionBlock|Identity] (+ skip), reference cnn.py:246-335."""

    def __init__(self, in_features: int, out_features: int, n_attn_heads: int = 0, n_layers: int = 2,
                 down_sample=False, up_sample=False, additional_embed: Optional[int] = None,
                 normalization: Optional[str] = "batchnorm", activation: Optional[str] = "relu",
                 residual: Optional[str] = None, equalized_lr: Optional[float] = None, dropout: float = 0.,
                 kernel_size=3, stride=1, padding=1, dilation=1, groups: int = 1, bias: bool = True) -> None:
        super().__init__()
        self.out_features = out_features
        embed_features = out_features // 2 if residual == "cat" else out_features
        self.block = FilterSequential(
            ConvLayer(in_features, embed_features, down_sample, up_sample, additional_embed, normalization, activation,
                      equalized_lr, dropout, kernel_size, stride, padding, dilation, groups, bias),
            *[ConvLayer(embed_features, embed_features, False, False, additional_embed, normalization, activation,
                        equalized_lr, dropout, kernel_size, stride, padding, dilation, groups, bias)
              for _ in range(n_layers - 1)],
            AttentionBlock(embed_features, n_attn_heads, additional_embed, normalization, equalized_lr, groups)
            if n_attn_heads > 0 else nn.Identity(),
        )
        self.residual = residual
        self.skip = Conv1x1(in_features, embed_features, down_sample=down_sample, up_sample=up_sample,
                            normalization=normalization, equalized_lr=equalized_lr, groups=groups) \
            if residual in ["cat", "add"] else None

    def forward(self, x: Tensor, embed: Optional[Tensor] = None) -> Tensor:
        layers = [l for l in self.block if not isinstance(l, nn.Identity)]
        first = layers[0]
        sk = None
        if self.skip is not None:
            # block[0] and skip normalise the same tensor: one statistics pass, one fused backward
            # (the skip output is only ever added to the block output: nobody normalises it -> no statistics)
            out, sk = HF.conv_layers(x, [first.branch(embed=embed), self.skip.branch(out_stats=False)], training=self.training)
        else:
            out = first(x, embed)
        # the skip is added inside the last layer's convolution epilogue -- unless something sits between that convolution and the
        # sum in the reference's order `dropout(down(conv(...))) + skip` (cnn.py:183-192,331-335): an active Dropout2d or a down-sampling module
        last_layer = layers[-1]
        plain_tail = not (isinstance(last_layer, ConvLayer) and ((last_layer.training and isinstance(last_layer._dropout, nn.Dropout2d))
                                                                  or last_layer._resample_modules[1]))
        fuse_add = self.residual == "add" and len(layers) > 1 and plain_tail
        for i, layer in enumerate(layers[1:], start=1):
            last = i == len(layers) - 1
            out = layer(out, embed, residual=sk if (fuse_add and last) else None)
        if self.residual == "add" and not fuse_add:
            out = out + sk
        elif self.residual == "cat":
            out = torch.cat([out, sk], dim=1)
        return out


class CNN(FilterSequential):
    """Stack of ConvBlocks whose widths/resolutions/heads are inferred exactly like the reference
    (networks/cnn.py:341-458)."""

    def __init__(self, in_features: int, out_features: int, in_resolution: Optional[int] = None,
                 out_resolution: Optional[int] = None, intermediate_features: Optional[List[int]] = None,
                 capacity: int = 8, max_attn_res: int = 16, n_layers: int = 2, residual: Optional[str] = None,
                 down_sample=False, up_sample=False, additional_embed: Optional[int] = None,
                 normalization: Optional[str] = "batchnorm", activation: Optional[str] = "relu",
                 equalized_lr: Optional[float] = None, dropout: float = 0., kernel_size=3, stride=1, padding=1,
                 dilation=1, groups: int = 1, bias: bool = True) -> None:
        if bool(up_sample) and bool(down_sample):
            raise ValueError("Both `up_sample` and `down_sample` are set.")
        if intermediate_features is not None:
            features = [in_features] + intermediate_features + [out_features]
            attn_resolutions = [max_attn_res] * len(features)
        else:
            if not all([in_resolution is not None, out_resolution is not None, bool(up_sample) or bool(down_sample)]):
                raise ValueError("`features` is None. Set `in_resolution`, `out_resolution` and"
                                 " (`up_sample` or `down_sample`)  to infer number of blocks")
            if bool(down_sample):
                if in_resolution <= out_resolution:
                    raise ValueError("`down_sample` set but `in_resolution` < `out_resolution`")
                if isinstance(down_sample, bool):
                    down_sample = 2
                features, resolutions = get_channel_list(in_features, out_features, in_resolution, out_resolution,
                                                         down_sample, capacity)
                attn_resolutions = resolutions[1:]
            else:
                if out_resolution <= in_resolution:
                    raise ValueError("`up_sample` set but `out_resolution` < `in_resolution`")
                if isinstance(up_sample, bool):
                    up_sample = 2
                features, resolutions = get_channel_list(out_features, in_features, out_resolution, in_resolution,
                                                         up_sample, capacity)
                features, resolutions = features[::-1], resolutions[::-1]
                attn_resolutions = resolutions[:-1]
        heads = lambda ch, res: div_sqrt(ch) if res <= max_attn_res else 0  # noqa: E731
        super().__init__(*[
            ConvBlock(ic, oc, heads(oc, r), n_layers, down_sample, up_sample, additional_embed, normalization,
                      activation, residual, equalized_lr, dropout, kernel_size, stride, padding, dilation, groups, bias)
            for ic, oc, r in zip(features[:-1], features[1:], attn_resolutions)])
        self.out_size = torch.Size([out_features, out_resolution, out_resolution])

    def forward(self, x: Tensor, embed: Optional[Tensor] = None) -> Tensor:
        return super().forward(x, embed=embed)


class AutoEncoder(nn.Module):
    """encoder CNN + mirrored decoder CNN with ``encode``/``decode``/``latent_size`` (reference cnn.py:463-600).
    Class/time conditioning feeds FiLM embeddings, which the MI355X ConvLayer does not implement."""

    def __init__(self, in_features: int, latent_features: int, in_resolution: Optional[int] = None,
                 latent_resolution: Optional[int] = None, intermediate_features: Optional[List[int]] = None,
                 capacity: int = 8, max_attn_res: int = 16, num_classes: Optional[int] = None,
                 time_embed_dim: Optional[int] = None, double_encoded_features: bool = False, n_layers: int = 2,
                 residual: Optional[str] = None, down_up_sample: Union[bool, int] = False,
                 normalization: Optional[str] = "batchnorm", activation: Optional[str] = "relu",
                 equalized_lr: Optional[float] = None, dropout: float = 0., kernel_size=3, stride=1, padding=1,
                 dilation=1, groups: int = 1, bias: bool = True) -> None:
        super().__init__()
        if bool(num_classes) or bool(time_embed_dim):
            raise NotImplementedError("class / time conditioned AutoEncoder is not supported on the MI355X path")
        enc_out = latent_features * (1 + int(double_encoded_features))
        self.latent_size = torch.Size([enc_out, latent_resolution, latent_resolution])
        self.class_embed = None
        self.time_embed = None
        self.encoder = CNN(in_features, enc_out, in_resolution, latent_resolution, intermediate_features, capacity,
                           max_attn_res, n_layers, residual, down_up_sample, False, None, normalization, activation,
                           equalized_lr, dropout, kernel_size, stride, padding, dilation, groups, bias)
        self.decoder = CNN(latent_features, in_features, latent_resolution, in_resolution,
                           intermediate_features[::-1] if intermediate_features is not None else None, capacity,
                           max_attn_res, n_layers, residual, False, down_up_sample, None, normalization, activation,
                           equalized_lr, dropout, kernel_size, stride, padding, dilation, groups, bias)

    def embed(self, labels: Optional[Tensor] = None, time: Optional[Tensor] = None):
        if labels is not None:
            warnings.warn("given conditional argument `labels` but `self.class_embed` is None.")
        if time is not None:
            warnings.warn("given conditional argument `time` but `self.time_embed` is None.")
        return None

    def encode(self, x: Tensor, labels: Optional[Tensor] = None, time: Optional[Tensor] = None) -> Tensor:
        return self.encoder(x, self.embed(labels, time))

    def decode(self, z: Tensor, labels: Optional[Tensor] = None, time: Optional[Tensor] = None) -> Tensor:
        return self.decoder(z, self.embed(labels, time))

    def forward(self, x: Tensor, labels: Optional[Tensor] = None, time: Optional[Tensor] = None) -> Tensor:
        return self.decode(self.encode(x, labels, time), labels, time)


# ---- architecture inference (reference cnn.py:605-672) -----------------------------------------------------------
def get_block_scaling(max_resolution: int, min_resolution: int, max_scaling: int) -> List[int]:
    """e.g. (64, 2, 4) -> [4, 4, 2]: greedy factorisation of the resolution ratio into powers of two <= max_scaling."""
    remaining = int(log2(max_resolution // min_resolution))
    exp = int(log2(max_scaling))
    factors: List[int] = []
    while remaining > 0:
        factors += [2 ** exp] * (remaining // exp)
        remaining %= exp
        exp -= 1
    return factors


def get_channel_list(in_features: int, out_features: int, in_resolution: int, out_resolution: int,
                     scaling_factor: int, capacity: int):
    """Widths double from ``capacity`` (clamped to [in_features, out_features]); last width is ``out_features``."""
    scalings = get_block_scaling(in_resolution, out_resolution, scaling_factor)
    features = [max(min(capacity << i, out_features), in_features) for i in range(len(scalings))]
    resolutions = [in_resolution]
    for sf in scalings:
        resolutions.append(resolutions[-1] // sf)
    features[-1] = out_features
    return [in_features] + features, resolutions


def div_sqrt(n: int) -> int:
    """The divisor of ``n`` the reference picks as head count: the smallest divisor >= sqrt(n)
    (np.searchsorted over sympy.divisors, cnn.py:660-672)."""
    assert isinstance(n, int) and n > 0, f"Error, n must be a positive integer. Given n={n}."
    root = sqrt(n)
    for d in range(1, n + 1):
        if n % d == 0 and d >= root:
            return d
    return n
