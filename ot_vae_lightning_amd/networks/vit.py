"""``ViT`` / ``PositionalEmbedding`` with the reference's constructor, token bookkeeping and ``state_dict`` keys
(networks/vit.py:33-248), running on the MI355X kernels: every Linear (patch embedding, attention projections,
feed-forward pairs, patch read-out) goes through the 1x1 convolution kernels (bias, the feed-forward ReLU fused as the
next layer's input activation), every LayerNorm through ``otvae_layernorm_*`` with the block's residual sum folded in,
and the attention through the fused QKV kernels with the 1/sqrt(head width) scale of ``nn.MultiheadAttention``.

The transformer is the reference's ``nn.TransformerEncoder`` of post-norm ``nn.TransformerEncoderLayer``s (ReLU,
``batch_first``): x = norm1(x + out_proj(attn(in_proj(x)))); x = norm2(x + linear2(relu(linear1(x)))).  Parameters are
drawn by constructing the very torch modules the reference constructs, in its order, so a seeded construction gives the
reference's initial weights (note that ``nn.TransformerEncoder`` deep-copies ONE layer: all layers start identical).

``preprocess_depth`` (networks/vit.py:171-181,240-244): the output tokens become the target of an ``nn.TransformerDecoder``
(post-norm layers: self-attention, cross-attention over the memory, feed-forward) whose memory is the remaining tokens, run
through ``preprocess_depth`` encoder layers first; the cross-attention kernel (``otvae_attn_cross_*``) reads queries and
keys / values of different token counts.  Not implemented (raises): ``time_dependant`` (the reference's own time token cannot
be constructed).  ``causal_mask`` runs on the generic attention kernels of ``attention_dropout.hip``
(softmax over tokens <= t).
Dropout > 0 in training mode: the three dropouts of a layer that act on token tensors are applied with
``torch.nn.functional.dropout`` between the kernels; the fourth, nn.MultiheadAttention's dropout on the attention
probabilities, happens inside the attention kernel (``otvae_attn_dropout_*``: the mask is a hash recomputed by the
backward pass, never stored).  The random streams differ from the CPU reference's, so value parity is tested with
dropout 0 / eval mode, and the dropout path against a reference that is handed the kernel's own mask."""
import warnings
from typing import Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import functional as HF

__all__ = ["PositionalEmbedding", "ViT", "AutoRegressive", "TokenLinear", "TokenLayerNorm", "TokenEncoderLayer", "TokenDecoderLayer"]


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def _adopt(weight: Tensor) -> nn.Parameter:
    """a torch-initialised [out, in] Linear weight moved onto [in][out] memory (what the 1x1 kernels read directly)"""
    w = HF.new_linear_weight(weight.shape[0], weight.shape[1])
    with torch.no_grad():
        w.copy_(weight)
    p = nn.Parameter(w)
    p._otvae_linear = True  # engine.HipTrainer keeps a resident transposed copy for the data-gradient kernel
    return p


class TokenLinear(nn.Module):
    """``nn.Linear`` on [N, T, D_in] tokens (keys ``weight``, ``bias``)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, like: Optional[nn.Linear] = None):
        super().__init__()
        like = like if like is not None else nn.Linear(in_features, out_features, bias=bias)
        self.in_features, self.out_features = in_features, out_features
        self.weight = _adopt(like.weight.data)
        self.bias = nn.Parameter(like.bias.data.clone()) if like.bias is not None else None

    def forward(self, x: Tensor, relu_input: bool = False) -> Tensor:
        return HF.linear_tokens(x, self.weight, self.bias, relu_input=relu_input)


class TokenLayerNorm(nn.Module):
    """``nn.LayerNorm(dim)`` (keys ``weight``, ``bias``) with an optional residual summed in before normalising."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim)), eps

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, dropout_p: float = 0.0, dropout_key: Optional[Tensor] = None,
                stream_id: int = 0) -> Tensor:
        return HF.layer_norm_tokens(x, self.weight, self.bias, self.eps, residual, dropout_p, dropout_key, stream_id)


class _SelfAttention(nn.Module):
    """parameter holder with ``nn.MultiheadAttention``'s names: in_proj_weight / in_proj_bias / out_proj.{weight,bias}"""

    def __init__(self, like: nn.MultiheadAttention):
        super().__init__()
        self.embed_dim, self.num_heads = like.embed_dim, like.num_heads
        self.in_proj_weight = _adopt(like.in_proj_weight.data)
        self.in_proj_bias = nn.Parameter(like.in_proj_bias.data.clone())
        self.out_proj = TokenLinear(like.embed_dim, like.embed_dim, like=like.out_proj)

    def forward(self, x: Tensor, dropout_p: float = 0.0, dropout_key: Optional[Tensor] = None, stream_id: int = 0,
                causal: bool = False) -> Tensor:
        qkv = HF.linear_tokens(x, self.in_proj_weight, self.in_proj_bias)
        return self.out_proj(HF.mha_attention_tokens(qkv, self.num_heads, dropout_p, dropout_key, stream_id, causal=causal))


class TokenEncoderLayer(nn.Module):
    """post-norm ``nn.TransformerEncoderLayer(dim, heads, mlp_dim, dropout, batch_first=True)`` (ReLU feed-forward)"""

    def __init__(self, like: nn.TransformerEncoderLayer):
        super().__init__()
        if getattr(like, "norm_first", False):
            raise NotImplementedError("norm_first transformer layers are not implemented on the MI355X path")
        self.self_attn = _SelfAttention(like.self_attn)
        self.linear1 = TokenLinear(like.linear1.in_features, like.linear1.out_features, like=like.linear1)
        self.linear2 = TokenLinear(like.linear2.in_features, like.linear2.out_features, like=like.linear2)
        self.norm1 = TokenLayerNorm(like.norm1.normalized_shape[0], like.norm1.eps)
        self.norm2 = TokenLayerNorm(like.norm2.normalized_shape[0], like.norm2.eps)
        self.p = float(like.dropout.p)

    def _drop(self, x: Tensor) -> Tensor:
        return F.dropout(x, self.p, True) if (self.training and self.p > 0) else x

    def forward(self, x: Tensor, dropout_key: Optional[Tensor] = None, stream_id: int = 0, causal: bool = False) -> Tensor:
        """``dropout_key`` (training with p > 0): the attention probabilities are thinned inside the attention kernel (what
        nn.MultiheadAttention does with the layer's p) and the two "x + dropout(sublayer(x))" inside the LayerNorm kernels;
        the dropout between the ReLU and linear2 is one kernel with the ReLU.  Call sites of one forward pass are told apart
        by stream ids: layer i uses i (attention), 1024 + 2 i and 1025 + 2 i (the two norms), 2048 + i (feed-forward)."""
        fused = self.training and self.p > 0 and dropout_key is not None
        p = self.p if fused else 0.0
        a = self.self_attn(x, p, dropout_key, stream_id, causal)
        x = self.norm1(a, x, p, dropout_key, 1024 + 2 * stream_id) if fused else self.norm1(self._drop(a), residual=x)
        h = self.linear1(x)
        if fused and h.shape[-1] % 4 == 0:  # dropout sits between the ReLU and linear2: one kernel for the pair
            f = self.linear2(HF.dropout_tokens(h, p, dropout_key, 2048 + stream_id, relu=True))
        elif self.training and self.p > 0:
            f = self.linear2(self._drop(torch.relu(h)))
        else:
            f = self.linear2(h, relu_input=True)
        return self.norm2(f, x, p, dropout_key, 1025 + 2 * stream_id) if fused else self.norm2(self._drop(f), residual=x)


class _Encoder(nn.Module):
    """``nn.TransformerEncoder``: keys ``layers.{i}.*``"""

    def __init__(self, like: nn.TransformerEncoder):
        super().__init__()
        self.layers = nn.ModuleList([TokenEncoderLayer(layer) for layer in like.layers])
        if like.norm is not None:
            raise NotImplementedError("a final norm on the TransformerEncoder is not part of the reference's ViT")

    def forward(self, x: Tensor, dropout_key: Optional[Tensor] = None, causal: bool = False) -> Tensor:
        for i, layer in enumerate(self.layers):
            x = layer(x, dropout_key, i, causal)
        return x


class _CrossAttention(nn.Module):
    """``nn.MultiheadAttention(x, memory, memory)`` with its parameter names; q from x, k | v from the memory tokens"""

    def __init__(self, like: nn.MultiheadAttention):
        super().__init__()
        self.embed_dim, self.num_heads = like.embed_dim, like.num_heads
        self.in_proj_weight = _adopt(like.in_proj_weight.data)
        self.in_proj_bias = nn.Parameter(like.in_proj_bias.data.clone())
        self.out_proj = TokenLinear(like.embed_dim, like.embed_dim, like=like.out_proj)

    def forward(self, x: Tensor, memory: Tensor, dropout_p: float = 0.0, dropout_key: Optional[Tensor] = None,
                stream_id: int = 0) -> Tensor:
        return self.out_proj(HF.cross_attention_tokens(x, memory, self.in_proj_weight, self.in_proj_bias, self.num_heads,
                                                       dropout_p, dropout_key, stream_id))


class TokenDecoderLayer(nn.Module):
    """post-norm ``nn.TransformerDecoderLayer(dim, heads, mlp_dim, dropout, batch_first=True)``:
    x = norm1(x + self_attn(x)); x = norm2(x + multihead_attn(x, memory, memory)); x = norm3(x + linear2(relu(linear1(x))))"""

    def __init__(self, like: nn.TransformerDecoderLayer):
        super().__init__()
        if getattr(like, "norm_first", False):
            raise NotImplementedError("norm_first transformer layers are not implemented on the MI355X path")
        self.self_attn = _SelfAttention(like.self_attn)
        self.multihead_attn = _CrossAttention(like.multihead_attn)
        self.linear1 = TokenLinear(like.linear1.in_features, like.linear1.out_features, like=like.linear1)
        self.linear2 = TokenLinear(like.linear2.in_features, like.linear2.out_features, like=like.linear2)
        self.norm1 = TokenLayerNorm(like.norm1.normalized_shape[0], like.norm1.eps)
        self.norm2 = TokenLayerNorm(like.norm2.normalized_shape[0], like.norm2.eps)
        self.norm3 = TokenLayerNorm(like.norm3.normalized_shape[0], like.norm3.eps)
        self.p = float(like.dropout.p)

    def _drop(self, x: Tensor) -> Tensor:
        return F.dropout(x, self.p, True) if (self.training and self.p > 0) else x

    def forward(self, x: Tensor, memory: Tensor, dropout_key: Optional[Tensor] = None, index: int = 0, causal: bool = False) -> Tensor:
        """Dropout call sites of layer i under one key (the encoder layers of ``prepocess`` own 0.., 1024.., 2048..): 256 + i
        self-attention, 512 + i cross-attention, 3072 + 3 i + {0, 1, 2} the three norms, 2304 + i the feed-forward pair."""
        fused = self.training and self.p > 0 and dropout_key is not None
        p, i = (self.p if fused else 0.0), index
        a = self.self_attn(x, p, dropout_key, 256 + i, causal)
        x = self.norm1(a, x, p, dropout_key, 3072 + 3 * i) if fused else self.norm1(self._drop(a), residual=x)
        c = self.multihead_attn(x, memory, p, dropout_key, 512 + i)
        x = self.norm2(c, x, p, dropout_key, 3073 + 3 * i) if fused else self.norm2(self._drop(c), residual=x)
        h = self.linear1(x)
        if fused and h.shape[-1] % 4 == 0:
            f = self.linear2(HF.dropout_tokens(h, p, dropout_key, 2304 + i, relu=True))
        elif self.training and self.p > 0:
            f = self.linear2(self._drop(torch.relu(h)))
        else:
            f = self.linear2(h, relu_input=True)
        return self.norm3(f, x, p, dropout_key, 3074 + 3 * i) if fused else self.norm3(self._drop(f), residual=x)


class _Decoder(nn.Module):
    """``nn.TransformerDecoder``: keys ``layers.{i}.*``"""

    def __init__(self, like: nn.TransformerDecoder):
        super().__init__()
        self.layers = nn.ModuleList([TokenDecoderLayer(layer) for layer in like.layers])
        if like.norm is not None:
            raise NotImplementedError("a final norm on the TransformerDecoder is not part of the reference's ViT")
        if len(self.layers) > 256:
            raise NotImplementedError("more than 256 decoder layers share dropout call-site ids with other layers")

    def forward(self, x: Tensor, memory: Tensor, dropout_key: Optional[Tensor] = None, causal: bool = False) -> Tensor:
        for i, layer in enumerate(self.layers):
            x = layer(x, memory, dropout_key, i, causal)
        return x


class PositionalEmbedding(nn.Module):
    """learned positions added to the tokens, then LayerNorm (+ dropout): reference networks/vit.py:33-58"""

    def __init__(self, max_length: int, d_model: int, dropout: float, batch_first: bool, device=None, dtype=None):
        super().__init__()
        self.d_model, self.batch_first = d_model, batch_first
        self.position_embeddings = nn.Embedding(max_length, d_model, device=device, dtype=dtype)
        self.LayerNorm = TokenLayerNorm(d_model)
        self.p = float(dropout)

    def forward(self, input: Tensor, dropout_key: Optional[Tensor] = None) -> Tensor:
        if input.size(-1) != self.d_model:
            raise RuntimeError("the feature number of `input` must be equal to d_model")
        batched = input.dim() == 3
        seq_dim = int(batched and self.batch_first)
        w = self.position_embeddings.weight
        n = input.shape[seq_dim]
        if batched and self.batch_first and input.is_cuda:
            # the positions broadcast over the batch: materialised once, their gradient summed over the batch by the library
            # (functional.expand_batch) -- into the parameter's gradient slot when all max_length positions are in use
            res = HF.expand_batch(w if n == w.shape[0] else w[:n].contiguous(), input.shape[0], w if n == w.shape[0] else None)
        else:
            pos = w[:n]
            if batched:
                pos = pos.unsqueeze(int(not self.batch_first))
            res = pos.expand_as(input)
        out = self.LayerNorm(input, residual=res)
        if not (self.training and self.p > 0):
            return out
        if dropout_key is not None and self.d_model % 4 == 0:
            return HF.dropout_tokens(out, self.p, dropout_key, 4000)
        return F.dropout(out, self.p, True)


class _Patchify(nn.Module):
    """'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'"""

    def __init__(self, p1: int, p2: int):
        super().__init__()
        self.p1, self.p2 = p1, p2

    def forward(self, x: Tensor) -> Tensor:
        b, c, hh, ww = x.shape
        h, w = hh // self.p1, ww // self.p2
        return x.reshape(b, c, h, self.p1, w, self.p2).permute(0, 2, 4, 3, 5, 1).reshape(b, h * w, self.p1 * self.p2 * c)


class _Unpatchify(nn.Module):
    """'b (h w) (p1 p2 c) -> b c (h p1) (w p2)'"""

    def __init__(self, h: int, p1: int, p2: int):
        super().__init__()
        self.h, self.p1, self.p2 = h, p1, p2

    def forward(self, x: Tensor) -> Tensor:
        b, n, d = x.shape
        h, w = self.h, n // self.h
        c = d // (self.p1 * self.p2)
        return x.reshape(b, h, w, self.p1, self.p2, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, h * self.p1, w * self.p2)


class ViT(nn.Module):
    def __init__(self, image_size: Union[int, Tuple[int, int]], dim: int, patch_size: Optional[Union[int, Tuple[int, int]]] = None,
                 depth: int = 6, preprocess_depth: Optional[int] = None, heads: int = 8, mlp_dim: Optional[int] = None,
                 channels: int = 3, dropout: float = 0.1, emb_dropout: float = 0., n_embed_tokens: Optional[Union[int, str]] = 1,
                 n_input_tokens: Optional[Union[int, str]] = None, output_tokens: Union[str, Sequence[str]] = "embed",
                 patch_to_embed: bool = True, embed_to_patch: bool = False, num_classes: Optional[int] = None,
                 time_dependant: bool = False, causal_mask: bool = False):
        super().__init__()
        if time_dependant:
            raise NotImplementedError("`time_dependant` (Fourier time token) is not implemented on the MI355X path")
        self.dim, self.causal_mask = dim, causal_mask
        self.attn_dropout = float(dropout)
        image_height, image_width = pair(image_size)
        mlp_dim = mlp_dim or dim * 4
        if patch_size is None:
            patch_size = min(image_height // 4, 16), min(image_width // 4, 16)
        patch_height, patch_width = pair(patch_size)
        if image_height % patch_height or image_width % patch_width:
            raise ValueError("Image dimensions must be divisible by the patch size.")
        n_patch_h, n_patch_w = image_height // patch_height, image_width // patch_width
        self.num_patches, self.patch_dim = n_patch_h * n_patch_w, channels * patch_height * patch_width
        self.n_tokens = {"input": self.num_patches if n_input_tokens is None else n_input_tokens,
                         "embed": self.num_patches if n_embed_tokens is None else n_embed_tokens,
                         "class": int(num_classes is not None), "time": 0}
        self.total_num_tokens = sum(self.n_tokens.values())
        self.token_indices, at = {}, 0
        for kind, count in self.n_tokens.items():
            self.token_indices[kind] = list(range(at, at + count))
            at += count
        output_tokens = [output_tokens] if isinstance(output_tokens, str) else list(output_tokens)
        if not all(kind in self.token_indices for kind in output_tokens):
            raise ValueError(f"`output_tokens` must contain only keys within {self.token_indices.keys()}")
        self.output_tokens_indices, self.cross_tokens_indices = [], []
        for kind, idx in self.token_indices.items():
            (self.output_tokens_indices if kind in output_tokens else self.cross_tokens_indices).extend(idx)

        # the reference's construction order (= its random-number consumption order): patch_to_embed, embed_to_patch,
        # embed_token, class_token, positional_embed, transformer
        self.patch_to_embed = nn.Sequential(_Patchify(patch_height, patch_width), TokenLinear(self.patch_dim, dim)) \
            if patch_to_embed else nn.Identity()
        self.embed_to_patch = nn.Sequential(TokenLinear(dim, self.patch_dim), _Unpatchify(n_patch_h, patch_height, patch_width)) \
            if embed_to_patch else nn.Identity()
        self.embed_token = nn.Parameter(torch.randn(1, self.n_tokens["embed"], dim)) if self.n_tokens["embed"] > 0 else None
        self.class_token = nn.Embedding(num_classes, dim) if self.n_tokens["class"] > 0 else None
        self.time_token = None
        self.positional_embed = PositionalEmbedding(self.total_num_tokens, dim, emb_dropout, True)
        if preprocess_depth is None:
            self.prepocess = None   # (sic: the reference's attribute and state_dict name)
            self.transformer = _Encoder(nn.TransformerEncoder(
                encoder_layer=nn.TransformerEncoderLayer(dim, heads, mlp_dim, dropout, batch_first=True), num_layers=depth,
                enable_nested_tensor=False))
        else:
            if not self.cross_tokens_indices:
                raise ValueError("the cross-attention ViT needs tokens outside `output_tokens` to attend to")
            self.prepocess = _Encoder(nn.TransformerEncoder(
                encoder_layer=nn.TransformerEncoderLayer(dim, heads, mlp_dim, dropout, batch_first=True),
                num_layers=preprocess_depth, enable_nested_tensor=False)) if preprocess_depth > 0 else nn.Identity()
            self.transformer = _Decoder(nn.TransformerDecoder(
                decoder_layer=nn.TransformerDecoderLayer(dim, heads, mlp_dim, dropout, batch_first=True), num_layers=depth))
        self.out_size = torch.Size([channels, image_height, image_width]) if embed_to_patch else \
            torch.Size([len(self.output_tokens_indices), dim])

    def _select(self, tokens: Tensor, which: str) -> Tensor:
        """tokens[:, indices]: a slice when the indices are one run (an index LIST would be uploaded from the host on every
        call, which a graph capture cannot hold), else through a device-resident index tensor"""
        idx = self.output_tokens_indices if which == "output" else self.cross_tokens_indices
        if not idx:   # e.g. output_tokens='time' without a time token: the reference's tokens[:, []] is an empty selection too
            return tokens[:, :0]
        if idx == list(range(idx[0], idx[0] + len(idx))):
            return tokens[:, idx[0]:idx[0] + len(idx)]
        cache = self.__dict__.setdefault("_index_cache", {})
        cached = cache.get(which)
        if cached is None or cached.device != tokens.device:
            cached = cache[which] = torch.tensor(idx, device=tokens.device)
        return tokens[:, cached]

    def _next_dropout_key(self, device) -> Optional[Tensor]:
        """{seed, call counter} for the attention-probability dropout of this forward pass, or None when nothing is
        dropped.  Seeded from torch's default generator on first use (an eager step: it reads the host generator); the
        counter advances by a device-side add, so a captured step draws new masks on every replay."""
        if not self.training or max(self.attn_dropout, self.positional_embed.p) <= 0:
            return None
        key = self.__dict__.get("_dropout_key")
        if key is None or key.device != device:
            key = self.__dict__["_dropout_key"] = HF.new_dropout_key(device)
        key[1:].add_(1)
        return key

    def _add_class_token(self, x: Tensor, labels: Optional[Tensor]) -> Tensor:
        if labels is not None and self.class_token is None:
            warnings.warn("given conditional argument `labels` but `self.class_token` is None. To enable a class-conditioned "
                          "ViT, use `ViT(num_classes=...)`.")
        if self.class_token is not None:
            if labels is None:
                raise ValueError("`num_classes` specified but `labels` is None. Can't infer the class token.")
            x = torch.cat((x, HF.embedding(self.class_token.weight, labels).unsqueeze(1)), dim=1)
        return x

    def _add_time_token(self, x: Tensor, time: Optional[Tensor]) -> Tensor:
        if time is not None:
            warnings.warn("given conditional argument `time` but `self.time_token` is None.")
        return x

    def _add_embed_token(self, x: Tensor) -> Tensor:
        if self.embed_token is not None:
            x = torch.cat((x, HF.expand_batch(self.embed_token, x.size(0), self.embed_token)), dim=1)
        return x

    def forward(self, x: Tensor, labels: Optional[Tensor] = None, time: Optional[Tensor] = None) -> Tensor:
        x = self.patch_to_embed(x)
        x = self._add_embed_token(x)
        x = self._add_class_token(x, labels)
        x = self._add_time_token(x, time)
        key = self._next_dropout_key(x.device)
        x = self.positional_embed(x, key)
        if self.prepocess is None:
            out = self._select(self.transformer(x, key, causal=self.causal_mask), "output")
        else:
            memory = self._select(x, "cross")
            if not isinstance(self.prepocess, nn.Identity):
                memory = self.prepocess(memory, key)
            out = self.transformer(self._select(x, "output"), memory, key, causal=self.causal_mask)
        if not isinstance(self.embed_to_patch, nn.Identity):
            out = out[:, -self.num_patches:]
        return self.embed_to_patch(out)


class AutoRegressive(ViT):
    """``AutoRegressive(vocab_size, **vit_kwargs)`` (reference networks/vit.py:249-260): token ids -> vocabulary embedding -> the
    ViT on the embedded tokens -> Linear head over the vocabulary (keys ``vocab_embed.weight``, ``head.{weight,bias}``)."""

    def __init__(self, vocab_size: int, **vit_kwargs):
        super().__init__(**vit_kwargs)
        self.vocab_embed = nn.Embedding(vocab_size, self.dim)
        self.head = TokenLinear(self.dim, vocab_size)

    def forward(self, x: Tensor, labels: Optional[Tensor] = None, time: Optional[Tensor] = None) -> Tensor:
        return self.head(super().forward(self.vocab_embed(x), labels, time))
