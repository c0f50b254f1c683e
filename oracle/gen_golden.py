"""
TEST INFRASTRUCTURE -- golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

Imports the *real* reference through ``oracle/ref_import.py``, drives its hot-path classes/functions with
deterministic closed-form inputs and weights (``oracle/detfill.py``) and records inputs + outputs as small
``.npz`` fixtures under ``tests/golden/``.  Only numbers are written; no reference source or bytecode.

    python oracle/gen_golden.py            # regenerates every fixture

Fixtures:
  convlayer.npz      ConvLayer fwd/bwd per distinct layer geometry, B=4, incl. BN running stats
  attention.npz      QKVAttention fwd/bwd for every (T,H,C) of the MNIST and CIFAR configs, B=2
  cnn_small.npz      CNN encoder+decoder (capacity 2) fwd + all parameter grads, residual add/None/cat
  nelbo_mnist.npz    VAE.nelbo on the MNIST test config (B=6): losses, preds/latents slices, per-parameter
                     grad checksums, parameter checksums after one Adam step
  prior.npz          GaussianPrior.forward (z, loss) incl. cosine annealing
  sinkhorn.npz       sinkhorn_log for several N, reg, thresholds, dtypes, incl. batched early exit
  gaussian_ot.npz    GaussianModel.update/fit, mean_cov, w2_gaussian, compute_transport_operators,
                     apply_transport, GaussianTransport.compute/transport
  codebook.npz       CodebookModel.predict argmax indices + encodings
  codebook_kmeans.npz  CodebookModel.update/fit/predict/w2 (streaming k-means)
  discrete.npz       CodebookModel in 'mean' mode, DiscreteTransport.compute/transport, CodebookPrior.forward
  gmm.npz            GaussianMixtureModel (diagonal) update/fit/energy/w2, GMMTransport.compute/transport
  vit.npz            ViT encoder / decoder (reference networks/vit.py) fwd + input and parameter gradients, dropout 0
  stochastic.npz     compute_transport_operators(stochastic=True) (eq. 19) for degenerate sources + apply_transport with noise
  gmm_full.npz       full covariances: batch_w2_dissimilarity_gaussian, batch_ot_gmm, gaussian_barycenter, GaussianMixtureModel
  mixture_modes.npz  CodebookPrior in the soft 'mean' mode with the entropy loss (values + encoder gradient); Gumbel assignment modes
  nelbo_b32.npz      VAE.nelbo at batch 32 with torch's default initialisation (seeded): the well-conditioned whole-network pin
  w2_prior.npz       GaussianModel._stats + mean_cov + w2_gaussian under torch.autograd: loss and dL/dz (GaussianW2Prior)
  vit_vae.npz        ConditionalGaussianPrior fwd/bwd (+ EMA variant) and VAE.nelbo of the conditional ViT VAE
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as R  # noqa: E402
from detfill import fill_state_dict, fill_vit_state_dict, det_input, mnist_like, normal  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(4)


def npy(t):
    return t.detach().cpu().numpy()


def save(name, d):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **d)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB, {len(d)} arrays")


# ------------------------------------------------------------------------------------------------ G1
CONV_CASES = [
    # name, cls, cin, cout, hw, kwargs
    ("enc_first", "ConvLayer", 1, 8, 32, dict(down_sample=2, normalization="batchnorm", activation="relu")),
    ("enc_same", "ConvLayer", 8, 8, 16, dict(normalization="batchnorm", activation="relu")),
    ("enc_down", "ConvLayer", 8, 16, 16, dict(down_sample=2, normalization="batchnorm", activation="relu")),
    ("enc_last", "ConvLayer", 16, 32, 2, dict(down_sample=2, normalization="batchnorm", activation="relu")),
    ("same_1x1res", "ConvLayer", 12, 12, 1, dict(normalization="batchnorm", activation="relu")),
    ("dec_up", "ConvLayer", 16, 8, 4, dict(up_sample=2, normalization="batchnorm", activation="relu")),
    ("dec_first", "ConvLayer", 12, 6, 1, dict(up_sample=2, normalization="batchnorm", activation="relu")),
    ("dec_last", "ConvLayer", 8, 1, 16, dict(up_sample=2, normalization="batchnorm", activation="relu")),
    ("dec_11", "ConvLayer", 1, 1, 32, dict(normalization="batchnorm", activation="relu")),
    ("rgb_in", "ConvLayer", 3, 16, 32, dict(down_sample=2, normalization="batchnorm", activation="relu")),
    ("qkv", "Conv1x1", 8, 24, 16, dict(normalization="batchnorm")),
    ("qkv1", "Conv1x1", 1, 3, 32, dict(normalization="batchnorm")),
    ("proj", "Conv1x1", 8, 8, 16, dict()),
    ("skip_down", "Conv1x1", 8, 16, 16, dict(down_sample=2, normalization="batchnorm")),
    ("skip_up", "Conv1x1", 16, 8, 4, dict(up_sample=2, normalization="batchnorm")),
    ("skip_up1", "Conv1x1", 8, 1, 16, dict(up_sample=2, normalization="batchnorm")),
    ("nonorm_relu", "ConvLayer", 4, 8, 8, dict(activation="relu")),
    # the options configs/vae/defaults_imagenet.yaml:26-27 turns on (activation: leaky, equalized_lr: 1.) and the other
    # activations of cnn.py:128-147; equalized_lr = 2 / 0.5 so that weight * conv_scale * lr_mult and bias * lr_mult both show
    ("leaky_eq", "ConvLayer", 8, 8, 8, dict(normalization="batchnorm", activation="leaky", equalized_lr=1.)),
    ("leaky_down_eq2", "ConvLayer", 3, 8, 16, dict(down_sample=2, normalization="batchnorm", activation="leaky_relu", equalized_lr=2.)),
    ("selu_down", "ConvLayer", 4, 8, 8, dict(down_sample=2, normalization="batchnorm", activation="selu")),
    ("gelu_up", "ConvLayer", 8, 4, 4, dict(up_sample=2, normalization="batchnorm", activation="gelu")),
    ("silu_nonorm", "ConvLayer", 4, 8, 8, dict(activation="silu")),
    ("swish_bn_1ch", "ConvLayer", 1, 4, 16, dict(normalization="batchnorm", activation="swish")),
    ("eq_1x1", "Conv1x1", 8, 24, 8, dict(normalization="batchnorm", equalized_lr=0.5)),
    # GroupNorm(div_sqrt(C), C) / InstanceNorm2d(C) in place of BatchNorm (cnn.py:123-124)
    ("gn_relu", "ConvLayer", 8, 8, 8, dict(normalization="groupnorm", activation="relu")),
    ("gn_down_leaky", "ConvLayer", 12, 8, 8, dict(down_sample=2, normalization="groupnorm", activation="leaky")),
    ("gn_1x1_up", "Conv1x1", 16, 8, 4, dict(up_sample=2, normalization="groupnorm")),
    ("in_silu", "ConvLayer", 6, 8, 8, dict(normalization="instancenorm", activation="silu")),
    ("in_relu_up", "ConvLayer", 8, 4, 4, dict(up_sample=2, normalization="instancenorm", activation="relu")),
    # FiLM conditioning (`additional_embed`, cnn.py:114-116,160-181): forward(x, embed) with embed [B, E]
    ("film_relu", "ConvLayer", 8, 8, 8, dict(normalization="batchnorm", activation="relu", additional_embed=5)),
    ("film_leaky_eq", "ConvLayer", 4, 8, 8, dict(down_sample=2, normalization="batchnorm", activation="leaky", equalized_lr=2., additional_embed=6)),
    ("film_1x1_gn", "Conv1x1", 8, 24, 8, dict(normalization="groupnorm", additional_embed=5)),
    # grouped / dilated layers (cnn.py:66-67,103-104: nn.Conv2d(..., dilation, groups); GroupNorm(div_sqrt(C // groups), C) :123)
    ("grp2_relu", "ConvLayer", 8, 8, 8, dict(normalization="batchnorm", activation="relu", groups=2)),
    ("grp4_down_leaky_eq", "ConvLayer", 8, 16, 8, dict(down_sample=2, normalization="batchnorm", activation="leaky", equalized_lr=2., groups=4)),
    ("grp2_1x1_gn", "Conv1x1", 8, 24, 8, dict(normalization="groupnorm", groups=2)),
    ("dil2_relu", "ConvLayer", 4, 8, 8, dict(normalization="batchnorm", activation="relu", dilation=2, padding=2)),
    ("dil3_grp2_up", "ConvLayer", 8, 4, 4, dict(up_sample=2, normalization="batchnorm", activation="relu", dilation=3, groups=2)),
    ("dil2_nobias_silu", "ConvLayer", 6, 6, 8, dict(activation="silu", dilation=2, groups=3, bias=False)),
    # module-valued resampling (cnn.py:97,106): the user's module is applied as given, up between activation and convolution,
    # down behind the convolution
    ("mod_up_bilinear", "ConvLayer", 4, 8, 4, dict(up_sample=torch.nn.Upsample(scale_factor=2, mode="bilinear"), normalization="batchnorm", activation="relu")),
    ("mod_down_avgpool", "ConvLayer", 4, 8, 8, dict(down_sample=torch.nn.AvgPool2d(2), normalization="batchnorm", activation="leaky")),
    ("up4_relu", "ConvLayer", 4, 4, 2, dict(up_sample=4, normalization="batchnorm", activation="relu")),   # nn.Upsample(scale_factor=4), cnn.py:107
    # strides other than 1 / 2 and footprints beyond 7 x 7 (round 4: the direct-convolution fallback): `down_sample = s` makes a
    # (2 s) x (2 s) kernel with stride s, padding s - 1 (cnn.py:98-101) -- the layers of CNN(scaling_factor=4)
    ("down4_relu", "ConvLayer", 4, 8, 16, dict(down_sample=4, normalization="batchnorm", activation="relu")),
    ("down4_skip", "Conv1x1", 8, 16, 8, dict(down_sample=4, normalization="batchnorm")),
    ("down8_leaky", "ConvLayer", 3, 8, 32, dict(down_sample=8, activation="leaky")),
    ("stride3_k5", "ConvLayer", 4, 6, 9, dict(kernel_size=5, stride=3, padding=2, normalization="batchnorm", activation="relu")),
    ("k9_same_relu", "ConvLayer", 3, 4, 12, dict(kernel_size=9, padding=4, activation="relu")),
    ("dil4_grp2_silu", "ConvLayer", 4, 6, 12, dict(normalization="batchnorm", activation="silu", dilation=4, padding=4, groups=2)),
    ("up2_k9", "ConvLayer", 4, 4, 4, dict(up_sample=2, kernel_size=9, padding=4, normalization="batchnorm", activation="relu")),
]


def gen_convlayer():
    cnn = R.ref("networks.cnn")
    out = {}
    for name, cls, cin, cout, hw, kw in CONV_CASES:
        layer = getattr(cnn, cls)(cin, cout, **kw)
        layer.train()
        fill_state_dict(layer.state_dict())
        x = det_input((4, cin, hw, hw), phase=0.3).requires_grad_(True)
        emb = None
        if kw.get("additional_embed"):
            emb = det_input((4, kw["additional_embed"]), phase=0.8, amp=0.9).requires_grad_(True)
        y = layer(x, emb) if emb is not None else layer(x)
        g = det_input(tuple(y.shape), phase=1.1, amp=0.7)
        y.backward(g)
        if emb is not None:
            out[f"{name}/embed"], out[f"{name}/gembed"] = npy(emb), npy(emb.grad)
        out[f"{name}/x"] = npy(x)
        out[f"{name}/gy"] = npy(g)
        out[f"{name}/y"] = npy(y)
        out[f"{name}/gx"] = npy(x.grad)
        for k, p in layer.named_parameters():
            out[f"{name}/param/{k}"] = npy(p)
            out[f"{name}/grad/{k}"] = npy(p.grad)
        for k, b in layer.named_buffers():
            out[f"{name}/buf/{k}"] = npy(b)
    save("convlayer.npz", out)


# ------------------------------------------------------------------------------------------------ G2
ATTN_CASES = [(256, 4, 2), (64, 4, 4), (16, 8, 4), (4, 8, 8), (1, 16, 16), (1024, 1, 1),
              (256, 4, 4), (64, 8, 4), (16, 8, 8), (4, 16, 8), (1, 32, 16), (1024, 3, 1)]


def gen_attention():
    nu = R.ref("networks.nets_utils")
    out = {}
    for t, h, c in ATTN_CASES:
        attn = nu.QKVAttention(h)
        qkv = det_input((2, 3 * h * c, t), phase=0.7, amp=1.3).requires_grad_(True)
        a = attn(qkv)
        g = det_input(tuple(a.shape), phase=2.0)
        a.backward(g)
        key = f"T{t}_H{h}_C{c}"
        out[f"{key}/qkv"] = npy(qkv)
        out[f"{key}/out"] = npy(a)
        out[f"{key}/gout"] = npy(g)
        out[f"{key}/gqkv"] = npy(qkv.grad)
    save("attention.npz", out)


# ------------------------------------------------------------------------------------------------ G3
def gen_cnn_small():
    cnn = R.ref("networks.cnn")
    out = {}
    for residual in ("add", None, "cat"):
        tag = str(residual)
        cap = 4 if residual == "cat" else 2
        enc = cnn.CNN(1, 16, 16, 1, capacity=cap, down_sample=True, residual=residual)
        nets = [(enc, "enc", det_input((3, 1, 16, 16), 0.2))]
        if residual != "cat":  # a 1-channel output cannot be split in two halves (cnn.py:308)
            dec = cnn.CNN(8, 1, 1, 16, capacity=cap, up_sample=True, residual=residual)
            nets.append((dec, "dec", det_input((3, 8, 1, 1), 0.9)))
        for net, nm, xin in nets:
            net.train()
            fill_state_dict(net.state_dict())
            x = xin.clone().requires_grad_(True)
            y = net(x)
            g = det_input(tuple(y.shape), 1.7, 0.5)
            y.backward(g)
            out[f"{tag}/{nm}/x"] = npy(x)
            out[f"{tag}/{nm}/y"] = npy(y)
            out[f"{tag}/{nm}/gy"] = npy(g)
            out[f"{tag}/{nm}/gx"] = npy(x.grad)
            out[f"{tag}/{nm}/heads"] = np.array([int(b.block[2].attention.n_heads) if hasattr(b.block[2], "attention")
                                                  else 0 for b in net])
            for k, p in net.named_parameters():
                out[f"{tag}/{nm}/grad/{k}"] = npy(p.grad)
            for k, b in net.named_buffers():
                if not k.endswith("num_batches_tracked"):
                    out[f"{tag}/{nm}/buf/{k}"] = npy(b)
    save("cnn_small.npz", out)


def gen_cnn_small_opts():
    """The same small encoder / decoder with the options of the reference's configs/vae/defaults_imagenet.yaml:26-27
    (``activation: leaky``, ``equalized_lr: 1.``; residual "add"): every ConvLayer, Conv1x1 and AttentionBlock of the network
    runs the multipliers of cnn.py:114-118,186-188 and LeakyReLU(0.2) in place of ReLU."""
    cnn = R.ref("networks.cnn")
    out = {}
    kw = dict(capacity=2, residual="add", activation="leaky", equalized_lr=1.0)
    enc = cnn.CNN(1, 16, 16, 1, down_sample=True, **kw)
    dec = cnn.CNN(8, 1, 1, 16, up_sample=True, **kw)
    for net, nm, xin in ((enc, "enc", det_input((3, 1, 16, 16), 0.2)), (dec, "dec", det_input((3, 8, 1, 1), 0.9))):
        net.train()
        fill_state_dict(net.state_dict())
        x = xin.clone().requires_grad_(True)
        y = net(x)
        g = det_input(tuple(y.shape), 1.7, 0.5)
        y.backward(g)
        out[f"{nm}/x"], out[f"{nm}/y"], out[f"{nm}/gy"], out[f"{nm}/gx"] = npy(x), npy(y), npy(g), npy(x.grad)
        for k, p in net.named_parameters():
            out[f"{nm}/grad/{k}"] = npy(p.grad)
        for k, b in net.named_buffers():
            if not k.endswith("num_batches_tracked"):
                out[f"{nm}/buf/{k}"] = npy(b)
    save("cnn_small_opts.npz", out)


CNN_VARIANTS = {
    # name: (ctor args, ctor kwargs, input shape, embed width | None) -- corners of the CNN / ConvBlock constructors that no config uses
    "cat_gn_silu_3layers": ((3, 12, 8, 2), dict(capacity=4, down_sample=True, residual="cat", n_layers=3, normalization="groupnorm",
                                                 activation="silu"), (3, 3, 8, 8), None),
    "cat_gn_silu_3layers_cap8": ((3, 16, 8, 2), dict(capacity=8, down_sample=True, residual="cat", n_layers=3, normalization="groupnorm",
                                                      activation="silu"), (3, 3, 8, 8), None),
    "add_film_gelu_eq_up": ((6, 3, 2, 8), dict(capacity=4, up_sample=True, residual="add", additional_embed=5, activation="gelu",
                                                equalized_lr=1.0), (3, 6, 2, 2), 5),
    "intermediate_nonorm_selu": ((4, 8), dict(intermediate_features=[6, 10], residual=None, normalization=None, activation="selu"),
                                 (3, 4, 4, 4), None),
    "intermediate_nonorm_selu_res": ((4, 8, 4, 4), dict(intermediate_features=[6, 10], residual=None, normalization=None, activation="selu"),
                                     (3, 4, 4, 4), None),
    "add_noattn_dilated_res": ((4, 4, 8, 8), dict(intermediate_features=[8], residual="add", max_attn_res=0, dilation=2, padding=2),
                               (3, 4, 8, 8), None),
    "grouped_in_leaky_nobias": ((2, 8, 8, 2), dict(capacity=4, down_sample=True, residual="add", groups=2, normalization="instancenorm",
                                                    activation="leaky", bias=False), (3, 2, 8, 8), None),
    "cat_1layer_k1_noattn_up": ((8, 2, 2, 8), dict(capacity=4, up_sample=True, residual="cat", max_attn_res=1, n_layers=1, kernel_size=1,
                                                    padding=0), (3, 8, 2, 2), None),
    "add_noattn_dilated": ((4, 4), dict(intermediate_features=[8], residual="add", max_attn_res=0, dilation=2, padding=2), (3, 4, 8, 8), None),
    # round 4: a scaling factor of 4 per block (get_block_scaling, cnn.py:605-621): 8 x 8 kernels with stride 4 on the way down,
    # nn.Upsample(4) + 3 x 3 on the way up
    "down4_add": ((2, 16, 16, 1), dict(capacity=4, down_sample=4, residual="add"), (3, 2, 16, 16), None),
    "up4_add": ((8, 2, 1, 16), dict(capacity=4, up_sample=4, residual="add"), (3, 8, 1, 1), None),
}


def gen_cnn_variants():
    """Whole CNNs at corners of the constructor space (residual "cat" with three layers per block and GroupNorm, FiLM embeddings with
    equalized_lr on the up-sampling path, `intermediate_features`, grouped InstanceNorm blocks without biases, one-layer 1x1 blocks,
    dilated blocks without attention): output, input / embedding gradients, every parameter gradient and buffer."""
    cnn = R.ref("networks.cnn")
    out = {}
    for name, (args, kw, xshape, ew) in CNN_VARIANTS.items():
        try:
            net = cnn.CNN(*args, **kw)
        except Exception as e:  # noqa: BLE001 -- a constructor corner the reference itself rejects: the error type is the golden
            out[f"{name}/error"] = np.frombuffer(type(e).__name__.encode(), dtype=np.uint8)
            print(f"  {name}: the reference raises {type(e).__name__}: {e}")
            continue
        net.train()
        fill_state_dict(net.state_dict())
        x = det_input(xshape, 0.35).requires_grad_(True)
        emb = det_input((xshape[0], ew), 0.8, 0.9).requires_grad_(True) if ew else None
        y = net(x, emb) if emb is not None else net(x)
        g = det_input(tuple(y.shape), 1.3, 0.6)
        y.backward(g)
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/gy"], out[f"{name}/gx"] = npy(x), npy(y), npy(g), npy(x.grad)
        if emb is not None:
            out[f"{name}/embed"], out[f"{name}/gembed"] = npy(emb), npy(emb.grad)
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = npy(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
        for k, b in net.named_buffers():
            if not k.endswith("num_batches_tracked"):
                out[f"{name}/buf/{k}"] = npy(b)
        print(f"  {name}: y {tuple(y.shape)}, {sum(p.numel() for p in net.parameters())} parameters")
    save("cnn_variants.npz", out)


# ------------------------------------------------------------------------------------------------ G4
class _FixedEps:
    """Makes Normal.rsample draw a recorded eps (prior/gaussian.py:93 uses the global RNG)."""

    def __init__(self, eps):
        self.eps = eps

    def __enter__(self):
        import torch.distributions.normal as N
        self._mod, self._orig = N, N._standard_normal
        N._standard_normal = lambda shape, dtype, device: self.eps.to(dtype).reshape(shape)

    def __exit__(self, *a):
        self._mod._standard_normal = self._orig


def _mnist_vae(residual="add", loss_coeff=0.1):
    cnn, pg, vae = R.ref("networks.cnn"), R.ref("prior.gaussian"), R.ref("model.vae")
    enc = cnn.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual=residual)
    dec = cnn.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual=residual)
    fill_state_dict(enc.state_dict())
    fill_state_dict(dec.state_dict())
    m = vae.VAE(metrics=R._MetricCollection(), encoder=enc, decoder=dec, prior=pg.GaussianPrior(loss_coeff=loss_coeff))
    m.train()
    return m


def gen_nelbo_expansion():
    """`VAE(expansion=3)` (model/vae.py:158-229; utils.replicate_batch / mean_replicated_batch, utils/__init__.py:154-175): the encoder
    output is replicated three times before the prior (three noise draws per image), the decoder runs on 3B latents, the reconstruction
    loss sees the mean over the replicas, `preds` / `latents` are the first replica.  Small network (capacity 4, 16x16, B = 8)."""
    cnn, pg, vae = R.ref("networks.cnn"), R.ref("prior.gaussian"), R.ref("model.vae")
    enc = cnn.CNN(1, 16, 16, 1, capacity=4, down_sample=True, residual="add")
    dec = cnn.CNN(8, 1, 1, 16, capacity=4, up_sample=True, residual="add")
    fill_state_dict(enc.state_dict())
    fill_state_dict(dec.state_dict())
    m = vae.VAE(metrics=R._MetricCollection(), encoder=enc, decoder=dec, prior=pg.GaussianPrior(loss_coeff=0.1), expansion=3)
    m.train()
    B = 8
    x = det_input((B, 1, 16, 16), 0.4)
    eps = normal((3 * B, 8, 1, 1), seed=47)
    with _FixedEps(eps):
        loss, logs, art = m.nelbo({"samples": x, "target": x, "kwargs": {}}, 0)
    loss.backward()
    out = {"x": npy(x), "eps": npy(eps)}
    out["loss"] = npy(torch.stack([logs["train/loss/total"], logs["train/loss/recon"], logs["train/loss/prior"]]))
    out["preds"], out["latents"], out["preds_mean"] = npy(art["preds"]), npy(art["latents"]), npy(art["preds_mean"])
    for pre, net in (("encoder.", m.encoder), ("decoder.", m.decoder)):
        for k, p in net.named_parameters():
            out[f"grad/{pre}{k}"] = npy(p.grad)
        for k, b in net.named_buffers():
            if not k.endswith("num_batches_tracked"):
                out[f"buf/{pre}{k}"] = npy(b)
    save("nelbo_expansion.npz", out)


def gen_nelbo():
    out = {}
    for residual in ("add", None):
        tag = str(residual)
        B = 6
        m = _mnist_vae(residual)
        x = mnist_like(B, seed=42)
        eps = normal((B, 128, 1, 1), seed=43)
        with _FixedEps(eps):
            loss, logs, art = m.nelbo({"samples": x, "target": x, "kwargs": {}}, 0)
        loss.backward()
        out[f"{tag}/loss"] = npy(torch.stack([logs["train/loss/total"], logs["train/loss/recon"], logs["train/loss/prior"]]))
        out[f"{tag}/preds"] = npy(art["preds"][:2])
        out[f"{tag}/latents"] = npy(art["latents"])
        names, gsum, gl2, psum = [], [], [], []
        params = []
        for pre, net in (("encoder.", m.encoder), ("decoder.", m.decoder)):
            for k, p in net.named_parameters():
                names.append(pre + k)
                gsum.append(p.grad.double().sum().item())
                gl2.append(p.grad.double().norm().item())
                params.append(p)
        out[f"{tag}/param_names"] = np.array(names)
        out[f"{tag}/grad_sum"] = np.array(gsum)
        out[f"{tag}/grad_l2"] = np.array(gl2)
        # a few full gradients for exact comparisons
        for k in ("encoder.0.block.0.weight", "encoder.4.block.2.qkv.weight", "decoder.4.block.2.qkv.weight",
                  "decoder.0.skip.weight", "encoder.2.block.1._normalization.weight", "decoder.3.block.1.bias"):
            if k in names:
                out[f"{tag}/grad_full/{k}"] = npy(params[names.index(k)].grad)
        # the same loss/gradients evaluated by the reference in fp64: quantifies how much of the fp32 result is
        # rounding noise (BatchNorm over 6 samples at 1x1 resolution is badly conditioned), so that tests can bound a
        # port by the reference's own fp32 accuracy instead of an arbitrary number
        m64 = _mnist_vae(residual).double()
        with _FixedEps(eps.double()):
            loss64, _, _ = m64.nelbo({"samples": x.double(), "target": x.double(), "kwargs": {}}, 0)
        loss64.backward()
        p64 = [p for net in (m64.encoder, m64.decoder) for _, p in net.named_parameters()]
        out[f"{tag}/grad_l2_f64"] = np.array([p.grad.norm().item() for p in p64])
        for k in list(out):
            if k.startswith(f"{tag}/grad_full/"):
                out[k.replace("grad_full/", "grad_full_f64/")] = npy(p64[names.index(k.split("grad_full/")[1])].grad)
        # sensitivity of the reference itself: the same fp32 computation with input and weights perturbed by ~1 ulp
        # (4 draws).  The spread of its gradients is the resolution at which ANY fp32 implementation of this
        # (deliberately hostile: sine weights, BatchNorm over 6 samples) case can be compared.
        spread_fwd = {"loss": 0.0, "preds": 0.0, "latents": 0.0}
        base_fwd = {"loss": torch.tensor(out[f"{tag}/loss"]), "preds": torch.tensor(out[f"{tag}/preds"]),
                    "latents": torch.tensor(out[f"{tag}/latents"])}
        spread_l2 = np.zeros(len(params))
        spread_full = {k: 0.0 for k in out if k.startswith(f"{tag}/grad_full/")}
        base_l2 = np.array(gl2)
        for trial in range(4):
            gen = torch.Generator().manual_seed(900 + trial)
            xp = x * (1.0 + (torch.rand(x.shape, generator=gen) - 0.5) * 2.4e-7)
            mp = _mnist_vae(residual)
            with torch.no_grad():  # ... and every weight by ~1 ulp: the analogue of a different-but-valid rounding
                for p_ in mp.parameters():  # order inside each layer
                    p_.mul_(1.0 + (torch.rand(p_.shape, generator=gen) - 0.5) * 2.4e-7)
            with _FixedEps(eps):
                lp, logp, artp = mp.nelbo({"samples": xp, "target": xp, "kwargs": {}}, 0)
            lp.backward()
            cur = {"loss": torch.stack([logp["train/loss/total"], logp["train/loss/recon"], logp["train/loss/prior"]]).detach(),
                   "preds": artp["preds"][:2].detach(), "latents": artp["latents"].detach()}
            for kf in spread_fwd:
                spread_fwd[kf] = max(spread_fwd[kf], (cur[kf] - base_fwd[kf]).abs().max().item())
            pp = [p for net in (mp.encoder, mp.decoder) for _, p in net.named_parameters()]
            l2p = np.array([p.grad.double().norm().item() for p in pp])
            spread_l2 = np.maximum(spread_l2, np.abs(l2p - base_l2))
            for k in spread_full:
                i = names.index(k.split("grad_full/")[1])
                spread_full[k] = max(spread_full[k], (pp[i].grad - params[i].grad).abs().max().item())
        for kf, vf in spread_fwd.items():
            out[f"{tag}/{kf}_spread"] = np.array(vf)
        out[f"{tag}/grad_l2_spread"] = spread_l2
        for k, v in spread_full.items():
            out[k.replace("grad_full/", "grad_full_spread/")] = np.array(v)
        opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999))
        opt.step()
        out[f"{tag}/param_sum_after_adam"] = np.array([p.double().sum().item() for p in params])
        out[f"{tag}/param_l2_after_adam"] = np.array([p.double().norm().item() for p in params])
        # BN running stats after the step (40 BN layers): checksum
        rs = []
        for pre, net in (("encoder.", m.encoder), ("decoder.", m.decoder)):
            for k, b in net.named_buffers():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    rs.append(b.double().sum().item())
        out[f"{tag}/running_stat_sums"] = np.array(rs)
    save("nelbo_mnist.npz", out)


# ------------------------------------------------------------------------------------------------ G5
def gen_prior():
    pg = R.ref("prior.gaussian")
    out = {}
    for tag, kw, step in (("plain", dict(loss_coeff=0.1), 0), ("anneal", dict(loss_coeff=0.5, annealing_steps=100), 30)):
        prior = pg.GaussianPrior(**kw)
        x = det_input((8, 256, 1, 1), 0.4, 0.8).requires_grad_(True)
        eps = normal((8, 128, 1, 1), seed=7)
        with _FixedEps(eps):
            z, loss, art = prior(x, step=step)
        gz = det_input(tuple(z.shape), 2.2)
        gl = det_input(tuple(loss.shape), 0.1)
        (z * gz).sum().add((loss * gl).sum()).backward()
        out[f"{tag}/x"], out[f"{tag}/eps"], out[f"{tag}/z"], out[f"{tag}/loss"] = npy(x), npy(eps), npy(z), npy(loss)
        out[f"{tag}/gz"], out[f"{tag}/gl"], out[f"{tag}/gx"] = npy(gz), npy(gl), npy(x.grad)
        out[f"{tag}/cfg"] = np.array([kw.get("loss_coeff", 1.0), kw.get("annealing_steps", 0), step], dtype=np.float64)
    # the options of prior/gaussian.py:38-41 (empirical_kl, fixed_var) and the temperature of encode(time=) (fixed_var only)
    for tag, kw, use_time in (("empirical", dict(loss_coeff=0.3, empirical_kl=True), False),
                              ("fixed_var", dict(loss_coeff=0.7, fixed_var=True), False),
                              ("fixed_var_time", dict(loss_coeff=0.7, fixed_var=True), True),
                              ("fixed_var_empirical", dict(loss_coeff=1.5, fixed_var=True, empirical_kl=True), True)):
        prior = pg.GaussianPrior(**kw)
        fixed = kw.get("fixed_var", False)
        x = det_input((8, 128 if fixed else 256, 1, 1), 0.4, 0.8).requires_grad_(True)
        eps = normal((8, 128, 1, 1), seed=8)
        time = (0.2 + 0.1 * torch.arange(8, dtype=torch.float32)) if use_time else None
        with _FixedEps(eps):
            z, loss, art = prior(x, step=0, **({"time": time} if use_time else {}))
        gz = det_input(tuple(z.shape), 2.2)
        gl = det_input(tuple(loss.shape), 0.1)
        (z * gz).sum().add((loss * gl).sum()).backward()
        out[f"{tag}/x"], out[f"{tag}/eps"], out[f"{tag}/z"], out[f"{tag}/loss"] = npy(x), npy(eps), npy(z), npy(loss)
        out[f"{tag}/gz"], out[f"{tag}/gl"], out[f"{tag}/gx"] = npy(gz), npy(gl), npy(x.grad)
        if use_time:
            out[f"{tag}/time"] = npy(time)
        out[f"{tag}/cfg"] = np.array([kw.get("loss_coeff", 1.0), float(kw.get("empirical_kl", False)), float(fixed)], dtype=np.float64)
    save("prior.npz", out)


PRIOR_CORNERS = {
    # name: (class, kwargs, x shape, labels?) -- re-parametrisation dimensions other than 1 and the options ConditionalGaussianPrior inherits
    "g_reparam2": ("GaussianPrior", dict(loss_coeff=0.4, reparam_dim=2), (5, 3, 8), False),
    "g_reparam_last_4d": ("GaussianPrior", dict(loss_coeff=1.0, reparam_dim=-1), (4, 2, 3, 6), False),
    "g_reparam2_empirical": ("GaussianPrior", dict(loss_coeff=0.6, reparam_dim=2, empirical_kl=True), (5, 3, 8), False),
    "c_empirical": ("ConditionalGaussianPrior", dict(dim=(6, 1, 1), num_classes=4, loss_coeff=0.3, empirical_kl=True), (6, 12, 1, 1), True),
    "c_fixed_var": ("ConditionalGaussianPrior", dict(dim=(6, 1, 1), num_classes=4, loss_coeff=0.8, fixed_var=True), (6, 6, 1, 1), True),
    "c_fixed_empirical": ("ConditionalGaussianPrior", dict(dim=(2, 5), num_classes=3, loss_coeff=1.2, fixed_var=True, empirical_kl=True), (6, 2, 5), True),
    "c_reparam2": ("ConditionalGaussianPrior", dict(dim=(3, 4), num_classes=5, loss_coeff=0.5, reparam_dim=2), (6, 3, 8), True),
    "c_reparam2_ema": ("ConditionalGaussianPrior", dict(dim=(3, 4), num_classes=5, loss_coeff=0.5, reparam_dim=2, embedding_ema_decay=0.9), (6, 3, 8), True),
}


def gen_prior_corners():
    """GaussianPrior with `reparam_dim` != 1 and ConditionalGaussianPrior with the options it inherits (empirical_kl, fixed_var,
    reparam_dim; prior/gaussian.py:58-96, prior/conditional_gaussian.py:44-93): z, loss and the gradients of a seeded scalar with
    respect to the input and the class embeddings, `out_size`; for the EMA variant the buffers after the step."""
    pg, pc = R.ref("prior.gaussian"), R.ref("prior.conditional_gaussian")
    out = {}
    for name, (cls, kw, xshape, cond) in PRIOR_CORNERS.items():
        torch.manual_seed(31)
        prior = (pg.GaussianPrior if cls == "GaussianPrior" else pc.ConditionalGaussianPrior)(**kw)
        prior.train()
        x = det_input(xshape, 0.4, 0.8).requires_grad_(True)
        zshape = list(xshape)
        if not kw.get("fixed_var", False):
            zshape[kw.get("reparam_dim", 1)] //= 2
        eps = normal(tuple(zshape), seed=9)
        labels = (torch.arange(xshape[0]) * 2 + 1) % kw["num_classes"] if cond else None
        with _FixedEps(eps):
            z, loss, art = prior(x, step=0, **({"labels": labels} if cond else {}))
        gz, gl = det_input(tuple(z.shape), 2.2), det_input(tuple(loss.shape), 0.1)
        (z * gz).sum().add((loss * gl).sum()).backward()
        out[f"{name}/x"], out[f"{name}/eps"], out[f"{name}/z"], out[f"{name}/loss"] = npy(x), npy(eps), npy(z), npy(loss)
        out[f"{name}/gz"], out[f"{name}/gl"], out[f"{name}/gx"] = npy(gz), npy(gl), npy(x.grad)
        out[f"{name}/out_size"] = np.array(list(prior.out_size(torch.Size(xshape[1:]))))
        if cond:
            out[f"{name}/labels"] = npy(labels)
            for k, v in prior.state_dict().items():
                out[f"{name}/state/{k}"] = npy(v)
            for k, p in prior.named_parameters():
                if p.grad is not None:
                    out[f"{name}/grad/{k}"] = npy(p.grad)
            # the embeddings BEFORE the step (the EMA variant rewrites them): regenerated from the seed by the tests
            torch.manual_seed(31)
            ref0 = pc.ConditionalGaussianPrior(**kw)
            out[f"{name}/init/_mu.weight"], out[f"{name}/init/_log_std.weight"] = npy(ref0._mu.weight), npy(ref0._log_std.weight)
    save("prior_corners.npz", out)


# ------------------------------------------------------------------------------------------------ G6
def gen_sinkhorn():
    w2 = R.ref("ot.w2_utils")
    out = {}

    def problem(lead, n, m, dtype, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(*lead, n, 5, generator=g, dtype=torch.float64)
        y = torch.randn(*lead, m, 5, generator=g, dtype=torch.float64) * 1.2 + 0.3
        C = ((x.unsqueeze(-2) - y.unsqueeze(-3)) ** 2).sum(-1)
        C = C / C.amax(dim=(-2, -1), keepdim=True)
        a = torch.rand(*lead, n, generator=g, dtype=torch.float64) + 0.1
        b = torch.rand(*lead, m, generator=g, dtype=torch.float64) + 0.1
        a, b = a / a.sum(-1, keepdim=True), b / b.sum(-1, keepdim=True)
        return a.to(dtype), b.to(dtype), C.to(dtype)

    cases = [
        ("n7_f64_reg05", (), 7, 7, torch.float64, 0.05, 50, 0.0),
        ("n7x9_f32_reg05", (), 7, 9, torch.float32, 0.05, 50, 0.0),
        ("n64_f32_reg05", (), 64, 64, torch.float32, 0.05, 50, 0.0),
        ("n64_f64_reg05_thr", (), 64, 64, torch.float64, 0.05, 200, 1e-6),
        ("n64_f64_reg1e-2", (), 64, 48, torch.float64, 1e-2, 100, 1e-8),
        ("batch23_f64_thr", (2, 3), 16, 16, torch.float64, 0.05, 300, 1e-5),
        ("batch23_f32", (2, 3), 33, 20, torch.float32, 0.1, 50, 0.0),
        ("n1024_f32_reg05", (), 1024, 1024, torch.float32, 0.05, 50, 0.0),
        ("n1024_f64_reg05", (), 1024, 1024, torch.float64, 0.05, 50, 0.0),
    ]
    for i, (name, lead, n, m, dt, reg, it, thr) in enumerate(cases):
        a, b, C = problem(lead, n, m, dt, 100 + i)
        pi = w2.sinkhorn_log(a, b, C, reg=reg, max_iter=it, threshold=thr)
        out[f"{name}/cfg"] = np.array([reg, it, thr], dtype=np.float64)
        if n <= 64:
            out[f"{name}/a"], out[f"{name}/b"], out[f"{name}/C"], out[f"{name}/pi"] = npy(a), npy(b), npy(C), npy(pi)
        else:
            out[f"{name}/seed"] = np.array([100 + i, n, m])
            out[f"{name}/row_sums"] = npy(pi.sum(-1))
            out[f"{name}/col_sums"] = npy(pi.sum(-2))
            out[f"{name}/pi_corner"] = npy(pi[..., :8, :8])
        out[f"{name}/cost"] = npy((C * pi).sum(dim=(-2, -1)))
    save("sinkhorn.npz", out)


def gen_sinkhorn_autograd():
    """The reference's ``sinkhorn_log`` is plain torch arithmetic: autograd runs through all of its iterations
    (ot/w2_utils.py:301-319).  Recorded here from the REAL function: (i) gradients of a weighted sum of the plan with respect to
    a, b and C (incl. a batched problem that leaves the loop early), (ii) the composition a minibatch-OT loss makes of it --
    C = |z_i - y_j|^2, Cn = C / max C, pi = sinkhorn_log(1/N, 1/M, Cn, 0.05, 50, 0), loss = sum(C * pi) (batch_ot_gmm's read-out,
    :265-269) and loss_n = sum(Cn * pi) -- with dloss/dz, next to the gradient the envelope convention (plan detached) gives."""
    w2 = R.ref("ot.w2_utils")
    out = {}

    def problem(lead, n, m, dtype, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(*lead, n, 5, generator=g, dtype=torch.float64)
        y = torch.randn(*lead, m, 5, generator=g, dtype=torch.float64) * 1.2 + 0.3
        C = ((x.unsqueeze(-2) - y.unsqueeze(-3)) ** 2).sum(-1)
        C = C / C.amax(dim=(-2, -1), keepdim=True)
        a = torch.rand(*lead, n, generator=g, dtype=torch.float64) + 0.1
        b = torch.rand(*lead, m, generator=g, dtype=torch.float64) + 0.1
        a, b = a / a.sum(-1, keepdim=True), b / b.sum(-1, keepdim=True)
        W = torch.randn(*lead, n, m, generator=g, dtype=torch.float64)
        return a.to(dtype), b.to(dtype), C.to(dtype), W.to(dtype)

    direct = [
        ("n7_f64", (), 7, 7, torch.float64, 0.05, 50, 0.0),
        ("n7x9_f32", (), 7, 9, torch.float32, 0.05, 50, 0.0),
        ("n64_f32", (), 64, 64, torch.float32, 0.05, 50, 0.0),
        ("n64_f64", (), 64, 64, torch.float64, 0.05, 50, 0.0),
        ("n48x80_f64_reg01", (), 48, 80, torch.float64, 0.1, 30, 0.0),
        ("batch23_f64_thr", (2, 3), 16, 24, torch.float64, 0.05, 300, 1e-5),
        ("n256_f32", (), 256, 256, torch.float32, 0.05, 50, 0.0),
        ("n256_f64", (), 256, 256, torch.float64, 0.05, 50, 0.0),
    ]
    for i, (name, lead, n, m, dt, reg, it, thr) in enumerate(direct):
        a, b, C, W = problem(lead, n, m, dt, 700 + i)
        a.requires_grad_(True), b.requires_grad_(True), C.requires_grad_(True)
        pi = w2.sinkhorn_log(a, b, C, reg=reg, max_iter=it, threshold=thr)
        (pi * W).sum().backward()
        k = f"direct/{name}"
        out[f"{k}/cfg"] = np.array([reg, it, thr], dtype=np.float64)
        out[f"{k}/seed"] = np.array([700 + i, n, m])   # inputs = problem(lead, n, m, dtype, seed) (tests/test_gpu_sinkhorn_autograd.py)
        keep = (("a", a), ("b", b), ("C", C), ("W", W), ("pi", pi)) if n <= 64 else (("pi_corner", pi[..., :8, :8]),)
        for nm, t in keep + (("ga", a.grad), ("gb", b.grad), ("gC", C.grad)):
            out[f"{k}/{nm}"] = npy(t)

    for i, (n, m, d, dt) in enumerate(((7, 7, 3, torch.float64), (7, 7, 3, torch.float32), (64, 64, 16, torch.float64),
                                       (64, 64, 16, torch.float32), (256, 256, 32, torch.float64), (256, 256, 32, torch.float32))):
        g = torch.Generator().manual_seed(800 + i)
        z0 = (torch.randn(n, d, generator=g, dtype=torch.float64) * 1.3 + 0.2).to(dt)
        y = torch.randn(m, d, generator=g, dtype=torch.float64).to(dt)
        a, b = torch.full((n,), 1.0 / n, dtype=dt), torch.full((m,), 1.0 / m, dtype=dt)
        k = f"prior/n{n}_{'f64' if dt == torch.float64 else 'f32'}"
        res = {}
        for mode in ("full", "envelope"):
            z = z0.clone().requires_grad_(True)
            C = (z ** 2).sum(-1, keepdim=True) + (y ** 2).sum(-1).unsqueeze(-2) - 2 * (z @ y.T)
            Cn = C / C.max()
            if mode == "full":
                pi = w2.sinkhorn_log(a, b, Cn, reg=0.05, max_iter=50, threshold=0.0)
            else:
                with torch.no_grad():
                    pi = w2.sinkhorn_log(a, b, Cn, reg=0.05, max_iter=50, threshold=0.0)
            loss = (C * pi).sum()
            loss.backward()
            res[mode] = (loss.detach(), z.grad.clone())
        z = z0.clone().requires_grad_(True)
        C = (z ** 2).sum(-1, keepdim=True) + (y ** 2).sum(-1).unsqueeze(-2) - 2 * (z @ y.T)
        Cn = C / C.max()
        pi = w2.sinkhorn_log(a, b, Cn, reg=0.05, max_iter=50, threshold=0.0)
        loss_n = (Cn * pi).sum()
        loss_n.backward()
        out[f"{k}/z"], out[f"{k}/y"] = npy(z0), npy(y)
        out[f"{k}/loss"], out[f"{k}/gz_full"], out[f"{k}/gz_envelope"] = npy(res["full"][0]), npy(res["full"][1]), npy(res["envelope"][1])
        out[f"{k}/loss_n"], out[f"{k}/gz_n"] = npy(loss_n), npy(z.grad)
    save("sinkhorn_autograd.npz", out)


# ------------------------------------------------------------------------------------------------ G7
def gen_gaussian_ot():
    mu_ = R.ref("ot.matrix_utils")
    w2 = R.ref("ot.w2_utils")
    gm = R.ref("ot.distribution_models.gaussian_model")
    gt = R.ref("ot.transport.gaussian_transport")
    out = {}
    for D, nb, B in ((8, 3, 64), (32, 2, 100), (128, 2, 256)):
        tag = f"D{D}"
        g = torch.Generator().manual_seed(500 + D)
        A = torch.randn(D, D, generator=g) / math.sqrt(D)
        src = [(torch.randn(B, D, generator=g) @ A * 1.5 + 0.5) for _ in range(nb)]
        tgt = [(torch.randn(B, D, generator=g) * 0.8 - 0.2) for _ in range(nb)]
        for decay in (None, 0.9):
            dtag = f"{tag}/decay{decay}"
            cfg = dict(update_decay=decay, dtype=torch.double)
            op = gt.GaussianTransport(D, source_cfg=cfg, target_cfg=cfg,
                                      transport_cfg=dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True,
                                                         verbose=False, dtype=torch.double))
            op.reset()
            op.source_model._n_obs.zero_()
            for s, t in zip(src, tgt):
                op.update(source_samples=s, target_samples=t)
            sm = op.source_model
            out[f"{dtag}/src_n"] = npy(sm._n_obs)
            out[f"{dtag}/src_sum"] = npy(sm._running_sum)
            out[f"{dtag}/src_sumcov"] = npy(sm._running_sum_cov)
            w = op.compute()
            out[f"{dtag}/w2"] = npy(w)
            out[f"{dtag}/src_mean"] = npy(op.source_model.mean)
            out[f"{dtag}/src_cov"] = npy(op.source_model.cov)
            out[f"{dtag}/tgt_mean"] = npy(op.target_model.mean)
            out[f"{dtag}/tgt_cov"] = npy(op.target_model.cov)
            out[f"{dtag}/T"] = npy(op.transport_operator)
            xin = src[0][:16]
            out[f"{dtag}/transported"] = npy(op.transport(xin))
        out[f"{tag}/src"] = npy(torch.stack(src))
        out[f"{tag}/tgt"] = npy(torch.stack(tgt))
        # plain functions
        n, sx, sxx = float(B), src[0].double().sum(0), src[0].double().T @ src[0].double()
        mean, cov = mu_.mean_cov(sx.clone(), sxx.clone(), torch.tensor(n, dtype=torch.double))
        out[f"{tag}/meancov_mean"], out[f"{tag}/meancov_cov"] = npy(mean), npy(cov)
        _, tcov = mu_.mean_cov(tgt[0].double().sum(0), tgt[0].double().T @ tgt[0].double(), torch.tensor(n, dtype=torch.double))
        tmean = tgt[0].double().mean(0)
        out[f"{tag}/w2_plain"] = npy(w2.w2_gaussian(mean, tmean, cov, tcov, make_pd=True))
        out[f"{tag}/sqrtm_cov"] = npy(mu_.sqrtm(cov))
        out[f"{tag}/invsqrtm_cov"] = npy(mu_.invsqrtm(cov + 1e-8 * torch.eye(D, dtype=torch.double)))
    # batched leading dims + known-answer (tests/test_w2_utils.py:35-41)
    g = torch.Generator().manual_seed(9)
    m1, m2 = torch.randn(2, 3, 3, generator=g), torch.randn(2, 3, 3, generator=g)
    c1 = torch.randn(2, 3, 3, 3, generator=g)
    c1 = c1 @ c1.transpose(-1, -2) + 1e-5 * torch.eye(3)
    c2 = torch.randn(2, 3, 3, 3, generator=g)
    c2 = c2 @ c2.transpose(-1, -2) + 1e-5 * torch.eye(3)
    out["batched/m1"], out["batched/m2"], out["batched/c1"], out["batched/c2"] = npy(m1), npy(m2), npy(c1), npy(c2)
    out["batched/w2"] = npy(w2.w2_gaussian(m1, m2, c1, c2))
    out["batched/w2_self"] = npy(w2.w2_gaussian(m1, m1, c1, c1))
    save("gaussian_ot.npz", out)


# ------------------------------------------------------------------------------------------------ G8
def gen_codebook():
    cb = R.ref("ot.distribution_models.codebook_model")
    out = {}
    for tag, lead, K, d, B in (("flat", (1,), 64, 16, 200), ("multi", (4,), 32, 8, 50)):
        torch.manual_seed(3)
        model = cb.CodebookModel(*lead, d, mixture_cfg=dict(n_components=K, training_mode="argmax", inference_mode="argmax"))
        model.eval()
        g = torch.Generator().manual_seed(77)
        code = torch.randn(*lead, K, d, generator=g)
        with torch.no_grad():
            model.codebook.copy_(code)
        x = torch.randn(*lead, B, d, generator=g) * 1.1
        preds, _, dist = model.predict(x)
        idx = dist.probs.argmax(-1)
        out[f"{tag}/codebook"], out[f"{tag}/x"] = npy(code), npy(x)
        out[f"{tag}/preds"], out[f"{tag}/indices"] = npy(preds), npy(idx)
    save("codebook.npz", out)


def gen_codebook_kmeans():
    """G9: CodebookModel.update x 6 (streaming k-means, with and without EMA decay) -> fit -> predict /
    distribution / w2, driven through the reference's own class in 'argmax' mode."""
    cb = R.ref("ot.distribution_models.codebook_model")
    out = {}
    for tag, lead, K, d, B, decay in (("sum", (2,), 6, 4, 64, None), ("ema", (1,), 10, 3, 128, 0.9)):
        torch.manual_seed(5)
        model = cb.CodebookModel(*lead, d, update_decay=decay,
                                 mixture_cfg=dict(n_components=K, training_mode="argmax", inference_mode="argmax"))
        model.train()
        g = torch.Generator().manual_seed(11)
        centres = torch.randn(*lead, K, d, generator=g) * 3.0
        out[f"{tag}/cfg"] = np.array([K, d, B, -1.0 if decay is None else decay])
        out[f"{tag}/vec_init"], out[f"{tag}/mat_init"] = npy(model.vec_init), npy(model.mat_init)
        batches = []
        for step in range(6):
            which = torch.randint(0, K, (*lead, B), generator=g)
            x = torch.gather(centres, -2, which.unsqueeze(-1).expand(*lead, B, d)) + 0.3 * torch.randn(*lead, B, d, generator=g)
            batches.append(x)
            if step == 0:
                torch.manual_seed(1234)  # the randperm of _init_parameters draws from the host generator
            model.update(x)
            out[f"{tag}/step{step}/codebook"] = npy(model.codebook).copy()  # buffers are updated in place: snapshot
            out[f"{tag}/step{step}/n_obs"] = npy(model._n_obs).copy()
            out[f"{tag}/step{step}/running_sum"] = npy(model._running_sum).copy()
        out[f"{tag}/batches"] = npy(torch.stack(batches))
        model.fit()
        out[f"{tag}/fit/codebook"] = npy(model.codebook)
        model.eval()
        probe = batches[-1]
        preds, _, dist = model.predict(probe)
        out[f"{tag}/predict/preds"], out[f"{tag}/predict/probs"] = npy(preds), npy(dist.probs)
        out[f"{tag}/predict/entropy"] = npy(dist.entropy())
        out[f"{tag}/weights"] = npy(model.weights)
        other = cb.CategoricalEmbeddings(centres, probs=torch.ones(*lead, K) / K)
        out[f"{tag}/centres"] = npy(centres)
        out[f"{tag}/w2"] = npy(model.w2(other))
    save("codebook_kmeans.npz", out)


def _clusters(g, lead, K, d, B, centres, noise=0.3):
    which = torch.randint(0, K, (*lead, B), generator=g)
    return torch.gather(centres, -2, which.unsqueeze(-1).expand(*lead, B, d)) + noise * torch.randn(*lead, B, d, generator=g)


def gen_discrete():
    """G10: the soft 'mean' mode of CodebookModel (update / predict), DiscreteTransport.compute / transport
    for transport_type 'mean' and 'argmax' (reference tests/test_latent_transport.py:92-101 uses training_mode 'mean',
    temperature 1e-2), and CodebookPrior.forward (one-hot mode, 'l2' loss: values, straight-through gradient, codebook
    after the step; evaluation-mode 'kl' loss)."""
    cb = R.ref("ot.distribution_models.codebook_model")
    dt = R.ref("ot.transport.discrete_transport")
    pr = R.ref("prior.codebook")
    out = {}
    # ---- (a) soft k-means
    lead, K, d, B = (2,), 5, 3, 48
    g = torch.Generator().manual_seed(21)
    centres = torch.randn(*lead, K, d, generator=g) * 2.0
    model = cb.CodebookModel(*lead, d, mixture_cfg=dict(n_components=K, training_mode="mean", inference_mode="mean",
                                                       temperature=0.5))
    model.train()
    xs = [_clusters(g, lead, K, d, B, centres) for _ in range(3)]
    out["mean/batches"], out["mean/cfg"] = npy(torch.stack(xs)), np.array([K, d, B, 0.5])
    for i, x in enumerate(xs):
        if i == 0:
            torch.manual_seed(77)
        model.update(x)
        out[f"mean/step{i}/codebook"], out[f"mean/step{i}/n_obs"] = npy(model.codebook).copy(), npy(model._n_obs).copy()
    model.eval()
    preds, _, dist = model.predict(xs[-1])
    out["mean/predict/preds"], out["mean/predict/probs"] = npy(preds), npy(dist.probs)
    # ---- (b) DiscreteTransport
    for ttype in ("mean", "argmax"):
        K, d, B = 8, 3, 96
        g = torch.Generator().manual_seed(31)
        cs, ct = torch.randn(K, d, generator=g) * 2.0, torch.randn(K, d, generator=g) * 2.0 + 1.0
        mix = dict(n_components=K, training_mode="mean", inference_mode="argmax", temperature=1e-2)
        op = dt.DiscreteTransport(d, source_cfg=dict(mixture_cfg=mix), target_cfg=dict(mixture_cfg=mix), transport_type=ttype,
                                  sinkhorn_reg=1e-2, sinkhorn_max_iter=200, sinkhorn_threshold=1e-9)
        op.train()
        src = [_clusters(g, (), K, d, B, cs) for _ in range(3)]
        tgt = [_clusters(g, (), K, d, B, ct) for _ in range(3)]
        for i, (a, b) in enumerate(zip(src, tgt)):
            # the first update of each model draws its initial atoms with torch.randperm from the host generator (and the
            # soft assignment then samples indices from it): seed each first call separately
            if i == 0:
                torch.manual_seed(78)
            op.update(source_samples=a)
            if i == 0:
                torch.manual_seed(178)
            op.update(target_samples=b)
        cost = op.compute()
        probe = _clusters(g, (), K, d, 40, cs)
        out[f"dt_{ttype}/src"], out[f"dt_{ttype}/tgt"] = npy(torch.stack(src)), npy(torch.stack(tgt))
        out[f"dt_{ttype}/source_codebook"], out[f"dt_{ttype}/target_codebook"] = npy(op.source_model.codebook), npy(op.target_model.codebook)
        out[f"dt_{ttype}/source_probs"], out[f"dt_{ttype}/target_probs"] = npy(op.source_distribution.probs), npy(op.target_distribution.probs)
        out[f"dt_{ttype}/cost"], out[f"dt_{ttype}/plan"] = npy(cost), npy(op.transport_matrix)
        out[f"dt_{ttype}/probe"], out[f"dt_{ttype}/moved"] = npy(probe), npy(op.transport(probe))
    # ---- (c) CodebookPrior
    # (the reference's first update copies samples[..., randperm(B)[:K], :] into the single shared codebook, which only
    # fits when the latent is embedded as a whole -- one position -- and B >= K)
    size, K, Bp = (6, 2, 2), 8, 16
    prior = pr.CodebookPrior(size, (1, 2, 3), loss="l2", loss_coeff=0.5, annealing_steps=10,
                             mixture_cfg=dict(n_components=K, training_mode="argmax", inference_mode="argmax"))
    prior.train()
    g = torch.Generator().manual_seed(41)
    w = torch.randn(Bp, *size, generator=g)
    for step in range(2):
        x = (torch.randn(Bp, *size, generator=g) * 1.5).requires_grad_(True)
        if step == 0:
            torch.manual_seed(79)
        z, loss, art = prior(x, step=3 + step)
        ((z * w).sum() + loss.sum()).backward()
        out[f"prior/step{step}/x"], out[f"prior/step{step}/z"], out[f"prior/step{step}/loss"] = npy(x), npy(z), npy(loss)
        out[f"prior/step{step}/gx"], out[f"prior/step{step}/probs"] = npy(x.grad), npy(art["distribution"].probs)
        out[f"prior/step{step}/codebook"] = npy(prior.codebook_model.codebook).copy()
    out["prior/w"] = npy(w)
    prior.eval()
    prior.loss = "kl"
    x = torch.randn(Bp, *size, generator=g)
    z, loss, art = prior(x, step=100)
    out["prior/eval/x"], out["prior/eval/z"], out["prior/eval/loss_kl"] = npy(x), npy(z), npy(loss)
    prior.loss = "first_kl"
    out["prior/eval/loss_first_kl"] = npy(prior(x, step=100)[1])
    save("discrete.npz", out)


def gen_gmm():
    """G11: GaussianMixtureModel with diagonal covariances (update x 3 -> fit -> energy / assign / w2) and
    GMMTransport.compute / transport ('argmax'), configured as the reference's tests/test_latent_transport.py:80-91
    (diag=True, argmax modes, double precision)."""
    gm = R.ref("ot.distribution_models.gassian_mixture_model")
    gt = R.ref("ot.transport.gmm_transport")
    out = {}
    w2_cfg = dict(diag=True, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
    mix = dict(metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax", inference_mode="argmax")
    # ---- the model alone, with a leading dimension and an EMA variant
    for tag, lead, K, d, B, decay in (("sum", (2,), 4, 3, 64, None), ("ema", (), 5, 2, 96, 0.8)):
        g = torch.Generator().manual_seed(51)
        centres = torch.randn(*lead, K, d, generator=g, dtype=torch.double) * 3.0
        model = gm.GaussianMixtureModel(*lead, d, mixture_cfg={**mix, "n_components": K}, w2_cfg=w2_cfg, update_decay=decay,
                                        dtype=torch.double)
        model.train()
        out[f"{tag}/cfg"] = np.array([K, d, B, -1.0 if decay is None else decay])
        out[f"{tag}/vec_init"] = npy(model.vec_init)
        xs = [_clusters(g, lead, K, d, B, centres.float(), noise=0.4).double() for _ in range(3)]
        out[f"{tag}/batches"] = npy(torch.stack(xs))
        for i, x in enumerate(xs):
            if i == 0:
                torch.manual_seed(81)
            model.update(x)
            out[f"{tag}/step{i}/mean"], out[f"{tag}/step{i}/cov"] = npy(model.mean).copy(), npy(model.cov).copy()
            out[f"{tag}/step{i}/weights"], out[f"{tag}/step{i}/n_obs"] = npy(model.weights).copy(), npy(model._n_obs).copy()
        model.fit()
        out[f"{tag}/fit/mean"], out[f"{tag}/fit/cov"], out[f"{tag}/fit/weights"] = npy(model.mean), npy(model.cov), npy(model.weights)
        model.eval()
        energy = model.energy(xs[-1])
        weights, _, dist = model.assign(xs[-1])
        out[f"{tag}/energy"], out[f"{tag}/assign_onehot"], out[f"{tag}/assign_probs"] = npy(energy), npy(weights), npy(dist.probs)
        other = torch.distributions.MixtureSameFamily(
            torch.distributions.Categorical(torch.ones(*lead, K, dtype=torch.double) / K),
            torch.distributions.Independent(torch.distributions.Normal(centres, torch.full_like(centres, 0.4)), 1))
        out[f"{tag}/centres"] = npy(centres)
        out[f"{tag}/w2"] = npy(model.w2(other))
    # ---- the transport operator
    K, d, B = 6, 4, 128
    g = torch.Generator().manual_seed(61)
    cs, ct = torch.randn(K, d, generator=g) * 2.5, torch.randn(K, d, generator=g) * 2.5 + 1.5
    cfg = dict(update_decay=None, update_with_autograd=False, dtype=torch.double, mixture_cfg={**mix, "n_components": K})
    op = gt.GMMTransport(d, transport_type="argmax", transport_cfg=w2_cfg, source_cfg=cfg, target_cfg=cfg)
    op.train()
    src = [_clusters(g, (), K, d, B, cs, noise=0.5).double() for _ in range(3)]
    tgt = [(_clusters(g, (), K, d, B, ct, noise=0.3) * torch.tensor([1.0, 0.5, 2.0, 1.0])).double() for _ in range(3)]
    for i, (a, b) in enumerate(zip(src, tgt)):
        if i == 0:
            torch.manual_seed(82)
        op.update(source_samples=a)
        if i == 0:
            torch.manual_seed(182)
        op.update(target_samples=b)
    total = op.compute()
    op.eval()
    probe = _clusters(g, (), K, d, 48, cs, noise=0.5)
    out["tr/src"], out["tr/tgt"], out["tr/probe"] = npy(torch.stack(src)), npy(torch.stack(tgt)), npy(probe)
    for side, m in (("source", op.source_model), ("target", op.target_model)):
        out[f"tr/{side}_mean"], out[f"tr/{side}_cov"], out[f"tr/{side}_weights"] = npy(m.mean), npy(m.cov), npy(m.weights)
    out["tr/total"], out["tr/coupling"] = npy(total), npy(op.transport_matrix)
    out["tr/moved"] = npy(op.transport(probe))
    save("gmm.npz", out)


def gen_gmm_autograd():
    """GaussianMixtureModel: (1) `model(samples)` of a fitted model in eval mode (= GaussianModel.predict through the class's MRO:
    the mixture log-density); (2) `update_with_autograd=True` (gassian_mixture_model.py:53-58, gaussian_model.py:52-93): the
    log-density and its gradients with respect to the samples, the means, the raw ExpScaleTril parameter and the raw soft-max
    weight parameter, diagonal and full covariances, with and without a leading dimension."""
    gm = R.ref("ot.distribution_models.gassian_mixture_model")
    out = {}
    mix = dict(metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax", inference_mode="argmax")
    # (1)
    for tag, diag, lead, K, d, B in (("fit_diag", True, (2,), 4, 3, 64), ("fit_full", False, (), 3, 4, 80)):
        g = torch.Generator().manual_seed(151)
        w2_cfg = dict(diag=diag, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
        centres = torch.randn(*lead, K, d, generator=g, dtype=torch.double) * 3.0
        model = gm.GaussianMixtureModel(*lead, d, mixture_cfg={**mix, "n_components": K}, w2_cfg=w2_cfg, update_decay=None, dtype=torch.double)
        model.train()
        xs = [_clusters(g, lead, K, d, B, centres.float(), noise=0.4).double() for _ in range(2)]
        for i, x in enumerate(xs):
            if i == 0:
                torch.manual_seed(83)
            model.update(x)
        model.fit()
        model.eval()
        out[f"{tag}/cfg"] = np.array([K, d, B, int(diag)])
        out[f"{tag}/mean"], out[f"{tag}/cov"], out[f"{tag}/weights"] = npy(model.mean), npy(model.cov), npy(model.weights)
        # what is stored behind the parametrisations (reading `cov` adds the positive-definite shift, `_weights` normalises)
        out[f"{tag}/raw_cov"] = npy(model.parametrizations.cov.original)
        out[f"{tag}/raw_weights"] = npy(model.parametrizations._weights.original)
        out[f"{tag}/probe"] = npy(xs[-1])
        out[f"{tag}/log_prob"] = npy(model(xs[-1]))
    # (2)
    for tag, diag, lead, K, d, B in (("auto_diag", True, (), 3, 5, 40), ("auto_diag_lead", True, (2,), 4, 3, 24),
                                     ("auto_full", False, (), 3, 4, 32), ("auto_full_lead", False, (2,), 2, 6, 20)):
        g = torch.Generator().manual_seed(161 + d)
        w2_cfg = dict(diag=diag, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
        model = gm.GaussianMixtureModel(*lead, d, mixture_cfg={**mix, "n_components": K}, w2_cfg=w2_cfg, update_with_autograd=True,
                                        dtype=torch.double)
        raw_cov = model.parametrizations.cov.original
        raw_w = model.parametrizations._weights.original
        with torch.no_grad():
            model.mean.copy_(torch.randn(model.mean.shape, generator=g, dtype=torch.double) * 1.5)
            raw_cov.copy_(torch.randn(raw_cov.shape, generator=g, dtype=torch.double) * 0.3)
            raw_w.copy_(torch.randn(raw_w.shape, generator=g, dtype=torch.double))
        x = (torch.randn(*lead, B, d, generator=g, dtype=torch.double) * 2.0).requires_grad_(True)
        seed = torch.randn(*lead, B, generator=g, dtype=torch.double)
        model.train()
        lp = model(x)
        (lp * seed).sum().backward()
        out[f"{tag}/cfg"] = np.array([K, d, B, int(diag)])
        out[f"{tag}/mean"], out[f"{tag}/raw_cov"], out[f"{tag}/raw_weights"] = npy(model.mean), npy(raw_cov), npy(raw_w)
        out[f"{tag}/x"], out[f"{tag}/seed"], out[f"{tag}/log_prob"] = npy(x), npy(seed), npy(lp)
        out[f"{tag}/g_x"], out[f"{tag}/g_mean"] = npy(x.grad), npy(model.mean.grad)
        out[f"{tag}/g_raw_cov"], out[f"{tag}/g_raw_weights"] = npy(raw_cov.grad), npy(raw_w.grad)
        out[f"{tag}/variances"], out[f"{tag}/weights"] = npy(model.variances), npy(model.weights)
    save("gmm_autograd.npz", out)


def gen_codebook_autograd():
    """CodebookModel(update_with_autograd=True) (codebook_model.py:89-93): `model(x)` = (weights @ codebook, indices, distribution)
    with the codebook a trained parameter; gradients of a scalar made of the predictions, the assignment probabilities and their
    entropy with respect to the samples and the codebook, in the soft 'mean' mode and the one-hot 'argmax' mode; and the same
    through CodebookPrior (loss='kl', soft mode: commitment term included) on a [B, 8, 2, 2] latent."""
    cb = R.ref("ot.distribution_models.codebook_model")
    out = {}
    for tag, mode, lead, K, d, B, T in (("mean", "mean", (), 6, 4, 32, 0.7), ("mean_lead", "mean", (2,), 5, 3, 24, 1.0),
                                        ("argmax", "argmax", (), 6, 4, 32, 1.0)):
        g = torch.Generator().manual_seed(171 + K)
        mix = dict(n_components=K, metric="euclidean", p=2., topk=None, temperature=T, training_mode=mode, inference_mode=mode)
        model = cb.CodebookModel(*lead, d, mixture_cfg=mix, update_with_autograd=True)
        with torch.no_grad():
            model.codebook.copy_(torch.randn(model.codebook.shape, generator=g) * 1.5)
        x = (torch.randn(*lead, B, d, generator=g) * 1.5).requires_grad_(True)
        s_pred = torch.randn(*lead, B, d, generator=g)
        s_prob = torch.randn(*lead, B, K, generator=g)
        s_ent = torch.randn(*lead, B, generator=g)
        model.train()
        torch.manual_seed(5)
        preds, _, dist = model(x)
        scalar = (preds * s_pred).sum() + (dist.probs * s_prob).sum() + (dist.entropy() * s_ent).sum()
        scalar.backward()
        out[f"{tag}/cfg"] = np.array([K, d, B, T])
        out[f"{tag}/codebook"], out[f"{tag}/x"] = npy(model.codebook), npy(x)
        out[f"{tag}/s_pred"], out[f"{tag}/s_prob"], out[f"{tag}/s_ent"] = npy(s_pred), npy(s_prob), npy(s_ent)
        out[f"{tag}/preds"], out[f"{tag}/probs"], out[f"{tag}/entropy"] = npy(preds), npy(dist.probs), npy(dist.entropy())
        out[f"{tag}/g_x"], out[f"{tag}/g_codebook"] = npy(x.grad), npy(model.codebook.grad)
    # through the prior
    pr = R.ref("prior.codebook")
    g = torch.Generator().manual_seed(181)
    K = 7
    mix = dict(n_components=K, metric="euclidean", p=2., topk=None, temperature=0.8, training_mode="mean", inference_mode="mean")
    prior = pr.CodebookPrior((8, 2, 2), embed_dims=(1,), loss="kl", loss_coeff=1.0, mixture_cfg=mix, update_with_autograd=True)
    with torch.no_grad():
        prior.codebook_model.codebook.copy_(torch.randn(prior.codebook_model.codebook.shape, generator=g))
    z = torch.randn(6, 8, 2, 2, generator=g).requires_grad_(True)
    s_z = torch.randn(6, 8, 2, 2, generator=g)
    prior.train()
    torch.manual_seed(6)
    enc, loss, _ = prior.encode(z)
    ((enc * s_z).sum() + loss.sum()).backward()
    out["prior/cfg"] = np.array([K, 0.8])
    out["prior/codebook"], out["prior/z"], out["prior/s_z"] = npy(prior.codebook_model.codebook), npy(z), npy(s_z)
    out["prior/enc"], out["prior/loss"] = npy(enc), npy(loss)
    out["prior/g_z"], out["prior/g_codebook"] = npy(z.grad), npy(prior.codebook_model.codebook.grad)
    save("codebook_autograd.npz", out)


class _RecordingOperator(torch.nn.Module):
    """stands in for a TransportOperator: records what the callback hands it"""

    def __init__(self, *size, **kwargs):
        super().__init__()
        self.size, self.calls = size, []

    def update(self, source_samples=None, target_samples=None):
        self.calls.append((1, source_samples) if source_samples is not None else (0, target_samples))

    def reset(self):
        self.calls.append((2, None))

    def compute(self):
        return torch.tensor([1.0, 3.0])

    def forward(self, x):
        return 2 * x


class _RoutingModule:
    training = True
    device = torch.device("cpu")

    def encode(self, x, **kw):
        return x[:, :2, ::2, ::2] * 10

    def decode(self, z, **kw):
        return z

    def eval(self):
        self.training = False

    def train(self):
        self.training = True

    def log(self, *a, **k):
        pass


ROUTING_GRID = [(s_, t_, u_, c_, dims, False, None) for s_ in (False, True) for t_ in (False, True) for u_ in (False, True)
                for c_ in (False, True) for dims in ((1,), (1, 2, 3))]
# + a verbose callback (its encode fallback for step outputs without latents, batch 0 only) and class filtering (`class_idx`, the
#   condition under the step output's key 'y')
ROUTING_GRID += [(False, True, False, False, (1,), True, None), (True, True, True, True, (1,), True, None),
                 (False, False, True, False, (1,), False, 1), (True, True, False, True, (2, 3), False, 0)]


def gen_latent_transport_routing():
    """The routing logic of the LatentTransport callback (ot/transport_callback.py:173-237, layouts :289-330) driven through its own
    hooks with a recording operator: for every combination of source_latents_from_train / target_latents_from_train / unpaired /
    common_operator and two `transport_dims`, the sequence of operator calls (reset / source / target and the tensors handed over)
    during four training and four validation batches -- even batches carry `latents` in the step output, odd ones only `samples` --
    and what `transport()` returns."""
    tc = R.ref("ot.transport_callback")
    out = {}
    x = [torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4) + 1000.0 * i for i in range(8)]
    labels = torch.tensor([1, 0])
    for idx, (src, tgt, unp, common, dims, verbose, cls) in enumerate(ROUTING_GRID):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cb = tc.LatentTransport(size=(2, 2, 2), transport_dims=dims, transport_operator=_RecordingOperator,
                                    transformations=lambda t: t + 100.0, logging_prefix="t", target_latents_from_train=tgt,
                                    source_latents_from_train=src, unpaired=unp, common_operator=common, verbose=verbose, class_idx=cls)
        mod = _RoutingModule()
        cb.on_validation_epoch_start(None, mod)
        for b in range(4):
            o = {"samples": x[b], "kwargs": {}, "y": labels}
            if b % 2 == 0:
                o["latents"] = mod.encode(x[b]) + 1.0
            cb.on_train_batch_end(None, mod, o, None, b)
        for b in range(4):
            o = {"samples": x[4 + b], "kwargs": {}, "y": labels}
            if b % 2 == 0:
                o["latents"] = mod.encode(x[4 + b]) + 1.0
            cb.on_validation_batch_end(None, mod, o, None, b, 0)
        calls = cb.transport_operator.calls
        out[f"{idx}/flags"] = np.array([int(src), int(tgt), int(unp), int(common), len(dims), int(verbose), -1 if cls is None else cls])
        out[f"{idx}/kinds"] = np.array([k for k, _ in calls])
        for j, (_, t) in enumerate(calls):
            if t is not None:
                out[f"{idx}/call{j}"] = npy(t)
        out[f"{idx}/transported"] = npy(cb.transport(mod.encode(x[0])))
        out[f"{idx}/op_size"] = np.array(cb.transport_operator.size)
    save("latent_transport_routing.npz", out)


def _layout_cases():
    import itertools
    cases = []
    for shape in ((2, 2, 3, 4), (3, 2, 3, 2, 2), (2, 5)):
        dims = list(range(1, len(shape)))
        for r in range(1, len(dims) + 1):
            for perm in itertools.permutations(dims, r):
                if len(perm) > 2 and list(perm) != sorted(perm) and perm != tuple(reversed(sorted(perm))):
                    continue   # (all orders of one and two axes, sorted / reversed orders of longer ones)
                for bf in (True, False):
                    for fb in (True, False):
                        cases.append((shape, perm, bf, fb))
    return cases


def gen_layouts():
    """utils.permute_and_flatten / unflatten_and_unpermute (utils/__init__.py:233-311) -- the [B, C, H, W] <-> [positions, B, dim] /
    [B * positions, dim] layouts of LatentTransport and CodebookPrior -- over every choice of axes (in any order), batch_first and
    flatten_batch for 2-, 4- and 5-dimensional inputs: the rearranged tensor and the round trip."""
    U = R.ref("utils")
    out = {}
    cases = _layout_cases()
    for i, (shape, perm, bf, fb) in enumerate(cases):
        x = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
        y = U.permute_and_flatten(x, perm, batch_first=bf, flatten_batch=fb)
        back = U.unflatten_and_unpermute(y, x.shape, perm, batch_first=bf, flatten_batch=fb)
        out[f"{i}/cfg"] = np.array(list(shape) + [-1] + list(perm) + [-1, int(bf), int(fb)])
        out[f"{i}/y"] = npy(y)
        out[f"{i}/roundtrip_ok"] = np.array([int(torch.equal(back, x))])
    out["n"] = np.array([len(cases)])
    save("layouts.npz", out)


def gen_cnn_shapes():
    """The shape inference of the CNN constructor (networks/cnn.py:605-672): get_block_scaling, get_channel_list and div_sqrt over a grid
    of (in_features, out_features, in_resolution, out_resolution, scaling factor, capacity) / all n <= 600."""
    cnn = R.ref("networks.cnn")
    rows = []
    for cin in (1, 3, 8, 24):
        for cout in (8, 16, 64, 256, 300):
            for rin in (4, 8, 16, 32, 64, 128):
                for rout in (1, 2, 4, 8):
                    if rout >= rin:
                        continue
                    for sf in (2, 4, 8):
                        for cap in (4, 8, 16, 24):
                            feats, res = cnn.get_channel_list(cin, cout, rin, rout, sf, cap)
                            rows.append([cin, cout, rin, rout, sf, cap, len(feats)] + list(feats) + list(res))
    width = max(len(r) for r in rows)
    out = {"channel_list": np.array([r + [-1] * (width - len(r)) for r in rows], dtype=np.int64)}
    out["div_sqrt"] = np.array([int(cnn.div_sqrt(n)) for n in range(1, 601)], dtype=np.int64)
    sc = []
    for hi in (2, 4, 8, 16, 32, 64, 128, 256):
        for lo in (1, 2, 4, 8, 16):
            if lo >= hi:
                continue
            for m in (2, 4, 8, 16):
                v = cnn.get_block_scaling(hi, lo, m)
                sc.append([hi, lo, m, len(v)] + list(v))
    width = max(len(r) for r in sc)
    out["block_scaling"] = np.array([r + [-1] * (width - len(r)) for r in sc], dtype=np.int64)
    save("cnn_shapes.npz", out)


def gen_utils_small():
    """The small tensor helpers of utils/__init__.py:123-218 the hot path calls: replicate_batch / mean_replicated_batch /
    std_replicated_batch (VAE expansion), ema / ema_inplace (running statistics), laplace_smoothing (mixture weights)."""
    U = R.ref("utils")
    out = {}
    g = torch.Generator().manual_seed(201)
    x = torch.randn(4, 3, 2, generator=g)
    for n in (0, 1, 2, 3):
        r = U.replicate_batch({"a": x, "b": [x[:, 0], 7]}, n)
        out[f"replicate{n}/a"], out[f"replicate{n}/b0"] = npy(r["a"]), npy(r["b"][0])
        e = torch.randn(max(n, 1) * 4, 3, generator=g)
        out[f"reduce{n}/in"], out[f"reduce{n}/mean"] = npy(e), npy(U.mean_replicated_batch(e, n))
        if n > 1:
            out[f"reduce{n}/std"] = npy(U.std_replicated_batch(e, n))
    avg, new = torch.randn(5, 3, generator=g), torch.randn(5, 3, generator=g)
    for tag, decay in (("none", None), ("d0", 0.0), ("d09", 0.9), ("d1", 1.0)):
        out[f"ema/{tag}"] = npy(U.ema(avg.clone(), new, decay))
        inp = avg.clone()
        U.ema_inplace(inp, new, decay)
        out[f"ema_inplace/{tag}"] = npy(inp)
    out["ema/avg"], out["ema/new"] = npy(avg), npy(new)
    cnt = torch.tensor([[0.0, 3.0, 1.0, 0.0], [2.0, 2.0, 2.0, 2.0], [0.0, 0.0, 0.0, 0.0]])
    out["laplace/in"] = npy(cnt)
    out["laplace/eps1e-5"] = npy(U.laplace_smoothing(cnt, 4))
    out["laplace/eps0.5"] = npy(U.laplace_smoothing(cnt, 4, eps=0.5))
    out["laplace/none"] = npy(U.laplace_smoothing(cnt, 4, eps=None))
    save("utils_small.npz", out)


CODEBOOK_OPTION_CASES = [
    # tag, metric, p, topk, mode, temperature
    ("cos_p2_mean", "cosine", 2.0, None, "mean", 0.5), ("cos_p1_argmax", "cosine", 1.0, None, "argmax", 1.0),
    ("cos_p05_mean_top3", "cosine", 0.5, 3, "mean", 0.7), ("euc_p1_mean", "euclidean", 1.0, None, "mean", 0.6),
    ("euc_p05_argmax", "euclidean", 0.5, None, "argmax", 1.0), ("euc_p3_mean_top2", "euclidean", 3.0, 2, "mean", 0.8),
    ("euc_p2_mean_top3", "euclidean", 2.0, 3, "mean", 0.5), ("euc_p2_argmax_top1", "euclidean", 2.0, 1, "argmax", 1.0),
    ("cos_p2_sample_top1", "cosine", 2.0, 1, "sample", 1.0),
]


def gen_codebook_options():
    """CodebookModel with `metric='cosine'`, p != 2 and `topk` (base.py:166-235, codebook_model.py:150-168), the codebook a trained
    parameter so that both gradients show: energies, predictions, assignment probabilities and the gradients of a seeded scalar of
    (predictions, probabilities) with respect to the samples and the codebook; GaussianMixtureModel with `topk` (assignment weights)."""
    cb = R.ref("ot.distribution_models.codebook_model")
    out = {}
    K, d, B = 6, 4, 24
    for tag, metric, p, topk, mode, T in CODEBOOK_OPTION_CASES:
        g = torch.Generator().manual_seed(211)
        mix = dict(n_components=K, metric=metric, p=p, topk=topk, temperature=T, training_mode=mode, inference_mode=mode)
        model = cb.CodebookModel(2, d, mixture_cfg=mix, update_with_autograd=True)
        with torch.no_grad():
            model.codebook.copy_(torch.randn(model.codebook.shape, generator=g) * 1.2 + 0.3)
        x = (torch.randn(2, B, d, generator=g) * 1.5 + 0.2).requires_grad_(True)
        s_pred, s_prob = torch.randn(2, B, d, generator=g), torch.randn(2, B, K, generator=g)
        model.train()
        torch.manual_seed(7)
        energy = model.energy(x)
        preds, _, dist = model(x)
        ((preds * s_pred).sum() + (dist.probs * s_prob).sum()).backward()
        out[f"{tag}/codebook"], out[f"{tag}/x"], out[f"{tag}/s_pred"], out[f"{tag}/s_prob"] = npy(model.codebook), npy(x), npy(s_pred), npy(s_prob)
        out[f"{tag}/energy"], out[f"{tag}/preds"], out[f"{tag}/probs"] = npy(energy), npy(preds), npy(dist.probs)
        out[f"{tag}/g_x"], out[f"{tag}/g_codebook"] = npy(x.grad), npy(model.codebook.grad)
    # a Gaussian mixture with topk
    gm = R.ref("ot.distribution_models.gassian_mixture_model")
    for tag, topk, mode in (("gmm_top2_mean", 2, "mean"), ("gmm_top1_sample", 1, "sample"), ("gmm_top3_argmax", 3, "argmax")):
        g = torch.Generator().manual_seed(221)
        mix = dict(n_components=5, topk=topk, temperature=0.9, training_mode=mode, inference_mode=mode)
        model = gm.GaussianMixtureModel(3, mixture_cfg=mix, w2_cfg=dict(diag=True, make_pd=True, dtype=torch.double), dtype=torch.double)
        with torch.no_grad():
            model.mean.copy_(torch.randn(model.mean.shape, generator=g, dtype=torch.double) * 2.0)
            model.cov = torch.rand(model.cov.shape, generator=g, dtype=torch.double) + 0.5
        x = torch.randn(20, 3, generator=g, dtype=torch.double) * 2.0
        model.eval()
        torch.manual_seed(8)
        w, _, dist = model.assign(x)
        out[f"{tag}/mean"], out[f"{tag}/cov"], out[f"{tag}/x"] = npy(model.mean), npy(model.parametrizations.cov.original), npy(x)
        out[f"{tag}/weights"], out[f"{tag}/probs"] = npy(w), npy(dist.probs)
    save("codebook_options.npz", out)


CKPT_KEYS = ["encoder.0.block.0.weight", "encoder.0.block.0.bias", "encoder.0.skip.weight", "encoder.1.block.0.weight",
             "decoder.0.block.0.weight", "decoder.encoder.weight", "prior._mu.weight", "encoder_extra.weight"]
CKPT_CASES = [(None, ""), ("encoder", ""), ("encoder", "enc."), ("encoder.0", ""), ("encoder.0.block", "b."), ("decoder", ""),
              ("prior", ""), ("enc", ""), ("missing", ""), ("encoder_extra", ""), ("decoder.encoder", "x.")]


def gen_partial_checkpoint():
    """PartialCheckpoint.state_dict's key selection (utils/partial_checkpoint.py:55-66) and human_format (:10-21): which keys of a
    checkpoint are taken for an attribute name, and what they are renamed to."""
    import tempfile
    pcm = R.ref("utils.partial_checkpoint")
    out = {}
    sd = {k: torch.full((1,), float(i)) for i, k in enumerate(CKPT_KEYS)}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "c.ckpt")
        torch.save({"state_dict": sd}, path)
        for i, (attr, rep) in enumerate(CKPT_CASES):
            got = pcm.PartialCheckpoint(path, attr_name=attr, replace_str=rep).state_dict
            out[f"case{i}/keys"] = np.array(list(got.keys()) or [""])
            out[f"case{i}/values"] = np.array([float(v) for v in got.values()] or [-1.0])
    nums = [0, 1, 12, 999, 1000, 1234, 99950, 1234567, 1e9, 2.5e12, 123456789012345.0, -4321]
    out["human/nums"] = np.array(nums, dtype=np.float64)
    out["human/text"] = np.array([pcm.human_format(n) for n in nums])
    save("partial_checkpoint.npz", out)


def _matrix_zoo():
    """small symmetric / asymmetric matrices with the structure the validators of ot/matrix_utils.py branch on"""
    g = torch.Generator().manual_seed(301)
    a = torch.randn(5, 5, generator=g, dtype=torch.double)
    pd = a @ a.T + 0.5 * torch.eye(5, dtype=torch.double)
    low = a[:, :2] @ a[:, :2].T                                              # rank 2: positive SEMI-definite
    indef = a + a.T
    asym = pd + 1e-3 * torch.triu(torch.ones(5, 5, dtype=torch.double), 1)   # symmetric up to 1e-3
    nearly = pd + 1e-6 * torch.triu(torch.ones(5, 5, dtype=torch.double), 1)  # ... up to 1e-6 (inside the 1e-8 sum-of-squares test)
    neg = -pd
    batch = torch.stack([pd, low, indef, neg])
    return dict(pd=pd, low=low, indef=indef, asym=asym, nearly=nearly, neg=neg, batch=batch, diag_vec=torch.tensor([[1.0, 0.0, -2.0], [0.5, 2.0, 3.0]], dtype=torch.double))


def gen_matrix_utils():
    """ot/matrix_utils.py:59-158 function by function on a zoo of matrices (definite, semi-definite, indefinite, asymmetric, batched):
    is_symmetric, min_eig, is_pd / is_spd (strict and not), make_psd (all four flag combinations, also diagonal 'matrices' given as
    vectors), sqrtm / invsqrtm, mean_cov (full and diag); and the exception type of w2_gaussian / compute_transport_operators on
    invalid arguments (ot/w2_utils.py:605-708)."""
    mu_, w2 = R.ref("ot.matrix_utils"), R.ref("ot.w2_utils")
    out = {}
    zoo = _matrix_zoo()
    for name, m in zoo.items():
        out[f"zoo/{name}"] = npy(m)
        if name == "diag_vec":
            for strict in (False, True):
                fixed, corr = mu_.make_psd(m.clone(), strict=strict, return_correction=True, diag=True)
                out[f"make_psd_diag/strict{int(strict)}/out"], out[f"make_psd_diag/strict{int(strict)}/corr"] = npy(fixed), npy(corr)
            continue
        out[f"{name}/is_symmetric"] = npy(mu_.is_symmetric(m).to(torch.int64))
        if name in ("asym",):
            continue
        out[f"{name}/min_eig"] = npy(mu_.min_eig(m))
        for strict in (False, True):
            out[f"{name}/is_pd/strict{int(strict)}"] = npy(mu_.is_pd(m, strict=strict).to(torch.int64))
            out[f"{name}/is_spd/strict{int(strict)}"] = npy(mu_.is_spd(m, strict=strict).to(torch.int64))
            fixed, corr = mu_.make_psd(m.clone(), strict=strict, return_correction=True)
            out[f"{name}/make_psd/strict{int(strict)}/out"], out[f"{name}/make_psd/strict{int(strict)}/corr"] = npy(fixed), npy(corr)
        if name in ("pd",):
            out[f"{name}/sqrtm"], out[f"{name}/invsqrtm"] = npy(mu_.sqrtm(m)), npy(mu_.invsqrtm(m))
        if name == "low":
            out[f"{name}/sqrtm"] = npy(mu_.sqrtm(m))
    # mean_cov
    g = torch.Generator().manual_seed(302)
    x = torch.randn(2, 30, 4, generator=g, dtype=torch.double)
    n = torch.tensor([30.0, 30.0], dtype=torch.double)
    mean, cov = mu_.mean_cov(x.sum(-2), x.transpose(-1, -2) @ x, n)
    out["mean_cov/x"], out["mean_cov/mean"], out["mean_cov/cov"] = npy(x), npy(mean), npy(cov)
    mean_d, var_d = mu_.mean_cov(x.sum(-2), (x ** 2).sum(-2), n, diag=True)
    out["mean_cov/mean_diag"], out["mean_cov/var_diag"] = npy(mean_d), npy(var_d)
    # argument errors
    pd, indef, asym = zoo["pd"], zoo["indef"], zoo["asym"]
    m5 = torch.zeros(5, dtype=torch.double)
    calls = {
        "w2_indef_source": lambda: w2.w2_gaussian(m5, m5, indef, pd),
        "w2_indef_target": lambda: w2.w2_gaussian(m5, m5, pd, indef),
        "w2_asym": lambda: w2.w2_gaussian(m5, m5, asym, pd),
        "w2_shape": lambda: w2.w2_gaussian(m5, torch.zeros(4, dtype=torch.double), pd, pd),
        "w2_indef_make_pd": lambda: w2.w2_gaussian(m5, m5, indef, pd, make_pd=True),
        "ops_indef": lambda: w2.compute_transport_operators(indef, pd, stochastic=False, diag=False, pg_star=0.0),
        "ops_pg_star_range": lambda: w2.compute_transport_operators(pd, pd, stochastic=False, diag=False, pg_star=1.5),
        "ops_diag_negative": lambda: w2.compute_transport_operators(torch.tensor([1.0, -1.0], dtype=torch.double), torch.ones(2, dtype=torch.double), stochastic=False, diag=True, pg_star=0.0),
    }
    for name, fn in calls.items():
        try:
            res = fn()
            val = res[0] if isinstance(res, tuple) else res
            out[f"err/{name}/value"] = npy(val)
            print(f"  {name}: returns")
        except Exception as e:  # noqa: BLE001
            out[f"err/{name}/error"] = np.frombuffer(type(e).__name__.encode(), dtype=np.uint8)
            print(f"  {name}: {type(e).__name__}: {str(e)[:80]}")
    save("matrix_utils.npz", out)


GT_CASES = {
    # name: (size, transport_cfg) -- GaussianTransport with a leading (per-position) shape, diagonal models, a perception-distortion mix
    "lead3_full": ((3, 6), dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)),
    "lead3_diag": ((3, 6), dict(diag=True, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)),
    "lead2x2_full_pg": ((2, 2, 5), dict(diag=False, stochastic=False, pg_star=0.3, make_pd=True, verbose=False, dtype=torch.double)),
    "nolead_diag_pg": ((7,), dict(diag=True, stochastic=False, pg_star=0.6, make_pd=True, verbose=False, dtype=torch.double)),
    "lead3_full_ema": ((3, 6), dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)),
}


def gen_gaussian_transport_shapes():
    """GaussianTransport with leading (per-position) shapes, diagonal models and pg_star > 0 (ot/transport/gaussian_transport.py:41-95,
    what LatentTransport(common_operator=False) builds): two updates per side, compute, transport of [*, B, D] and of [*, D] inputs."""
    gt = R.ref("ot.transport.gaussian_transport")
    out = {}
    for name, (size, tcfg) in GT_CASES.items():
        g = torch.Generator().manual_seed(401 + len(size))
        lead, d, B = size[:-1], size[-1], 40
        decay = 0.8 if name.endswith("_ema") else None
        cfg = dict(update_decay=decay, dtype=torch.double)
        op = gt.GaussianTransport(*size, source_cfg=cfg, target_cfg=cfg, transport_cfg=tcfg)
        mix = torch.randn(*lead, d, d, generator=g, dtype=torch.double) / math.sqrt(d)
        src = [torch.randn(*lead, B, d, generator=g, dtype=torch.double) @ mix * 1.3 + 0.4 for _ in range(2)]
        tgt = [torch.randn(*lead, B, d, generator=g, dtype=torch.double) * 0.7 - 0.1 for _ in range(2)]
        for a, b in zip(src, tgt):
            op.update(source_samples=a, target_samples=b)
        dist = op.compute()
        out[f"{name}/src"], out[f"{name}/tgt"] = npy(torch.stack(src)), npy(torch.stack(tgt))
        out[f"{name}/w2"], out[f"{name}/T"] = npy(dist), npy(op.transport_operator)
        out[f"{name}/src_mean"], out[f"{name}/src_cov"] = npy(op.source_model.mean), npy(op.source_model.cov)
        probe = src[0][..., :5, :]
        out[f"{name}/moved_batch"] = npy(op.transport(probe))
        out[f"{name}/moved_single"] = npy(op.transport(probe[..., 0, :]))
    save("gaussian_transport_shapes.npz", out)


def gen_edge_calls():
    """Edge cases of the OT helpers (ot/w2_utils.py: batch_ot_gmm's weight / variance validation, sinkhorn_log with 0 / 1 iterations, a
    threshold that stops at once and float32 inputs, apply_transport's shape check / zero noise / diagonal operator) and of QKVAttention
    (a width the heads do not divide): the value the reference returns, or the type of the exception it raises."""
    w2, nu = R.ref("ot.w2_utils"), R.ref("networks.nets_utils")
    out = {}
    from detfill import edge_calls
    for name, fn in edge_calls(w2, nu).items():
        try:
            res = fn()
            for i, v in enumerate(res if isinstance(res, tuple) else (res,)):
                out[f"{name}/value{i}"] = npy(v)
            print(f"  {name}: returns")
        except Exception as e:  # noqa: BLE001
            out[f"{name}/error"] = np.frombuffer(type(e).__name__.encode(), dtype=np.uint8)
            print(f"  {name}: {type(e).__name__}: {str(e).strip()[:70]}")
    save("edge_calls.npz", out)


STATE_DICT_CASES = {
    # name: (module path, class, args, kwargs)
    "gaussian_full": ("ot.distribution_models.gaussian_model", "GaussianModel", (3, 4), dict(dtype=torch.double)),
    "gaussian_diag": ("ot.distribution_models.gaussian_model", "GaussianModel", (4,), dict(dtype=torch.double, w2_cfg=dict(diag=True))),
    "gaussian_autograd": ("ot.distribution_models.gaussian_model", "GaussianModel", (4,), dict(dtype=torch.double, update_with_autograd=True)),
    "gaussian_autograd_diag": ("ot.distribution_models.gaussian_model", "GaussianModel", (2, 4), dict(dtype=torch.double, update_with_autograd=True, w2_cfg=dict(diag=True))),
    "gmm_diag": ("ot.distribution_models.gassian_mixture_model", "GaussianMixtureModel", (2, 3), dict(dtype=torch.double, mixture_cfg=dict(n_components=4), w2_cfg=dict(diag=True))),
    "gmm_full_autograd": ("ot.distribution_models.gassian_mixture_model", "GaussianMixtureModel", (3,), dict(dtype=torch.double, mixture_cfg=dict(n_components=4), update_with_autograd=True)),
    "codebook": ("ot.distribution_models.codebook_model", "CodebookModel", (2, 3), dict(mixture_cfg=dict(n_components=5))),
    "codebook_autograd": ("ot.distribution_models.codebook_model", "CodebookModel", (3,), dict(mixture_cfg=dict(n_components=5), update_with_autograd=True)),
    "gaussian_transport": ("ot.transport.gaussian_transport", "GaussianTransport", (2, 4), dict(source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double), store_source=True)),
    "gmm_transport": ("ot.transport.gmm_transport", "GMMTransport", (4,), dict(transport_type="argmax", transport_cfg=dict(diag=True, dtype=torch.double),
                      source_cfg=dict(dtype=torch.double, mixture_cfg=dict(n_components=3)), target_cfg=dict(dtype=torch.double, mixture_cfg=dict(n_components=3)))),
    "discrete_transport": ("ot.transport.discrete_transport", "DiscreteTransport", (4,), dict(transport_type="argmax",
                           source_cfg=dict(mixture_cfg=dict(n_components=3)), target_cfg=dict(mixture_cfg=dict(n_components=3)))),
    "gaussian_prior": ("prior.gaussian", "GaussianPrior", (), dict(loss_coeff=0.5)),
    "cond_prior": ("prior.conditional_gaussian", "ConditionalGaussianPrior", (), dict(dim=(2, 3), num_classes=4)),
    "cond_prior_ema": ("prior.conditional_gaussian", "ConditionalGaussianPrior", (), dict(dim=(2, 3), num_classes=4, embedding_ema_decay=0.9)),
    "codebook_prior": ("prior.codebook", "CodebookPrior", ((8, 2, 2), (1,)), dict(loss="kl", mixture_cfg=dict(n_components=6))),
}


def gen_state_dicts():
    """The state_dict keys, shapes, dtypes and `requires_grad` flags of the distribution models, transport operators and priors for
    several constructor options (what a checkpoint written by the reference holds and `load_state_dict` here must accept)."""
    out = {}
    for name, (mod, cls, args, kw) in STATE_DICT_CASES.items():
        torch.manual_seed(1)
        m = getattr(R.ref(mod), cls)(*args, **kw)
        rows = [f"{k}|{tuple(v.shape)}|{v.dtype}" for k, v in m.state_dict().items()]
        grads = [f"{k}|{int(p.requires_grad)}" for k, p in m.named_parameters()]
        out[f"{name}/state"] = np.array(rows or [""])
        out[f"{name}/params"] = np.array(grads or [""])
    save("state_dicts.npz", out)


def gen_stochastic():
    """The stochastic transport operator, eq. 19 (ot/w2_utils.py:391-458,732-786) for a DEGENERATE source (its raison d'etre):
    (T, Cw) for diagonal and full covariances, and ``apply_transport`` with the noise the reference drew (recorded as the
    difference to the noiseless transport)."""
    w2 = R.ref("ot.w2_utils")
    out = {}
    g = torch.Generator().manual_seed(91)
    # diagonal: two entries of every source variance vector vanish
    cs = torch.rand(2, 6, generator=g, dtype=torch.double) + 0.2
    cs[:, [1, 4]] = 0.0
    ct = torch.rand(2, 6, generator=g, dtype=torch.double) + 0.3
    T, Cw = w2.compute_transport_operators(cs.clone(), ct, stochastic=True, diag=True, pg_star=0.2, make_pd=True)
    out["diag/cs"], out["diag/ct"], out["diag/T"], out["diag/Cw"] = npy(cs), npy(ct), npy(T), npy(Cw)
    x = torch.randn(2, 9, 6, generator=g, dtype=torch.double)
    ms, mt = torch.randn(2, 6, generator=g, dtype=torch.double), torch.randn(2, 6, generator=g, dtype=torch.double)
    quiet = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-2), torch.zeros_like(T).unsqueeze(-2), diag=True)
    torch.manual_seed(31)
    noisy = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-2), Cw.unsqueeze(-2).abs() + 0.05, diag=True)
    out["diag/x"], out["diag/ms"], out["diag/mt"] = npy(x), npy(ms), npy(mt)
    out["diag/moved"], out["diag/moved_noisy"], out["diag/Cw_used"] = npy(quiet), npy(noisy), npy(Cw.abs() + 0.05)
    # full: nearly rank-deficient source (rank 3 of 6 plus 1e-6 I; exactly singular sources make the reference's own sqrtm return
    # NaN: the zero eigenvalues of Ct^1/2 Cs Ct^1/2 come out slightly negative), positive definite target
    d = 6
    a = torch.randn(2, d, 3, generator=g, dtype=torch.double)
    cs = a @ a.transpose(-1, -2) + 1e-6 * torch.eye(d, dtype=torch.double)
    b = torch.randn(2, d, d, generator=g, dtype=torch.double) / math.sqrt(d)
    ct = b @ b.transpose(-1, -2) + 0.4 * torch.eye(d, dtype=torch.double)
    T, Cw = w2.compute_transport_operators(cs, ct, stochastic=True, diag=False, pg_star=0.1, make_pd=True)
    out["full/cs"], out["full/ct"], out["full/T"], out["full/Cw"] = npy(cs), npy(ct), npy(T), npy(Cw)
    out["full/Cw_min_eig"] = npy(torch.linalg.eigvalsh(Cw)[..., 0])
    x = torch.randn(2, 9, d, generator=g, dtype=torch.double)
    quiet = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-3), torch.zeros_like(T).unsqueeze(-3), diag=False)
    cw_used = Cw + 0.05 * torch.eye(d, dtype=torch.double)          # strictly positive definite, as the sampler requires
    torch.manual_seed(32)
    noisy = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-3), cw_used.unsqueeze(-3), diag=False, make_pd=True)
    out["full/x"], out["full/moved"], out["full/moved_noisy"], out["full/Cw_used"] = npy(x), npy(quiet), npy(noisy), npy(cw_used)
    # well-conditioned FULL-RANK sources with stochastic=True (the common case: unit-scale target, source variances >= 1): Cw is
    # analytically zero and comes out at rounding / 1e-8 size (negative in the diagonal case), which ``apply_transport`` must treat
    # as "no noise" like the reference's allclose(Cw, 0) test (ot/w2_utils.py:507) -- the output is deterministic
    cs = torch.rand(2, 6, generator=g, dtype=torch.double) + 1.0
    ct = torch.rand(2, 6, generator=g, dtype=torch.double) * 0.5 + 0.3
    T, Cw = w2.compute_transport_operators(cs.clone(), ct, stochastic=True, diag=True, pg_star=0.2, make_pd=True)
    x = torch.randn(2, 9, 6, generator=g, dtype=torch.double)
    moved = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-2), Cw.unsqueeze(-2), diag=True)
    out["wc_diag/cs"], out["wc_diag/ct"], out["wc_diag/T"], out["wc_diag/Cw"] = npy(cs), npy(ct), npy(T), npy(Cw)
    out["wc_diag/x"], out["wc_diag/moved"] = npy(x), npy(moved)
    a = torch.randn(2, d, d, generator=g, dtype=torch.double) / math.sqrt(d)
    cs = a @ a.transpose(-1, -2) + 1.0 * torch.eye(d, dtype=torch.double)
    b = torch.randn(2, d, d, generator=g, dtype=torch.double) / math.sqrt(d)
    ct = 0.3 * (b @ b.transpose(-1, -2)) + 0.2 * torch.eye(d, dtype=torch.double)
    T, Cw = w2.compute_transport_operators(cs, ct, stochastic=True, diag=False, pg_star=0.1, make_pd=True)
    # full matrices: the 1e-8 regularisation of the swapped-role operator leaves Cw ~ 1.9e-8 I (positive definite, ABOVE allclose's
    # 1e-8): the reference does draw noise of standard deviation ~1.4e-4 here, with and without make_pd
    quiet = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-3), torch.zeros_like(T).unsqueeze(-3), diag=False)
    torch.manual_seed(33)
    moved = w2.apply_transport(x, ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-3), Cw.unsqueeze(-3), diag=False, make_pd=False)
    out["wc_full/cs"], out["wc_full/ct"], out["wc_full/T"], out["wc_full/Cw"] = npy(cs), npy(ct), npy(T), npy(Cw)
    out["wc_full/x"], out["wc_full/moved_quiet"], out["wc_full/moved"] = npy(x), npy(quiet), npy(moved)
    out["wc/Cw_absmax"] = np.array([out["wc_diag/Cw"].__abs__().max(), out["wc_full/Cw"].__abs__().max()])
    save("stochastic.npz", out)


def gen_gmm_full():
    """SURVEY 8f-2, full covariances: ``batch_w2_dissimilarity_gaussian`` / ``batch_ot_gmm(diag=False)`` (ot/w2_utils.py:138-270),
    ``gaussian_barycenter`` diagonal and full (w2_utils.py:325-385; the index its fixed point starts from is the one the reference
    draws under the recorded seed), and GaussianMixtureModel with full covariances: update x 3 -> fit -> energy / assign /
    predict_mean_var / w2 (gassian_mixture_model.py)."""
    w2 = R.ref("ot.w2_utils")
    gm = R.ref("ot.distribution_models.gassian_mixture_model")
    out = {}
    g = torch.Generator().manual_seed(71)

    def spd(*shape):
        d = shape[-1]
        a = torch.randn(*shape, d, generator=g, dtype=torch.double) / math.sqrt(d)
        return a @ a.transpose(-1, -2) + 0.3 * torch.eye(d, dtype=torch.double)

    lead, N, M, d = (2,), 3, 4, 5
    ms, mt = torch.randn(*lead, N, d, generator=g, dtype=torch.double), torch.randn(*lead, M, d, generator=g, dtype=torch.double) + 0.5
    cs, ct = spd(*lead, N, d), spd(*lead, M, d)
    ws = torch.softmax(torch.randn(*lead, N, generator=g, dtype=torch.double), -1)
    wt = torch.softmax(torch.randn(*lead, M, generator=g, dtype=torch.double), -1)
    out["fn/ms"], out["fn/mt"], out["fn/cs"], out["fn/ct"], out["fn/ws"], out["fn/wt"] = npy(ms), npy(mt), npy(cs), npy(ct), npy(ws), npy(wt)
    out["fn/dissimilarity"] = npy(w2.batch_w2_dissimilarity_gaussian(ms, mt, cs, ct, make_pd=True))
    total, coupling = w2.batch_ot_gmm(ms, mt, cs, ct, diag=False, weight_source=ws, weight_target=wt, max_iter=100)
    out["fn/ot_total"], out["fn/ot_coupling"] = npy(total), npy(coupling)
    mb, cb = w2.gaussian_barycenter(ms, torch.diagonal(cs, dim1=-2, dim2=-1), ws, diag=True)
    out["fn/bary_diag_mean"], out["fn/bary_diag_var"] = npy(mb), npy(cb)
    torch.manual_seed(5)
    out["fn/bary_init_index"] = np.array(int(torch.randint(size=(1,), high=N).item()))
    torch.manual_seed(5)
    mb, cb = w2.gaussian_barycenter(ms, cs, ws, diag=False, n_iter=100)
    out["fn/bary_full_mean"], out["fn/bary_full_cov"] = npy(mb), npy(cb)
    # ---- the model with full covariances
    w2_cfg = dict(diag=False, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
    mix = dict(metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax", inference_mode="argmax")
    for tag, lead, K, d, B, decay in (("sum", (2,), 3, 3, 96, None), ("ema", (), 4, 2, 128, 0.8)):
        g = torch.Generator().manual_seed(52)
        centres = torch.randn(*lead, K, d, generator=g, dtype=torch.double) * 3.0
        shape_mat = torch.randn(d, d, generator=g) * 0.5 + torch.eye(d)
        model = gm.GaussianMixtureModel(*lead, d, mixture_cfg={**mix, "n_components": K}, w2_cfg=w2_cfg, update_decay=decay,
                                        dtype=torch.double)
        model.train()
        out[f"{tag}/cfg"] = np.array([K, d, B, -1.0 if decay is None else decay])
        out[f"{tag}/vec_init"] = npy(model.vec_init)
        xs = [(_clusters(g, lead, K, d, B, centres.float(), noise=1.0) - centres.float().mean(-2, keepdim=True)).double() @ shape_mat.double()
              + centres.mean(-2, keepdim=True) for _ in range(3)]
        out[f"{tag}/batches"] = npy(torch.stack(xs))
        for i, x in enumerate(xs):
            if i == 0:
                torch.manual_seed(83)
            model.update(x)
            out[f"{tag}/step{i}/mean"], out[f"{tag}/step{i}/cov"] = npy(model.mean).copy(), npy(model.cov).copy()
            out[f"{tag}/step{i}/weights"], out[f"{tag}/step{i}/n_obs"] = npy(model.weights).copy(), npy(model._n_obs).copy()
        model.fit()
        out[f"{tag}/fit/mean"], out[f"{tag}/fit/cov"], out[f"{tag}/fit/weights"] = npy(model.mean), npy(model.cov), npy(model.weights)
        model.eval()
        energy = model.energy(xs[-1])
        weights, _, dist = model.assign(xs[-1])
        pm, pv = model.predict_mean_var(weights)
        out[f"{tag}/energy"], out[f"{tag}/assign_onehot"], out[f"{tag}/assign_probs"] = npy(energy), npy(weights), npy(dist.probs)
        out[f"{tag}/pred_mean"], out[f"{tag}/pred_cov"] = npy(pm), npy(pv)
        oc = spd(*lead, K, d)
        other = torch.distributions.MixtureSameFamily(
            torch.distributions.Categorical(torch.ones(*lead, K, dtype=torch.double) / K),
            torch.distributions.MultivariateNormal(centres, covariance_matrix=oc))
        out[f"{tag}/centres"], out[f"{tag}/other_cov"] = npy(centres), npy(oc)
        out[f"{tag}/w2"] = npy(model.w2(other))
    save("gmm_full.npz", out)


VIT_CASES = [
    # tag, common cfg, batch
    ("d32", dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.0, emb_dropout=0.,
                 num_classes=10), 3),
    ("d128", dict(image_size=32, patch_size=8, dim=128, depth=1, heads=4, mlp_dim=256, channels=3, dropout=0.0, emb_dropout=0.,
                  num_classes=None), 2),
]


def gen_vit():
    """G12: the reference's ViT as encoder (patches + 2 embed tokens (+ class token) -> the embed tokens)
    and as decoder (1 latent token + learned patch tokens -> image), configured like tests/test_conditional_vit_vae.py:
    41-67 with dropout 0 and deterministic closed-form weights: outputs, input gradient, every parameter gradient."""
    vit = R.ref("networks.vit")
    out = {}
    for tag, cfg, B in VIT_CASES:
        enc = vit.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False,
                      **cfg)
        dec = vit.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True,
                      **cfg)
        labels = torch.arange(B) % 10 if cfg["num_classes"] else None
        nets = [(enc, "enc", det_input((B, cfg["channels"], cfg["image_size"], cfg["image_size"]), 0.3)),
                (dec, "dec", det_input((B, 1, cfg["dim"]), 0.8))]
        for net, nm, xin in nets:
            net.train()
            fill_vit_state_dict(net.state_dict())
            x = xin.clone().requires_grad_(True)
            y = net(x, labels=labels)
            g = det_input(tuple(y.shape), 1.1, 0.5)
            y.backward(g)
            out[f"{tag}/{nm}/x"], out[f"{tag}/{nm}/y"], out[f"{tag}/{nm}/gy"], out[f"{tag}/{nm}/gx"] = npy(x), npy(y), npy(g), npy(x.grad)
            for k, p in net.named_parameters():
                if p.numel() <= 4096:
                    out[f"{tag}/{nm}/grad/{k}"] = npy(p.grad)
                else:  # large matrices: checksums (sum, L2, the first 16 entries) keep the fixture small
                    gd = p.grad.double().flatten()
                    out[f"{tag}/{nm}/gradsum/{k}"] = np.concatenate([[gd.sum().item(), gd.norm().item()], gd[:16].numpy()])
        if labels is not None:
            out[f"{tag}/labels"] = npy(labels)
    save("vit.npz", out)


VIT_VARIANTS = {
    # name: (ctor kwargs, input kind) -- corners of the ViT constructor no golden of rounds 1-2 visits
    "out_embed_and_class": (dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=2,
                                 output_tokens=["embed", "class"], num_classes=5), "image"),
    "patches_in_patches_out": (dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=1,
                                    output_tokens="input", embed_to_patch=True), "image"),
    "default_patch_rect_image": (dict(image_size=(8, 16), dim=16, depth=1, heads=4, channels=1, n_embed_tokens=1), "image"),
    "preprocess_identity": (dict(image_size=8, patch_size=4, dim=16, depth=1, preprocess_depth=0, heads=2, mlp_dim=32, channels=2,
                                 n_embed_tokens=2, output_tokens="embed", num_classes=3), "image"),
    "class_only": (dict(image_size=8, patch_size=4, dim=16, depth=2, heads=2, mlp_dim=32, channels=2, n_embed_tokens=1,
                        output_tokens="class", num_classes=3), "image"),
    "tokens_in_embed_none": (dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=None,
                                  n_input_tokens=3, patch_to_embed=False, output_tokens="embed"), "tokens"),
    "no_embed_tokens": (dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=0,
                             output_tokens="input"), "image"),
    "time_output_without_time": (dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, channels=2, output_tokens="time"), "image"),
    "bad_patch": (dict(image_size=10, patch_size=4, dim=16, depth=1, heads=2, channels=2), "image"),
    "bad_output_token": (dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, channels=2, output_tokens="latent"), "image"),
}


def gen_vit_variants():
    """The ViT at corners of its constructor (networks/vit.py:71-193): several output token types, patches in -> patches out, the
    default patch size on a rectangular image, `preprocess_depth=0`, the class token as only output, token inputs with
    `n_embed_tokens=None`, no embed tokens; and the exception types of invalid configurations.  dropout 0, closed-form weights."""
    vit = R.ref("networks.vit")
    out = {}
    B = 3
    for name, (kw, kind) in VIT_VARIANTS.items():
        try:
            net = vit.ViT(dropout=0.0, emb_dropout=0., **kw)
            net.train()
            fill_vit_state_dict(net.state_dict())
            if kind == "image":
                h, w = kw["image_size"] if isinstance(kw["image_size"], tuple) else (kw["image_size"],) * 2
                x = det_input((B, kw["channels"], h, w), 0.3)
            else:
                x = det_input((B, kw["n_input_tokens"], kw["dim"]), 0.8)
            x = x.clone().requires_grad_(True)
            labels = torch.arange(B) % kw["num_classes"] if kw.get("num_classes") else None
            y = net(x, labels=labels)
            g = det_input(tuple(y.shape), 1.1, 0.5)
            y.backward(g)
        except Exception as e:  # noqa: BLE001 -- a corner the reference rejects (or breaks on): the error type is the golden
            out[f"{name}/error"] = np.frombuffer(type(e).__name__.encode(), dtype=np.uint8)
            print(f"  {name}: the reference raises {type(e).__name__}: {str(e)[:90]}")
            continue
        out[f"{name}/x"], out[f"{name}/y"], out[f"{name}/gy"], out[f"{name}/gx"] = npy(x), npy(y), npy(g), npy(x.grad)
        out[f"{name}/out_size"] = np.array(list(net.out_size))
        out[f"{name}/param_names"] = np.array([k for k, _ in net.state_dict().items()])
        out[f"{name}/param_shapes"] = np.array([";".join(str(d) for d in v.shape) for _, v in net.state_dict().items()])
        if labels is not None:
            out[f"{name}/labels"] = npy(labels)
        for k, p in net.named_parameters():
            out[f"{name}/grad/{k}"] = npy(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), dtype=np.float32)
        print(f"  {name}: y {tuple(y.shape)}")
    save("vit_variants.npz", out)


def gen_vit_causal():
    """G16: the d32 ViT of G12 with ``causal_mask=True`` (every layer's attention sees tokens <= t only,
    networks/vit.py:215-217,225): output and input gradient for both roles."""
    vit = R.ref("networks.vit")
    out = {}
    tag, cfg, B = VIT_CASES[0]
    roles = {"enc": dict(n_embed_tokens=2, n_input_tokens=None, patch_to_embed=True, embed_to_patch=False),
             "dec": dict(n_embed_tokens=None, n_input_tokens=1, patch_to_embed=False, embed_to_patch=True)}
    labels = torch.arange(B) % 10
    for nm, role in roles.items():
        net = vit.ViT(output_tokens="embed", causal_mask=True, **role, **cfg)
        net.train()
        fill_vit_state_dict(net.state_dict())
        shape = (B, cfg["channels"], cfg["image_size"], cfg["image_size"]) if nm == "enc" else (B, 1, cfg["dim"])
        x = det_input(shape, 0.3 if nm == "enc" else 0.8).clone().requires_grad_(True)
        y = net(x, labels=labels)
        g = det_input(tuple(y.shape), 1.1, 0.5)
        y.backward(g)
        out[f"{nm}/x"], out[f"{nm}/y"], out[f"{nm}/gy"], out[f"{nm}/gx"] = npy(x), npy(y), npy(g), npy(x.grad)
        out[f"{nm}/grad_l2"] = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
    out["labels"] = npy(labels)
    out["tag"] = np.array([ord(c) for c in tag])
    save("vit_causal.npz", out)


VIT_CROSS_CASES = {
    # role, preprocess_depth, causal_mask
    "enc_p1": ("enc", 1, False),          # 2 embed tokens attend to 16 patches + class token after one encoder layer
    "dec_p0_causal": ("dec", 0, True),    # 16 patch tokens (causal self-attention) attend to latent + class token, no preprocessing
}


def gen_vit_cross():
    """G17 (the last open item): the reference's cross-attention ViT -- ``preprocess_depth`` makes the output
    tokens the target of an nn.TransformerDecoder whose memory is the other tokens (networks/vit.py:171-181,240-244) -- in the
    d32 configuration of G12 for both roles: outputs, input gradient, every parameter gradient."""
    vit = R.ref("networks.vit")
    out = {}
    _, cfg, B = VIT_CASES[0]
    roles = {"enc": dict(n_embed_tokens=2, n_input_tokens=None, patch_to_embed=True, embed_to_patch=False),
             "dec": dict(n_embed_tokens=None, n_input_tokens=1, patch_to_embed=False, embed_to_patch=True)}
    labels = torch.arange(B) % 10
    for tag, (role, pre, causal) in VIT_CROSS_CASES.items():
        net = vit.ViT(output_tokens="embed", preprocess_depth=pre, causal_mask=causal, **roles[role], **cfg)
        net.train()
        fill_vit_state_dict(net.state_dict())
        shape = (B, cfg["channels"], cfg["image_size"], cfg["image_size"]) if role == "enc" else (B, 1, cfg["dim"])
        x = det_input(shape, 0.3 if role == "enc" else 0.8).clone().requires_grad_(True)
        y = net(x, labels=labels)
        g = det_input(tuple(y.shape), 1.1, 0.5)
        y.backward(g)
        out[f"{tag}/x"], out[f"{tag}/y"], out[f"{tag}/gy"], out[f"{tag}/gx"] = npy(x), npy(y), npy(g), npy(x.grad)
        out[f"{tag}/param_names"] = np.array(list(net.state_dict().keys()))
        for k, p in net.named_parameters():
            if p.numel() <= 4096:
                out[f"{tag}/grad/{k}"] = npy(p.grad)
            else:
                gd = p.grad.double().flatten()
                out[f"{tag}/gradsum/{k}"] = np.concatenate([[gd.sum().item(), gd.norm().item()], gd[:16].numpy()])
    out["labels"] = npy(labels)
    save("vit_cross.npz", out)


AR_CFG = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.0, emb_dropout=0.,
              num_classes=10, n_input_tokens=7, n_embed_tokens=0, output_tokens="input", patch_to_embed=False, embed_to_patch=False,
              causal_mask=True)


def gen_vit_autoregressive():
    """G18: ``AutoRegressive`` (networks/vit.py:249-260) -- 7 token ids of a 13-word vocabulary, causal self-attention, class token
    -> logits: output, parameter gradients (the ids carry none)."""
    vit = R.ref("networks.vit")
    out = {}
    B, V = 4, 13
    net = vit.AutoRegressive(vocab_size=V, **AR_CFG)
    net.train()
    fill_vit_state_dict(net.state_dict())
    ids = (torch.arange(B * 7).reshape(B, 7) * 5 + 3) % V
    labels = torch.arange(B) % 10
    y = net(ids, labels=labels)
    g = det_input(tuple(y.shape), 1.1, 0.5)
    y.backward(g)
    out["ids"], out["labels"], out["y"], out["gy"] = npy(ids), npy(labels), npy(y), npy(g)
    out["param_names"] = np.array(list(net.state_dict().keys()))
    for k, p in net.named_parameters():
        out[f"grad/{k}"] = npy(p.grad)
    save("vit_autoregressive.npz", out)


def gen_vit_vae():
    """G13: ConditionalGaussianPrior alone (learned embeddings: z, loss, gradients of x and of the two
    embeddings; EMA embeddings: the buffers after two training steps) and VAE.nelbo of the conditional ViT VAE of
    tests/test_conditional_vit_vae.py:41-85 (dropout 0, dim 32 / 16x16 images to keep the fixture small, closed-form
    weights, explicit eps): the three loss scalars, preds, latents and gradient checksums."""
    vit, pc, vae = R.ref("networks.vit"), R.ref("prior.conditional_gaussian"), R.ref("model.vae")
    out = {}
    # ---- the prior alone
    B, D, C = 6, 8, 4
    labels = torch.tensor([0, 3, 1, 1, 2, 0])
    x = det_input((B, 2, D), 0.4).requires_grad_(True)
    eps = normal((B, 1, D), 91)
    torch.manual_seed(7)
    prior = pc.ConditionalGaussianPrior(dim=(1, D), num_classes=C, loss_coeff=0.3, annealing_steps=10)
    prior.train()
    with _FixedEps(eps):
        z, loss, _ = prior(x, step=4, labels=labels)
    w = det_input((B, 1, D), 2.2)
    ((z * w).sum() + loss.sum()).backward()
    out["prior/x"], out["prior/eps"], out["prior/labels"], out["prior/w"] = npy(x), npy(eps), npy(labels), npy(w)
    out["prior/mu_weight"], out["prior/log_std_weight"] = npy(prior._mu.weight), npy(prior._log_std.weight)
    out["prior/z"], out["prior/loss"], out["prior/gx"] = npy(z), npy(loss), npy(x.grad)
    out["prior/g_mu"], out["prior/g_log_std"] = npy(prior._mu.weight.grad), npy(prior._log_std.weight.grad)
    torch.manual_seed(8)
    ema = pc.ConditionalGaussianPrior(dim=(1, D), num_classes=C, loss_coeff=1.0, embedding_ema_decay=0.9)
    ema.train()
    out["ema/mu_weight0"], out["ema/log_std_weight0"] = npy(ema._mu.weight).copy(), npy(ema._log_std.weight).copy()
    for step in range(2):
        xs = det_input((B, 2, D), 0.4 + step)
        es = normal((B, 1, D), 92 + step)
        with _FixedEps(es):
            z, loss, _ = ema(xs, step=0, labels=labels)
        out[f"ema/step{step}/x"], out[f"ema/step{step}/eps"] = npy(xs), npy(es)
        out[f"ema/step{step}/z"], out[f"ema/step{step}/loss"] = npy(z), npy(loss)
        out[f"ema/step{step}/mu"], out[f"ema/step{step}/log_std"] = npy(ema._mu.weight).copy(), npy(ema._log_std.weight).copy()
        out[f"ema/step{step}/size"] = npy(ema._size).copy()
    # ---- the conditional ViT VAE
    cfg = dict(image_size=16, patch_size=4, dim=32, depth=2, heads=4, mlp_dim=64, channels=3, dropout=0.0, emb_dropout=0.,
               num_classes=10)
    enc = vit.ViT(n_embed_tokens=2, n_input_tokens=None, output_tokens="embed", patch_to_embed=True, embed_to_patch=False, **cfg)
    dec = vit.ViT(n_embed_tokens=None, n_input_tokens=1, output_tokens="embed", patch_to_embed=False, embed_to_patch=True, **cfg)
    fill_vit_state_dict(enc.state_dict())
    fill_vit_state_dict(dec.state_dict())
    torch.manual_seed(9)
    prior = pc.ConditionalGaussianPrior(dim=(1, 32), num_classes=10, loss_coeff=0.1, empirical_kl=False, reparam_dim=1,
                                        annealing_steps=1000)
    m = vae.VAE(metrics=R._MetricCollection(), encoder=enc, decoder=dec, prior=prior, conditional=True)
    m.train()
    B = 5
    x = det_input((B, 3, 16, 16), 0.6)
    labels = torch.tensor([3, 0, 9, 3, 7])
    eps = normal((B, 1, 32), 95)
    with _FixedEps(eps):
        loss, logs, art = m.nelbo({"samples": x, "target": x, "kwargs": {"labels": labels}}, 0)
    loss.backward()
    out["vae/x"], out["vae/labels"], out["vae/eps"] = npy(x), npy(labels), npy(eps)
    out["vae/mu_weight"], out["vae/log_std_weight"] = npy(prior._mu.weight), npy(prior._log_std.weight)
    out["vae/loss"] = npy(torch.stack([logs["train/loss/total"], logs["train/loss/recon"], logs["train/loss/prior"]]))
    out["vae/preds"], out["vae/latents"] = npy(art["preds"]), npy(art["latents"])
    names, gsum, gl2 = [], [], []
    for pre, net in (("encoder.", m.encoder), ("decoder.", m.decoder), ("prior.", m.prior)):
        for k, p in net.named_parameters():
            names.append(pre + k)
            gsum.append(p.grad.double().sum().item())
            gl2.append(p.grad.double().norm().item())
    out["vae/param_names"], out["vae/grad_sum"], out["vae/grad_l2"] = np.array(names), np.array(gsum), np.array(gl2)
    save("vit_vae.npz", out)


def gen_gmm_recovery():
    """G15: what the reference's GaussianMixtureModel itself reaches on its own recovery experiment (the reference's test
    of it cannot run: it names an undefined variable, test_distribution_models.py:180, and its W2 < 0.1 bound is far from
    what the class achieves).  Recorded: W2(model, truth) after ``fit(samples)`` and after streaming ``update`` + ``fit``."""
    gm = R.ref("ot.distribution_models.gassian_mixture_model")
    from detfill import gmm_recovery_inputs
    (lead, k, dim, n), mean, var, truth, samples, order = gmm_recovery_inputs()
    cfg = dict(w2_cfg={"diag": True}, dtype=torch.double, mixture_cfg={"n_components": k, "training_mode": "argmax", "topk": None})
    torch.manual_seed(103)
    fitted = gm.GaussianMixtureModel(*lead, dim, **cfg)
    fitted.train()
    fitted.fit(samples)
    torch.manual_seed(105)
    streamed = gm.GaussianMixtureModel(*lead, dim, **cfg, update_decay=None)
    streamed.train()
    for batch in samples[:, order].split(100, dim=-2):
        streamed.update(batch)
    streamed.fit()
    out = {"w2_fit": npy(fitted.w2(truth)), "w2_update": npy(streamed.w2(truth)),
           "samples_checksum": np.array([samples.sum().item(), samples.square().sum().item(), float(order[:16].sum())]),
           "fit_mean": npy(fitted.mean), "update_mean": npy(streamed.mean)}
    save("gmm_recovery.npz", out)
    print("gmm_recovery", out["w2_fit"], out["w2_update"])


def gen_mixture_modes():
    """SURVEY 8f-2, the assignment modes beyond one-hot: (a) CodebookPrior in the soft 'mean' training mode with the 'kl' entropy
    loss (prior/codebook.py:80-105): values and the gradient that reaches the encoder through the assignment probabilities;
    (b) 'gumbel-softmax' / 'gumbel-hardmax' assignments (distribution_models/base.py:234-235) of a CodebookModel and of a
    diagonal GaussianMixtureModel, with the Gumbel draws the reference made recorded next to the results."""
    import torch.nn.functional as F
    cm = R.ref("ot.distribution_models.codebook_model")
    gm = R.ref("ot.distribution_models.gassian_mixture_model")
    pr = R.ref("prior.codebook")
    out = {}
    # ---- (a) soft prior
    size, K, Bp = (6, 2, 2), 8, 16
    prior = pr.CodebookPrior(size, (1, 2, 3), loss="kl", loss_coeff=0.7,
                             mixture_cfg=dict(n_components=K, training_mode="mean", inference_mode="mean", temperature=0.5))
    prior.train()
    g = torch.Generator().manual_seed(141)
    w = torch.randn(Bp, *size, generator=g)
    for step in range(2):
        x = (torch.randn(Bp, *size, generator=g) * 1.5).requires_grad_(True)
        if step == 0:
            torch.manual_seed(179)
        z, loss, art = prior(x, step=step)
        ((z * w).sum() + loss.sum()).backward()
        out[f"soft_prior/step{step}/x"], out[f"soft_prior/step{step}/z"], out[f"soft_prior/step{step}/loss"] = npy(x), npy(z), npy(loss)
        out[f"soft_prior/step{step}/gx"], out[f"soft_prior/step{step}/probs"] = npy(x.grad), npy(art["distribution"].probs)
        out[f"soft_prior/step{step}/codebook"] = npy(prior.codebook_model.codebook).copy()
    out["soft_prior/w"] = npy(w)
    # ---- (b) Gumbel modes: torch's own F.gumbel_softmax arithmetic with the draws recorded
    rec = {}
    orig = F.gumbel_softmax

    def recording(logits, tau=1, hard=False, eps=1e-10, dim=-1):
        gum = -torch.empty_like(logits).exponential_().log()
        rec["g"] = gum.clone()
        y = ((logits + gum) / tau).softmax(dim)
        if hard:
            idx = y.max(dim, keepdim=True)[1]
            return torch.zeros_like(logits).scatter_(dim, idx, 1.0) - y.detach() + y
        return y

    F.gumbel_softmax = recording
    try:
        for mode in ("gumbel-softmax", "gumbel-hardmax"):
            torch.manual_seed(7)
            m = cm.CodebookModel(5, mixture_cfg=dict(n_components=6, training_mode=mode, temperature=0.7))
            with torch.no_grad():
                m.codebook.copy_(det_input((6, 5), phase=0.4, amp=1.2))
            m.train()
            x = det_input((24, 5), phase=1.1, amp=1.5).requires_grad_(True)
            wgt = det_input((24, 6), phase=2.0)
            weights, _, _ = m.assign(x)
            (weights * wgt).sum().backward()
            k = f"codebook/{mode}"
            out[f"{k}/x"], out[f"{k}/codebook"], out[f"{k}/gumbel"] = npy(x), npy(m.codebook), npy(rec["g"])
            out[f"{k}/weights"], out[f"{k}/gx"], out[f"{k}/w"] = npy(weights), npy(x.grad), npy(wgt)
            torch.manual_seed(8)
            mm = gm.GaussianMixtureModel(5, w2_cfg={"diag": True}, dtype=torch.double,
                                         mixture_cfg=dict(n_components=4, training_mode=mode, temperature=1.3))
            with torch.no_grad():
                mm.mean.copy_(det_input((4, 5), phase=0.2, amp=1.0, dtype=torch.double))
                mm.cov = det_input((4, 5), phase=0.9, amp=0.3, dtype=torch.double).abs() + 0.5
                mm._weights = torch.tensor([0.1, 0.4, 0.3, 0.2], dtype=torch.double)
            mm.train()
            xs = det_input((20, 5), phase=0.6, amp=1.4, dtype=torch.double)
            weights, _, _ = mm.assign(xs)
            k = f"gmm/{mode}"
            out[f"{k}/x"], out[f"{k}/mean"], out[f"{k}/var"] = npy(xs), npy(mm.mean), npy(mm.cov)
            out[f"{k}/gumbel"], out[f"{k}/weights"] = npy(rec["g"]), npy(weights)
    finally:
        F.gumbel_softmax = orig
    save("mixture_modes.npz", out)


def gen_nelbo_b32():
    """A WELL-CONDITIONED whole-network pin straight from the reference: the MNIST test configuration with torch's default
    initialisation under manual_seed(1234) (encoder built first, then decoder -- the product's classes make the same RNG
    draws), batch 32 (BatchNorm over >= 32 positions everywhere), explicit eps.  Losses, every parameter's gradient norm and
    sum, a few full gradients, the reference's own fp64 evaluation beside them."""
    cnn, pg, vae = R.ref("networks.cnn"), R.ref("prior.gaussian"), R.ref("model.vae")
    out = {}
    for residual in ("add", None):
        tag = str(residual)

        def make():
            torch.manual_seed(1234)
            enc = cnn.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual=residual)
            dec = cnn.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual=residual)
            m = vae.VAE(metrics=R._MetricCollection(), encoder=enc, decoder=dec, prior=pg.GaussianPrior(loss_coeff=0.1))
            return m.train()

        B = 32
        m = make()
        x, eps = mnist_like(B, seed=52), normal((B, 128, 1, 1), seed=53)
        with _FixedEps(eps):
            loss, logs, art = m.nelbo({"samples": x, "target": x, "kwargs": {}}, 0)
        loss.backward()
        params = [(pre + k, p) for pre, net in (("encoder.", m.encoder), ("decoder.", m.decoder)) for k, p in net.named_parameters()]
        out[f"{tag}/loss"] = npy(torch.stack([logs["train/loss/total"], logs["train/loss/recon"], logs["train/loss/prior"]]))
        out[f"{tag}/preds"] = npy(art["preds"][:2])
        out[f"{tag}/param_names"] = np.array([k for k, _ in params])
        out[f"{tag}/param_sum"] = np.array([p.detach().double().sum().item() for _, p in params])
        out[f"{tag}/param_l2"] = np.array([p.detach().double().norm().item() for _, p in params])
        out[f"{tag}/grad_l2"] = np.array([p.grad.double().norm().item() for _, p in params])
        out[f"{tag}/grad_sum"] = np.array([p.grad.double().sum().item() for _, p in params])
        names = [k for k, _ in params]
        for k in ("encoder.0.block.0.weight", "encoder.2.block.1.weight", "encoder.4.block.2.qkv.weight", "decoder.4.block.2.qkv.weight",
                  "decoder.0.skip.weight", "decoder.2.block.0._normalization.weight"):
            if k in names:
                out[f"{tag}/grad_full/{k}"] = npy(params[names.index(k)][1].grad)
        m64 = make().double()
        with _FixedEps(eps.double()):
            loss64, _, _ = m64.nelbo({"samples": x.double(), "target": x.double(), "kwargs": {}}, 0)
        loss64.backward()
        out[f"{tag}/grad_l2_f64"] = np.array([p.grad.norm().item() for net in (m64.encoder, m64.decoder) for _, p in net.named_parameters()])
    save("nelbo_b32.npz", out)


def gen_w2_prior():
    """Gaussian W2 with empirical covariance as a differentiable loss term (the reference's pieces are
    GaussianModel._stats + mean_cov + w2_gaussian; here they are composed under torch.autograd exactly as
    DistributionModel's update_with_autograd path composes them, ot/distribution_models/base.py:82-89): loss and dL/dz."""
    mu_ = R.ref("ot.matrix_utils")
    w2 = R.ref("ot.w2_utils")
    gm = R.ref("ot.distribution_models.gaussian_model")
    out = {}
    for D, B in ((16, 64), (128, 256), (128, 1024)):
        g = torch.Generator().manual_seed(900 + D + B)
        mix = torch.randn(D, D, generator=g) / math.sqrt(D)
        z = (torch.randn(B, D, generator=g) @ (0.6 * mix + 0.7 * torch.eye(D)) + 0.3 * torch.randn(D, generator=g)).float()
        tm = 0.2 * torch.randn(D, generator=g, dtype=torch.double)
        tq = torch.randn(D, D, generator=g, dtype=torch.double) / math.sqrt(D)
        tc = tq @ tq.T + 0.5 * torch.eye(D, dtype=torch.double)
        model = gm.GaussianModel(D, dtype=torch.double)
        for tag, tmean, tcov in (("std", torch.zeros(D, dtype=torch.double), torch.eye(D, dtype=torch.double)), ("gen", tm, tc)):
            zz = z.clone().requires_grad_(True)
            n, sx, sxx = model._stats(zz.double(), reduce=False)
            mean, cov = mu_.mean_cov(sx, sxx, n)
            loss = w2.w2_gaussian(mean, tmean, cov, tcov, make_pd=True)
            loss.backward()
            k = f"D{D}_B{B}/{tag}"
            out[f"{k}/loss"] = npy(loss)
            if B <= 256:
                out[f"{k}/gz"] = npy(zz.grad)
            else:  # benchmark-sized batch: the first rows, every row's sum and every column's sum of the gradient
                out[f"{k}/gz_head"], out[f"{k}/gz_rowsum"], out[f"{k}/gz_colsum"] = \
                    npy(zz.grad[:8]), npy(zz.grad.double().sum(1)), npy(zz.grad.double().sum(0))
            out[f"{k}/mean"], out[f"{k}/cov_trace"] = npy(mean), npy(torch.diagonal(cov).sum())
        kk = f"D{D}_B{B}"
        if B <= 256:
            out[f"{kk}/z"] = npy(z)
        else:  # the benchmark-sized case is regenerated from its seed by the test (detfill-free: plain torch generator calls)
            out[f"{kk}/z_checksum"] = np.array([z.double().sum().item(), z.double().square().sum().item()])
        out[f"{kk}/target_mean"], out[f"{kk}/target_cov"] = npy(tm), npy(tc)
    save("w2_prior.npz", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["convlayer", "attention", "cnn_small", "nelbo", "prior", "sinkhorn", "gaussian_ot", "codebook",
                             "codebook_kmeans", "discrete", "gmm", "vit", "vit_vae", "gmm_recovery", "vit_causal", "w2_prior", "nelbo_b32", "mixture_modes", "gmm_full", "stochastic", "vit_cross", "vit_autoregressive", "cnn_small_opts", "gmm_autograd", "codebook_autograd", "cnn_variants", "nelbo_expansion", "latent_transport_routing", "layouts", "cnn_shapes", "utils_small", "codebook_options", "vit_variants", "prior_corners", "partial_checkpoint", "matrix_utils", "gaussian_transport_shapes", "edge_calls", "state_dicts", "sinkhorn_autograd"]
    for w in which:
        globals()["gen_" + w]()
