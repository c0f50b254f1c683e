"""GPU parity tests (run with ``-m gpu`` on the MI355X box).  Every case drives the product path (Python API ->
ctypes -> C ABI -> HIP kernels) and compares with
  * the golden vectors recorded from the real reference (``tests/golden``), and
  * the CPU oracle (``oracle/otvae_oracle.py``) on the same seeded inputs at larger sizes.
Bounds: fp32 network / Sinkhorn results 1e-4 relative (max-norm, the tolerance BASELINE.json's north_star states);
fp64 OT arithmetic 1e-8; indices bit-exact.  A full report of every measured error is appended to
``gpurun_out/parity_report.txt``.
"""
import math
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, group, load_golden, rel_err
from detfill import det_input, fill_state_dict, mnist_like, normal

pytestmark = pytest.mark.gpu

TOL32 = 1e-4
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.txt")


# whole-network gradients, tensor by tensor (Report.check_grads): fp32 forward + backward through ~50 convolutions, 40 training-
# mode BatchNorms and 10 attention blocks against the CPU oracle's fp32 autograd
GRAD_TOL_L2, GRAD_TOL_MAX = 2e-3, 5e-3


class Report:
    def __init__(self, title):
        self.title, self.rows, self.failed = title, [], []

    def check(self, name, got, want, tol=TOL32, exact=False, floor=0.0):
        """max|got-want| / max(|want|_inf, floor) < tol.  ``floor`` gives an absolute scale to quantities whose exact
        value is zero (e.g. the bias gradient of a conv feeding a training-mode BatchNorm), where the reference itself
        only holds rounding noise."""
        if exact:
            ok = torch.equal(got.cpu(), want.cpu())
            err = 0.0 if ok else float("inf")
        else:
            a, b = got.detach().double().cpu(), want.detach().double().cpu()
            err = (a - b).abs().max().item() / max(b.abs().max().item(), floor, 1e-30)
            ok = err < tol and math.isfinite(err)
        self.rows.append((name, err, tol, ok))
        if not ok:
            self.failed.append((name, err, tol))

    def check_vs_truth(self, name, got, ref32, truth, floor=0.0, contract=TOL32, factor=1.5, l2=False):
        """The bound VERDICT r3 #7 prescribes instead of a constant fitted to the measurement: with the fp64 truth of the quantity
        computed in the test, |hip - truth| <= max(contract * scale, factor * |reference_fp32 - truth|) -- the north-star's fp32
        contract (1e-4 relative) or, where fp32 itself cannot hold that, 1.5x the reference's OWN fp32 error on the same inputs.
        scale = max(|truth|_inf, floor); both errors are recorded."""
        t = truth.detach().double().cpu()
        if l2:   # relative L2 over the whole tensor: sees an error that is small everywhere but coherent (the same sign along a row)
            scale = max(float(t.norm()), 1e-30)
            e_hip = float((got.detach().double().cpu() - t).norm()) / scale
            e_ref = float((ref32.detach().double().cpu() - t).norm()) / scale
        else:
            scale = max(t.abs().max().item(), floor, 1e-30)
            e_hip = (got.detach().double().cpu() - t).abs().max().item() / scale
            e_ref = (ref32.detach().double().cpu() - t).abs().max().item() / scale
        tol = max(contract, factor * e_ref)
        ok = e_hip <= tol and math.isfinite(e_hip)
        self.rows.append((f"{name} vs fp64 truth [reference fp32 vs truth: {e_ref:.2e}]", e_hip, tol, ok))
        if not ok:
            self.failed.append((name, e_hip, tol))

    def check_grads_vs_truth(self, name, got, ref32, truth, names=None, contract_l2=GRAD_TOL_L2, contract_max=GRAD_TOL_MAX, factor=1.5,
                             known=None):
        """``check_grads`` with the evidence-based bound of ``check_vs_truth``: per tensor, HIP and the fp32 reference are both
        measured against the fp64 truth (same floors as ``check_grads``), and the tensor's bound is the standing contract or
        ``factor`` x the reference's own fp32 error on it, whichever is larger."""
        got, ref32, truth = ([t.detach().double().cpu() for t in lst] for lst in (got, ref32, truth))
        rms_scale = max(float(b.norm()) / max(b.numel(), 1) ** 0.5 for b in truth)
        max_scale = max(float(b.abs().max()) for b in truth)
        worst = (0.0, 1.0, 0.0, "")
        for i, (a, r, b) in enumerate(zip(got, ref32, truth)):
            n = b.numel() ** 0.5
            d2, dm = max(float(b.norm()), 1e-2 * rms_scale * n, 1e-30), max(float(b.abs().max()), 1e-2 * max_scale, 1e-30)
            h2, hm = float((a - b).norm()) / d2, float((a - b).abs().max()) / dm
            r2, rm = float((r - b).norm()) / d2, float((r - b).abs().max()) / dm
            t2, tm = max(contract_l2, factor * r2), max(contract_max, factor * rm)
            tag = f"{name}: {names[i] if names else i}"
            if known and names and names[i] in known:   # a documented, measured accuracy gap: its own (looser) bound, reported as such
                t2, tm = known[names[i]]
                self.rows.append((tag + f" KNOWN GAP (rel L2) [reference fp32: {r2:.2e}]", h2, t2, h2 <= t2))
            if not (h2 <= t2 and hm <= tm and math.isfinite(h2) and math.isfinite(hm)):
                self.failed.append((tag, (h2, hm), (t2, tm)))
                self.rows.append((tag + f" (rel L2) [reference fp32: {r2:.2e}]", h2, t2, h2 <= t2))
                self.rows.append((tag + f" (max) [reference fp32: {rm:.2e}]", hm, tm, hm <= tm))
            if max(h2 / t2, hm / tm) > worst[0] / worst[1]:
                worst = (max(h2, hm), t2 if h2 / t2 >= hm / tm else tm, max(r2, rm), tag)
        self.rows.append((f"{name}: {len(got)} tensors vs fp64 truth, tightest [{worst[3]}; reference fp32 there: {worst[2]:.2e}]",
                          worst[0], worst[1], worst[0] <= worst[1]))

    def check_grads(self, name, got, want, names=None, tol_l2=GRAD_TOL_L2, tol_max=GRAD_TOL_MAX):
        """Element-wise gradient comparison, tensor by tensor (per-parameter L2 NORMS pass a permuted or
        sign-flipped gradient).  Per tensor: |got - want|_2 / |want|_2 and max|got - want| / max|want|, each against the
        tensor's own size but never below 1e-2 of the network's gradient scale (the largest per-tensor value over all
        parameters): a tensor whose exact gradient is zero -- the bias of a convolution feeding a training-mode BatchNorm --
        holds only rounding noise on both sides (measured: 2.5e-6 of the network's scale at batch 250)."""
        got = [g.detach().double().cpu() for g in got]
        want = [w.detach().double().cpu() for w in want]
        assert len(got) == len(want) and all(a.shape == b.shape for a, b in zip(got, want)), "gradient lists differ in shape"
        rms_scale = max(float(b.norm()) / max(b.numel(), 1) ** 0.5 for b in want)
        max_scale = max(float(b.abs().max()) for b in want)
        worst = (0.0, 0.0, "")
        for i, (a, b) in enumerate(zip(got, want)):
            n = b.numel() ** 0.5
            e2 = float((a - b).norm()) / max(float(b.norm()), 1e-2 * rms_scale * n, 1e-30)
            em = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-2 * max_scale, 1e-30)
            tag = f"{name}: {names[i] if names else i}"
            ok = e2 < tol_l2 and em < tol_max and math.isfinite(e2) and math.isfinite(em)
            if not ok:
                self.failed.append((tag, (e2, em), (tol_l2, tol_max)))
                self.rows.append((tag + " (rel L2)", e2, tol_l2, e2 < tol_l2))
                self.rows.append((tag + " (max)", em, tol_max, em < tol_max))
            if max(e2 / tol_l2, em / tol_max) > max(worst[0] / tol_l2, worst[1] / tol_max):
                worst = (e2, em, tag)
        self.rows.append((f"{name}: {len(got)} tensors element-wise, worst rel L2 [{worst[2]}]", worst[0], tol_l2, worst[0] < tol_l2))
        self.rows.append((f"{name}: worst max-norm", worst[1], tol_max, worst[1] < tol_max))

    def finish(self):
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(f"== {self.title}\n")
            for name, err, tol, ok in self.rows:
                f.write(f"{'ok  ' if ok else 'FAIL'} {name:70s} err={err:.3e} tol={tol:.1e}\n")
        assert not self.failed, f"{self.title}: {len(self.failed)} mismatches, first: {self.failed[:6]}"


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import ot_vae_lightning_amd as pkg
    from ot_vae_lightning_amd import _lib
    _lib.load()
    return pkg


def cuda(t):
    return t.cuda() if isinstance(t, torch.Tensor) else t


# ------------------------------------------------------------------------------------------------ ABI / device
def test_library_and_device(A):
    import ctypes as C
    from ot_vae_lightning_amd import _lib
    lib = _lib.load()
    ncu, wave = C.c_int(0), C.c_int(0)
    arch = C.create_string_buffer(64)
    assert lib.otvae_device_info(C.byref(ncu), C.byref(wave), arch, 64) == 0
    assert wave.value == 64
    assert b"gfx950" in arch.value, arch.value
    # a bad geometry is refused with ValueError, nothing launched
    from ot_vae_lightning_amd import functional as HF
    x = torch.zeros(2, 3, 8, 8, device="cuda")
    w = torch.zeros(4, 5, 3, 3, device="cuda")
    with pytest.raises(ValueError):
        HF.conv_layers(x, [dict(weight=w, bias=None, stride=1, pad=1, up=1, relu=False)])
    with pytest.raises(RuntimeError):
        HF.conv_layers(torch.zeros(2, 3, 8, 8), [dict(weight=w.cpu(), bias=None, stride=1, pad=1, up=1, relu=False)])


# ------------------------------------------------------------------------------------------------ G1 ConvLayer
from test_oracle_vs_golden import CONV_GEOM  # noqa: E402

CONV_CTOR = {
    "enc_first": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="relu")),
    "enc_same": ("ConvLayer", dict(normalization="batchnorm", activation="relu")),
    "enc_down": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="relu")),
    "enc_last": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="relu")),
    "same_1x1res": ("ConvLayer", dict(normalization="batchnorm", activation="relu")),
    "dec_up": ("ConvLayer", dict(up_sample=2, normalization="batchnorm", activation="relu")),
    "dec_first": ("ConvLayer", dict(up_sample=2, normalization="batchnorm", activation="relu")),
    "dec_last": ("ConvLayer", dict(up_sample=2, normalization="batchnorm", activation="relu")),
    "dec_11": ("ConvLayer", dict(normalization="batchnorm", activation="relu")),
    "rgb_in": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="relu")),
    "qkv": ("Conv1x1", dict(normalization="batchnorm")),
    "qkv1": ("Conv1x1", dict(normalization="batchnorm")),
    "proj": ("Conv1x1", dict()),
    "skip_down": ("Conv1x1", dict(down_sample=2, normalization="batchnorm")),
    "skip_up": ("Conv1x1", dict(up_sample=2, normalization="batchnorm")),
    "skip_up1": ("Conv1x1", dict(up_sample=2, normalization="batchnorm")),
    "nonorm_relu": ("ConvLayer", dict(activation="relu")),
    # configs/vae/defaults_imagenet.yaml:26-27 (leaky + equalized_lr) and the other activations of cnn.py:128-147
    "leaky_eq": ("ConvLayer", dict(normalization="batchnorm", activation="leaky", equalized_lr=1.)),
    "leaky_down_eq2": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="leaky_relu", equalized_lr=2.)),
    "selu_down": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="selu")),
    "gelu_up": ("ConvLayer", dict(up_sample=2, normalization="batchnorm", activation="gelu")),
    "silu_nonorm": ("ConvLayer", dict(activation="silu")),
    "swish_bn_1ch": ("ConvLayer", dict(normalization="batchnorm", activation="swish")),
    "eq_1x1": ("Conv1x1", dict(normalization="batchnorm", equalized_lr=0.5)),
    "gn_relu": ("ConvLayer", dict(normalization="groupnorm", activation="relu")),
    "gn_down_leaky": ("ConvLayer", dict(down_sample=2, normalization="groupnorm", activation="leaky")),
    "gn_1x1_up": ("Conv1x1", dict(up_sample=2, normalization="groupnorm")),
    "in_silu": ("ConvLayer", dict(normalization="instancenorm", activation="silu")),
    "in_relu_up": ("ConvLayer", dict(up_sample=2, normalization="instancenorm", activation="relu")),
    "film_relu": ("ConvLayer", dict(normalization="batchnorm", activation="relu", additional_embed=5)),
    "film_leaky_eq": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="leaky", equalized_lr=2., additional_embed=6)),
    "film_1x1_gn": ("Conv1x1", dict(normalization="groupnorm", additional_embed=5)),
    # grouped / dilated layers (cnn.py:66-67,103-104) through the weight expansion of csrc/weight_expand.hip
    "grp2_relu": ("ConvLayer", dict(normalization="batchnorm", activation="relu", groups=2)),
    "grp4_down_leaky_eq": ("ConvLayer", dict(down_sample=2, normalization="batchnorm", activation="leaky", equalized_lr=2., groups=4)),
    "grp2_1x1_gn": ("Conv1x1", dict(normalization="groupnorm", groups=2)),
    "dil2_relu": ("ConvLayer", dict(normalization="batchnorm", activation="relu", dilation=2, padding=2)),
    "dil3_grp2_up": ("ConvLayer", dict(up_sample=2, normalization="batchnorm", activation="relu", dilation=3, groups=2)),
    "dil2_nobias_silu": ("ConvLayer", dict(activation="silu", dilation=2, groups=3, bias=False)),
    # user-supplied resampling modules (cnn.py:97,106) around the layer's kernels
    "mod_up_bilinear": ("ConvLayer", dict(up_sample=torch.nn.Upsample(scale_factor=2, mode="bilinear"), normalization="batchnorm", activation="relu")),
    "mod_down_avgpool": ("ConvLayer", dict(down_sample=torch.nn.AvgPool2d(2), normalization="batchnorm", activation="leaky")),
    "up4_relu": ("ConvLayer", dict(up_sample=4, normalization="batchnorm", activation="relu")),
    # round 4: strides other than 1 / 2, footprints beyond 7 x 7 (csrc/conv_generic.hip)
    "down4_relu": ("ConvLayer", dict(down_sample=4, normalization="batchnorm", activation="relu")),
    "down4_skip": ("Conv1x1", dict(down_sample=4, normalization="batchnorm")),
    "down8_leaky": ("ConvLayer", dict(down_sample=8, activation="leaky")),
    "stride3_k5": ("ConvLayer", dict(kernel_size=5, stride=3, padding=2, normalization="batchnorm", activation="relu")),
    "k9_same_relu": ("ConvLayer", dict(kernel_size=9, padding=4, activation="relu")),
    "dil4_grp2_silu": ("ConvLayer", dict(normalization="batchnorm", activation="silu", dilation=4, padding=4, groups=2)),
    "up2_k9": ("ConvLayer", dict(up_sample=2, kernel_size=9, padding=4, normalization="batchnorm", activation="relu")),
}


def test_conv_layer_vs_reference_golden(A):
    z = load_golden("convlayer.npz")
    rep = Report("ConvLayer fwd/bwd vs reference golden (B=4)")
    for name in sorted(CONV_CTOR):
        g = group(z, name)
        cls, kw = CONV_CTOR[name]
        cin, cout = g["x"].shape[1], g["y"].shape[1]
        layer = getattr(A, cls)(cin, cout, **kw)
        layer.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param/")}, strict=False)
        layer = layer.cuda().train()
        x = g["x"].cuda().requires_grad_(True)
        emb = g["embed"].cuda().requires_grad_(True) if "embed" in g else None
        y = layer(x, emb) if emb is not None else layer(x)
        y.backward(g["gy"].cuda())
        rep.check(f"{name}/y", y, g["y"])
        rep.check(f"{name}/gx", x.grad, g["gx"])
        if emb is not None:
            rep.check(f"{name}/gembed", emb.grad, g["gembed"])
        for k, p in layer.named_parameters():
            rep.check(f"{name}/grad/{k}", p.grad, g[f"grad/{k}"])
        for k, b in layer.named_buffers():
            if k.endswith("num_batches_tracked"):
                rep.check(f"{name}/buf/{k}", b, g[f"buf/{k}"], exact=True)
            else:
                rep.check(f"{name}/buf/{k}", b, g[f"buf/{k}"])
    rep.finish()


# ------------------------------------------------------------------------------------------------ G2 attention
def test_attention_vs_reference_golden(A):
    z = load_golden("attention.npz")
    rep = Report("QKVAttention fwd/bwd vs reference golden (B=2)")
    for key in sorted({k.split("/")[0] for k in z.files}):
        g = group(z, key)
        heads = int(key.split("_")[1][1:])
        attn = A.QKVAttention(heads)
        qkv = g["qkv"].cuda().requires_grad_(True)
        out = attn(qkv)
        out.backward(g["gout"].cuda())
        rep.check(f"{key}/out", out, g["out"])
        rep.check(f"{key}/gqkv", qkv.grad, g["gqkv"])
    rep.finish()


# ------------------------------------------------------------------------------------------------ G3 small CNN
def _cnn_small_truth(nm, cap, res, g):
    """float64 parameter gradients of the small CNN (oracle, CPU) on the golden's input / output gradient and the same weight fill"""
    import otvae_oracle as O
    if nm == "enc":
        arch = O.cnn_arch(1, 16, 16, 1, capacity=cap, down_sample=True, residual=res)
        shell = __import__("ot_vae_lightning_amd").CNN(1, 16, 16, 1, capacity=cap, down_sample=True, residual=res)
    else:
        arch = O.cnn_arch(8, 1, 1, 16, capacity=cap, up_sample=True, residual=res)
        shell = __import__("ot_vae_lightning_amd").CNN(8, 1, 1, 16, capacity=cap, up_sample=True, residual=res)
    sd = shell.state_dict()
    fill_state_dict(sd)
    names = [k for k, _ in shell.named_parameters()]
    p64 = {k: v.detach().double().contiguous().clone() for k, v in sd.items()}
    for k in names:
        p64[k].requires_grad_(True)
    y = O.cnn_forward(g["x"].double(), p64, arch)
    y.backward(g["gy"].double())
    return {k: p64[k].grad for k in names}


@pytest.mark.parametrize("residual", ["add", "None", "cat"])
def test_cnn_small_vs_reference_golden(A, residual):
    z = load_golden("cnn_small.npz")
    res = None if residual == "None" else residual
    cap = 4 if res == "cat" else 2
    rep = Report(f"CNN capacity {cap} residual={residual} vs reference golden")
    nets = [("enc", lambda: A.CNN(1, 16, 16, 1, capacity=cap, down_sample=True, residual=res))]
    if res != "cat":
        nets.append(("dec", lambda: A.CNN(8, 1, 1, 16, capacity=cap, up_sample=True, residual=res)))
    for nm, make in nets:
        g = group(z, f"{residual}/{nm}")
        net = make()
        fill_state_dict(net.state_dict())
        net = net.cuda().train()
        x = g["x"].cuda().requires_grad_(True)
        y = net(x)
        y.backward(g["gy"].cuda())
        rep.check(f"{nm}/y", y, g["y"])
        rep.check(f"{nm}/gx", x.grad, g["gx"], tol=2e-4)
        gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith("grad/"))
        # fp64 truth of every parameter gradient from the oracle on the same deterministic weights: the bound is the fp32 contract
        # or 1.5x the REFERENCE's own fp32 error (the golden), not a constant (round 3: 5e-4 against a measured 4.943e-4)
        truth = _cnn_small_truth(nm, cap, res, g)
        for k, p in net.named_parameters():
            # contract: the standing 5e-4 of these per-layer gradient checks -- but against the TRUTH, where the worst tensor measures
            # 1.3e-4 (the 4.9e-4 of round 3 was mostly the golden's own fp32 error: 4.9e-5 ... 2.0e-4 from the truth)
            rep.check_vs_truth(f"{nm}/grad/{k}", p.grad, g[f"grad/{k}"], truth[k], floor=3e-3 * gscale, contract=5e-4)
        for k, b in net.named_buffers():
            if not k.endswith("num_batches_tracked"):
                rep.check(f"{nm}/buf/{k}", b, g[f"buf/{k}"])
    rep.finish()


def test_conv_layer_dropout2d_matches_torch_given_its_own_mask(A):
    """``ConvLayer(dropout=p)`` = nn.Dropout2d behind the convolution (reference cnn.py:112,191): whole (sample, channel) maps are
    dropped.  The mask is a hash the backward recomputes; parity: the layer's output and gradients against the dropout-free layer
    multiplied by the kernel's own mask / (1 - p); the drop rate; a fresh mask per call; evaluation mode drops nothing."""
    from ot_vae_lightning_amd import functional as HF
    torch.manual_seed(3)
    p = 0.3
    layer = A.ConvLayer(8, 16, normalization="batchnorm", activation="relu", dropout=p).cuda().train()
    plain = A.ConvLayer(8, 16, normalization="batchnorm", activation="relu").cuda().train()
    plain.load_state_dict(layer.state_dict())
    x = normal((64, 8, 8, 8), 5).cuda()
    gy = normal((64, 16, 8, 8), 6).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya = layer(xa)
    y_ref = plain(xb)
    # recover the mask from the output itself (kept maps equal y_ref / (1 - p) exactly in fp32 up to one rounding; dropped ones are 0)
    kept = (ya.detach().abs().sum((2, 3)) > 0)
    rate = 1.0 - float(kept.float().mean())
    assert abs(rate - p) < 0.06, rate
    want = y_ref * kept[:, :, None, None] / (1 - p)
    assert rel_err(ya.detach().cpu(), want.detach().cpu()) < 1e-6
    ya.backward(gy)
    want.backward(gy)
    assert rel_err(xa.grad.cpu(), xb.grad.cpu()) < 1e-5
    for (ka, pa), (kb, pb) in zip(layer.named_parameters(), plain.named_parameters()):
        assert rel_err(pa.grad.cpu(), pb.grad.cpu()) < 1e-5, ka
    # the exported mask agrees with what the output shows; a second call draws another one
    y2, used = HF.dropout2d(y_ref.detach(), p, HF.new_dropout_key(x.device, seed=11), return_used=True)
    m2 = HF.dropout2d_mask(used, 64, 16, p)
    assert torch.equal(m2, y2.abs().sum((2, 3)) > 0)
    kept2 = (layer(x).detach().abs().sum((2, 3)) > 0)
    assert not torch.equal(kept, kept2)
    plain.load_state_dict(layer.state_dict())   # (the layers have seen different numbers of training batches by now)
    layer.eval()
    plain.eval()
    assert torch.equal(layer(x), plain(x))


CNN_VARIANTS = {   # oracle/gen_golden.py: CNN_VARIANTS (constructor corners no config of the reference uses)
    "cat_gn_silu_3layers": ((3, 12, 8, 2), dict(capacity=4, down_sample=True, residual="cat", n_layers=3, normalization="groupnorm", activation="silu")),
    "cat_gn_silu_3layers_cap8": ((3, 16, 8, 2), dict(capacity=8, down_sample=True, residual="cat", n_layers=3, normalization="groupnorm", activation="silu")),
    "add_film_gelu_eq_up": ((6, 3, 2, 8), dict(capacity=4, up_sample=True, residual="add", additional_embed=5, activation="gelu", equalized_lr=1.0)),
    "intermediate_nonorm_selu": ((4, 8), dict(intermediate_features=[6, 10], residual=None, normalization=None, activation="selu")),
    "intermediate_nonorm_selu_res": ((4, 8, 4, 4), dict(intermediate_features=[6, 10], residual=None, normalization=None, activation="selu")),
    "add_noattn_dilated_res": ((4, 4, 8, 8), dict(intermediate_features=[8], residual="add", max_attn_res=0, dilation=2, padding=2)),
    "grouped_in_leaky_nobias": ((2, 8, 8, 2), dict(capacity=4, down_sample=True, residual="add", groups=2, normalization="instancenorm", activation="leaky", bias=False)),
    "cat_1layer_k1_noattn_up": ((8, 2, 2, 8), dict(capacity=4, up_sample=True, residual="cat", max_attn_res=1, n_layers=1, kernel_size=1, padding=0)),
    "add_noattn_dilated": ((4, 4), dict(intermediate_features=[8], residual="add", max_attn_res=0, dilation=2, padding=2)),
    "down4_add": ((2, 16, 16, 1), dict(capacity=4, down_sample=4, residual="add")),
    "up4_add": ((8, 2, 1, 16), dict(capacity=4, up_sample=4, residual="add")),
}


@pytest.mark.parametrize("name", sorted(CNN_VARIANTS))
def test_cnn_constructor_corners_vs_reference_golden(A, name):
    """Whole CNNs at corners of the constructor space against the reference's own class (cnn_variants.npz): residual "cat" with
    three layers per block under GroupNorm + SiLU, FiLM embeddings with equalized_lr on the up-sampling path, `intermediate_features`,
    grouped InstanceNorm blocks without biases, one-layer 1x1 blocks, dilated blocks without attention; where the reference's
    constructor raises, this one raises the same exception type."""
    g = group(load_golden("cnn_variants.npz"), name)
    args, kw = CNN_VARIANTS[name]
    if "error" in g:
        want = bytes(g["error"].numpy().astype("uint8")).decode()
        with pytest.raises(Exception) as info:
            A.CNN(*args, **kw)
        assert type(info.value).__name__ == want, (type(info.value).__name__, want)
        return
    net = A.CNN(*args, **kw)
    fill_state_dict(net.state_dict())
    net = net.cuda().train()
    x = g["x"].cuda().requires_grad_(True)
    emb = g["embed"].cuda().requires_grad_(True) if "embed" in g else None
    y = net(x, emb) if emb is not None else net(x)
    y.backward(g["gy"].cuda())
    rep = Report(f"CNN corner `{name}` vs reference golden")
    rep.check("y", y, g["y"])
    rep.check("gx", x.grad, g["gx"], tol=3e-4)
    if emb is not None:
        rep.check("gembed", emb.grad, g["gembed"], tol=3e-4)
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith("grad/"))
    names = {k for k, _ in net.named_parameters()}
    assert names == {k[5:] for k in g if k.startswith("grad/")}, "parameter names differ from the reference's"
    for k, p in net.named_parameters():
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        rep.check(f"grad/{k}", got, g[f"grad/{k}"], tol=5e-4, floor=1e-2 * gscale)
    for k, b in net.named_buffers():
        if not k.endswith("num_batches_tracked"):
            rep.check(f"buf/{k}", b, g[f"buf/{k}"])
    rep.finish()


def test_conv_block_dropout_sits_before_the_residual_sum(A):
    """`ConvBlock(residual="add", dropout=p)` without attention: the reference sums `dropout(conv(...)) + skip(x)`
    (cnn.py:183-192,331-335), so a (sample, channel) map the last layer's Dropout2d removed shows the skip branch alone -- not zero,
    which is what fusing the sum into the convolution in front of the dropout (rounds 1-2's shortcut for this case) gave."""
    torch.manual_seed(5)
    block = A.ConvBlock(8, 8, n_attn_heads=0, n_layers=2, residual="add", dropout=0.4).cuda().train()
    x = normal((32, 8, 8, 8), 3).cuda()
    out = block(x).detach()
    skip = block.skip(x).detach()                 # training-mode BatchNorm statistics depend on x only: the same branch value
    diff = (out - skip).abs().sum((2, 3))
    dropped = diff == 0
    rate = float(dropped.float().mean())
    assert 0.25 < rate < 0.55, rate               # the last layer's maps vanish at about p; the sum does not
    assert float(out.abs().sum((2, 3))[dropped].min()) > 0
    block.eval()
    a = block(x)
    plain = A.ConvBlock(8, 8, n_attn_heads=0, n_layers=2, residual="add").cuda().eval()
    plain.load_state_dict(block.state_dict())
    assert torch.equal(a, plain(x))


def test_cnn_leaky_equalized_lr_vs_reference_golden(A):
    """a whole encoder / decoder with ``activation="leaky", equalized_lr=1.`` (the reference's
    configs/vae/defaults_imagenet.yaml:26-27) against the reference's own CNN: outputs, input gradient, every parameter gradient
    and the BatchNorm running buffers."""
    z = load_golden("cnn_small_opts.npz")
    rep = Report("CNN capacity 2, residual=add, activation=leaky, equalized_lr=1 vs reference golden")
    kw = dict(capacity=2, residual="add", activation="leaky", equalized_lr=1.0)
    for nm, make in (("enc", lambda: A.CNN(1, 16, 16, 1, down_sample=True, **kw)), ("dec", lambda: A.CNN(8, 1, 1, 16, up_sample=True, **kw))):
        g = group(z, nm)
        net = make()
        fill_state_dict(net.state_dict())
        net = net.cuda().train()
        x = g["x"].cuda().requires_grad_(True)
        y = net(x)
        y.backward(g["gy"].cuda())
        rep.check(f"{nm}/y", y, g["y"])
        rep.check(f"{nm}/gx", x.grad, g["gx"], tol=2e-4)
        gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith("grad/"))
        for k, p in net.named_parameters():
            # floor 1e-2 of the network's gradient scale: the biases that feed a training-mode BatchNorm have an exactly zero
            # gradient; with the equalized_lr multipliers their rounding noise measured 5e-6 of that scale
            rep.check(f"{nm}/grad/{k}", p.grad, g[f"grad/{k}"], tol=5e-4, floor=1e-2 * gscale)
        for k, b in net.named_buffers():
            if not k.endswith("num_batches_tracked"):
                rep.check(f"{nm}/buf/{k}", b, g[f"buf/{k}"])
    rep.finish()


# ------------------------------------------------------------------------------------------------ G4 nelbo
def _mnist_vae(A, residual, loss_coeff=0.1):
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual=residual)
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual=residual)
    fill_state_dict(enc.state_dict())
    fill_state_dict(dec.state_dict())
    return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=loss_coeff)).cuda().train()


def test_nelbo_with_expansion_vs_reference_golden(A):
    """`VAE(expansion=3)` (model/vae.py:158-229; utils.replicate_batch / mean_replicated_batch): three noise draws per image, the decoder
    on 3B latents, the reconstruction loss on the mean over the replicas, `preds` / `latents` = the first replica -- loss vector,
    artifacts, every parameter gradient and BatchNorm buffer against the reference's own classes (nelbo_expansion.npz); then the same
    model through HipTrainer (captured step = eager step)."""
    z = load_golden("nelbo_expansion.npz")
    g = {k: torch.from_numpy(z[k]) for k in z.files}

    def make():
        enc = A.CNN(1, 16, 16, 1, capacity=4, down_sample=True, residual="add")
        dec = A.CNN(8, 1, 1, 16, capacity=4, up_sample=True, residual="add")
        fill_state_dict(enc.state_dict())
        fill_state_dict(dec.state_dict())
        return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1), expansion=3).cuda().train()

    model = make()
    x, eps = g["x"].cuda(), g["eps"].cuda()
    loss, logs, art = model.nelbo({"samples": x, "target": x, "kwargs": {"eps": eps}}, 0)
    loss.backward()
    rep = Report("VAE(expansion=3).nelbo vs reference golden (B=8)")
    rep.check("loss [total, recon, prior]", torch.stack([logs["train/loss/total"], logs["train/loss/recon"], logs["train/loss/prior"]]), g["loss"])
    rep.check("preds", art["preds"], g["preds"])
    rep.check("latents", art["latents"], g["latents"])
    rep.check("preds_mean", art["preds_mean"], g["preds_mean"])
    gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith("grad/"))
    for pre, net in (("encoder.", model.encoder), ("decoder.", model.decoder)):
        for k, p in net.named_parameters():
            rep.check(f"grad/{pre}{k}", p.grad, g[f"grad/{pre}{k}"], tol=5e-4, floor=1e-2 * gscale)
        for k, b in net.named_buffers():
            if not k.endswith("num_batches_tracked"):
                rep.check(f"buf/{pre}{k}", b, g[f"buf/{pre}{k}"])
    rep.finish()
    # the trainer sizes its noise buffer for the replicas; captured and eager steps agree
    ta = A.HipTrainer(make(), batch_shape=(8, 1, 16, 16), use_graph=True)
    tb = A.HipTrainer(make(), batch_shape=(8, 1, 16, 16), use_graph=False)
    assert ta.eps.shape[0] == 24
    for _ in range(2):
        assert torch.equal(ta.step(x, eps), tb.step(x, eps))
    assert torch.equal(ta.pflat, tb.pflat)
    ta.close(); tb.close()


@pytest.mark.parametrize("residual", ["add", "None"])
def test_nelbo_and_adam_vs_reference_golden(A, residual):
    g = group(load_golden("nelbo_mnist.npz"), residual)
    res = None if residual == "None" else residual
    rep = Report(f"VAE.nelbo + Adam (MNIST cfg, B=6, residual={residual}) vs reference golden")
    model = _mnist_vae(A, res)
    assert model.latent_size == torch.Size((128, 1, 1))
    x, eps = mnist_like(6, 42).cuda(), normal((6, 128, 1, 1), 43).cuda()
    trainer = A.HipTrainer(model, batch_shape=(6, 1, 32, 32), use_graph=False)
    out = trainer.step(x, eps)
    torch.cuda.synchronize()
    # forward quantities: 1e-4, or 3x the reference's own 1-ulp sensitivity where that is larger (hostile case, see below)
    def fwd_tol(key):
        return max(TOL32, 3.0 * float(g[key + "_spread"]) / max(g[key].abs().max().item(), 1e-30))

    rep.check("loss[total,recon,prior]", out, g["loss"], tol=fwd_tol("loss"))
    logs = trainer._logs
    names, params = [], []
    for pre, net in (("encoder.", model.encoder), ("decoder.", model.decoder)):
        for k, p in net.named_parameters():
            names.append(pre + k)
            params.append(p)
    assert names == [str(s) for s in g["param_names"]]
    # Gradients through BatchNorm over 6 samples at 1x1 resolution with closed-form sine weights (|dL/dW| up to 9e3) are
    # a deliberately hostile, badly conditioned case.  The golden file records how much the REFERENCE's own fp32 gradients
    # move (a) when the same computation runs in fp64 and (b) when the input is perturbed by one ulp (4 draws); a port is
    # held to that resolution: |ours - ref32| <= max(3e-4 * scale, 5 * fp64 discrepancy, 3 * one-ulp spread).
    # (tests/diag_nelbo.py prints the per-parameter picture; the well-conditioned batch-256 step below holds 3e-4.)
    def check_grad(name, got, ref32, ref64, spread, floor=0.0):
        denom = max(ref32.double().abs().max().item(), floor, 1e-30)
        noise = (ref32.double() - ref64.double()).abs().max().item() / denom
        spr = float(torch.as_tensor(spread).double().abs().max().item()) / denom
        tol = max(3e-4, 5.0 * noise, 3.0 * spr)
        e32 = (got.detach().double().cpu() - ref32.double()).abs().max().item() / denom
        ok = e32 < tol
        rep.rows.append((f"{name} [fp64 discrepancy {noise:.2e} | 1-ulp spread {spr:.2e}]", e32, tol, ok))
        if not ok:
            rep.failed.append((name, e32, tol))

    gl2 = torch.tensor([p.grad.double().norm().item() for p in params])
    check_grad("grad_l2 (all parameters)", gl2, g["grad_l2"], g["grad_l2_f64"], g["grad_l2_spread"])
    gmax = max(v.abs().max().item() for k, v in g.items() if k.startswith("grad_full/"))
    for k, v in g.items():
        if k.startswith("grad_full/"):
            check_grad(k, params[names.index(k[10:])].grad, v, g["grad_full_f64/" + k[10:]],
                       g["grad_full_spread/" + k[10:]], floor=1e-3 * gmax)
    # Adam's first step moves every weight by lr*g/(|g|+1e-8): for parameters whose exact gradient is zero (biases in
    # front of a BatchNorm) that is a function of rounding noise, in the reference too -> compare the others
    pl2 = torch.tensor([p.double().norm().item() for p in params])
    sig = g["grad_l2"] > 1e-3 * g["grad_l2"].max()
    assert int(sig.sum()) > 40
    # (|update| = lr for every element whatever the gradient magnitude, so single elements whose tiny gradient changes
    # sign between two fp32 evaluations move the norm by O(lr / |p|): 2e-4 covers it)
    rep.check("param_l2 after one Adam step (parameters with a non-zero gradient)", pl2[sig], g["param_l2_after_adam"][sig],
              tol=2e-4)
    rs = [b.double().sum().item() for net in (model.encoder, model.decoder) for k, b in net.named_buffers()
          if k.endswith("running_mean") or k.endswith("running_var")]
    rep.check("BatchNorm running stats (40 layers)", torch.tensor(rs), g["running_stat_sums"])
    # forward artifacts
    model2 = _mnist_vae(A, res)
    loss, logs, art = model2.nelbo({"samples": x, "target": x, "kwargs": {"eps": eps}}, 0)
    rep.check("preds[:2]", art["preds"][:2], g["preds"], tol=fwd_tol("preds"))
    rep.check("latents", art["latents"], g["latents"], tol=fwd_tol("latents"))
    rep.finish()


# ------------------------------------------------------------------------------------------------ G5 prior
@pytest.mark.parametrize("residual", ["add", None])
def test_whole_network_step_batch32_vs_reference_golden(A, residual):
    """The HIP path held to the REFERENCE directly on a well-conditioned whole-network case (tests/golden/nelbo_b32.npz: the
    reference's VAE.nelbo + backward, torch default initialisation under manual_seed(1234), batch 32, explicit eps): losses,
    reconstructions, every parameter's gradient norm and sum and six full gradients at 3e-4."""
    from conftest import load_golden
    G = load_golden("nelbo_b32.npz")
    tag = str(residual)
    rep = Report(f"whole network, batch 32, default init (residual={residual}) vs the reference")
    torch.manual_seed(1234)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual=residual)
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual=residual)
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    named = [(pre + k, p) for pre, net in (("encoder.", model.encoder), ("decoder.", model.decoder)) for k, p in net.named_parameters()]
    assert [k for k, _ in named] == list(G[f"{tag}/param_names"])
    rep.check("initialisation (parameter sums)", torch.tensor([p.detach().double().sum().item() for _, p in named]),
              torch.from_numpy(G[f"{tag}/param_sum"]), tol=1e-6)
    x, eps = mnist_like(32, seed=52).cuda(), normal((32, 128, 1, 1), seed=53).cuda()
    loss, logs, art = model.nelbo({"samples": x, "target": x, "kwargs": {"eps": eps}}, 0)
    loss.backward()
    rep.check("loss[total,recon,prior]", model._last_out3, torch.from_numpy(G[f"{tag}/loss"]))
    rep.check("preds[:2]", art["preds"][:2], torch.from_numpy(G[f"{tag}/preds"]))
    rep.check("grad_l2 (all parameters)", torch.tensor([p.grad.double().norm().item() for _, p in named]),
              torch.from_numpy(G[f"{tag}/grad_l2"]), tol=3e-4)
    rep.check("grad_sum (all parameters)", torch.tensor([p.grad.double().sum().item() for _, p in named]),
              torch.from_numpy(G[f"{tag}/grad_sum"]), tol=3e-4)
    lookup = dict(named)
    for k in G.files:
        if k.startswith(f"{tag}/grad_full/"):
            name = k.split("grad_full/")[1]
            rep.check(f"grad {name}", lookup[name].grad, torch.from_numpy(G[k]), tol=3e-4)
    rep.finish()


@pytest.mark.parametrize("tag", ["plain", "anneal"])
def test_gaussian_prior_vs_reference_golden(A, tag):
    g = group(load_golden("prior.npz"), tag)
    coeff, ann, step = g["cfg"].tolist()
    rep = Report(f"GaussianPrior.forward ({tag}) vs reference golden")
    prior = A.GaussianPrior(loss_coeff=coeff, annealing_steps=int(ann))
    x = g["x"].cuda().requires_grad_(True)
    z, loss, art = prior(x, step=int(step), eps=g["eps"].cuda())
    ((z * g["gz"].cuda()).sum() + (loss * g["gl"].cuda()).sum()).backward()
    rep.check("z", z, g["z"])
    rep.check("loss", loss, g["loss"])
    rep.check("gx", x.grad, g["gx"])
    assert prior.out_size(torch.Size((256, 1, 1))) == torch.Size((128, 1, 1))
    assert rel_err(art["distribution"].mean, g["x"][:, :128]) == 0
    rep.finish()


@pytest.mark.parametrize("tag", ["empirical", "fixed_var", "fixed_var_time", "fixed_var_empirical"])
def test_gaussian_prior_options_vs_reference_golden(A, tag):
    """``GaussianPrior(empirical_kl=, fixed_var=)`` and the temperature of ``encode(time=)`` (reference
    prior/gaussian.py:38-41,63-96; prior/base.py:65-68) on ``otvae_gaussian_prior_ex_fwd / _bwd``."""
    g = group(load_golden("prior.npz"), tag)
    coeff, emp, fixed = g["cfg"].tolist()
    rep = Report(f"GaussianPrior options ({tag}) vs reference golden")
    prior = A.GaussianPrior(loss_coeff=coeff, empirical_kl=bool(emp), fixed_var=bool(fixed))
    x = g["x"].cuda().requires_grad_(True)
    kw = {"time": g["time"].cuda()} if "time" in g else {}
    z, loss, art = prior(x, step=0, eps=g["eps"].cuda(), **kw)
    ((z * g["gz"].cuda()).sum() + (loss * g["gl"].cuda()).sum()).backward()
    rep.check("z", z, g["z"])
    rep.check("loss", loss, g["loss"])
    rep.check("gx", x.grad, g["gx"])
    want = torch.Size((128, 1, 1))
    assert prior.out_size(torch.Size(tuple(g["x"].shape[1:]))) == want
    if "time" in g:
        assert rel_err(art["distribution"].stddev.flatten(1)[:, 0], g["time"] + 1e-8) < 1e-6
    rep.finish()


# ------------------------------------------------------------------------------------------------ G6 sinkhorn
from test_oracle_vs_golden import _sinkhorn_problem  # noqa: E402


def test_sinkhorn_vs_reference_golden(A):
    z = load_golden("sinkhorn.npz")
    rep = Report("sinkhorn_log vs reference golden")
    for name in sorted({k.split("/")[0] for k in z.files}):
        g = group(z, name)
        reg, it, thr = g["cfg"].tolist()
        if "pi" in g:
            a, b, C = g["a"], g["b"], g["C"]
        else:
            seed, n, m = g["seed"].tolist()
            a, b, C = _sinkhorn_problem((), n, m, torch.float32 if "f32" in name else torch.float64, seed)
        pi = A.sinkhorn_log(a.cuda(), b.cuda(), C.cuda(), reg=reg, max_iter=int(it), threshold=thr)
        tol = TOL32 if pi.dtype == torch.float32 else 1e-8
        if "pi" in g:
            rep.check(f"{name}/pi", pi, g["pi"], tol)
        else:
            rep.check(f"{name}/row_sums", pi.sum(-1), g["row_sums"], tol)
            rep.check(f"{name}/col_sums", pi.sum(-2), g["col_sums"], tol)
            rep.check(f"{name}/pi_corner", pi[:8, :8], g["pi_corner"], tol)
        rep.check(f"{name}/cost", A.ot_cost(C.cuda(), pi), g["cost"], tol)
    rep.finish()


# ------------------------------------------------------------------------------------------------ G7 Gaussian OT
@pytest.mark.parametrize("D", [8, 32, 128])
def test_gaussian_transport_vs_reference_golden(A, D):
    z = load_golden("gaussian_ot.npz")
    base = group(z, f"D{D}")
    rep = Report(f"GaussianModel/GaussianTransport D={D} vs reference golden")
    src, tgt = base["src"].cuda(), base["tgt"].cuda()
    for decay in (None, 0.9):
        g = group(z, f"D{D}/decay{decay}")
        cfg = dict(update_decay=decay, dtype=torch.double)
        op = A.GaussianTransport(D, source_cfg=cfg, target_cfg=cfg,
                                 transport_cfg=dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True,
                                                    verbose=False, dtype=torch.double)).cuda()
        op.reset()
        for s, t in zip(src, tgt):
            op.update(source_samples=s, target_samples=t)
        sm = op.source_model
        rep.check(f"decay{decay}/n_obs", sm._n_obs, g["src_n"], 1e-12)
        rep.check(f"decay{decay}/running_sum", sm._running_sum, g["src_sum"], 1e-12)
        rep.check(f"decay{decay}/running_sum_cov", sm._running_sum_cov, g["src_sumcov"], 1e-12)
        w = op.compute()
        rep.check(f"decay{decay}/w2", w, g["w2"], 1e-8)
        rep.check(f"decay{decay}/src_mean", op.source_model.mean, g["src_mean"], 1e-12)
        rep.check(f"decay{decay}/src_cov", op.source_model.cov, g["src_cov"], 1e-10)
        rep.check(f"decay{decay}/tgt_cov", op.target_model.cov, g["tgt_cov"], 1e-10)
        rep.check(f"decay{decay}/T", op.transport_operator, g["T"], 1e-7)
        rep.check(f"decay{decay}/transported", op.transport(src[0][:16]), g["transported"], 1e-5)
    n = float(src[0].shape[0])
    sx, sxx = src[0].double().sum(0), src[0].double().T @ src[0].double()
    mean, cov = A.mean_cov(sx, sxx, n)
    rep.check("mean_cov/mean", mean, base["meancov_mean"], 1e-12)
    rep.check("mean_cov/cov", cov, base["meancov_cov"], 1e-12)
    rep.check("sqrtm(cov)", A.sqrtm(cov), base["sqrtm_cov"], 1e-8)
    rep.check("invsqrtm(cov + 1e-8 I)", A.invsqrtm(cov + 1e-8 * torch.eye(D, device="cuda", dtype=torch.double)),
              base["invsqrtm_cov"], 1e-6)
    tmean = tgt[0].double().mean(0)
    _, tcov = A.mean_cov(tgt[0].double().sum(0), tgt[0].double().T @ tgt[0].double(), n)
    rep.check("w2_gaussian(plain)", A.w2_gaussian(mean, tmean, cov, tcov, make_pd=True), base["w2_plain"], 1e-8)
    rep.finish()


def test_w2_known_answers(A):
    """The reference's own known-answer tests (tests/test_w2_utils.py:35-41, 179-195)."""
    g = group(load_golden("gaussian_ot.npz"), "batched")
    rep = Report("w2_gaussian batched + self distance")
    m1, m2, c1, c2 = (g[k].cuda() for k in ("m1", "m2", "c1", "c2"))
    w = A.w2_gaussian(m1, m2, c1, c2)
    assert w.shape == (2, 3)
    rep.check("w2 batched [2,3]", w, g["w2"], 1e-8)
    w0 = A.w2_gaussian(m1, m1, c1, c1)
    assert w0.abs().max().item() < 1e-6
    with pytest.raises(ValueError):
        A.w2_gaussian(m1, m2, c1 - 5 * torch.eye(3, device="cuda"), c2, make_pd=False)
    rep.finish()


# ------------------------------------------------------------------------------------------------ G8 codebook
@pytest.mark.parametrize("tag", ["flat", "multi"])
def test_codebook_indices_bit_exact(A, tag):
    g = group(load_golden("codebook.npz"), tag)
    enc, idx = A.codebook_assign(g["x"].cuda(), g["codebook"].cuda())
    assert torch.equal(idx.cpu(), g["indices"]), "latent indices must be bit-exact"
    assert torch.equal(enc.cpu(), g["preds"])


# ------------------------------------------------------------------------------------------------ vs oracle, larger
def test_training_step_vs_oracle_batch256_and_graph(A):
    """MNIST test config (residual=add, default torch init, seed 0) at batch 256: one GPU step (eager and hipGraph
    replay) vs the CPU oracle on the same weights/batch/eps; then the replayed graph must train."""
    import otvae_oracle as O
    rep = Report("training step B=256 vs CPU oracle; graph replay vs eager")
    B = 256
    x, eps = mnist_like(B, 7), normal((B, 128, 1, 1), 8)

    def make():
        torch.manual_seed(0)
        enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
        dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
        return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1))

    model = make()
    ea = O.cnn_arch(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    da = O.cnn_arch(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    enc = {k: v.detach().clone().contiguous() for k, v in model.encoder.state_dict().items()}
    dec = {k: v.detach().clone().contiguous() for k, v in model.decoder.state_dict().items()}
    leaves = []
    for d in (enc, dec):
        for k, v in d.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
                leaves.append(v)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    r = O.vae_nelbo(x, eps, enc, dec, ea, da, loss_coeff=0.1)
    r["loss"].backward()
    want_loss = torch.stack([r["loss"], r["recon"], r["prior"]]).detach()
    want_gl2 = torch.tensor([v.grad.double().norm().item() for v in leaves])

    model = model.cuda().train()
    tr = A.HipTrainer(model, batch_shape=(B, 1, 32, 32), use_graph=False)
    out = tr.step(x.cuda(), eps.cuda()).clone()
    params = [p for net in (model.encoder, model.decoder) for p in net.parameters()]
    rep.check("loss", out, want_loss)
    rep.check("grad_l2", torch.tensor([p.grad.double().norm().item() for p in params]), want_gl2, tol=3e-4)
    pnames = [f"{net}.{k}" for net, m_ in (("encoder", model.encoder), ("decoder", model.decoder)) for k, _ in m_.named_parameters()]
    rep.check_grads("gradients B=256", [p.grad for p in params], [v.grad for v in leaves], pnames)
    p_eager = tr.pflat.clone()

    model_g = make().cuda().train()
    trg = A.HipTrainer(model_g, batch_shape=(B, 1, 32, 32), use_graph=True)
    outg = trg.step(x.cuda(), eps.cuda()).clone()
    torch.cuda.synchronize()
    rep.check("graph vs eager: loss", outg, out, tol=1e-6)
    rep.check("graph vs eager: parameters after Adam", trg.pflat, p_eager, tol=1e-6)
    for k, b in model_g.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(b) == 1, (k, int(b))
    first = float(outg[1])
    for _ in range(40):
        last = trg.step(x.cuda(), eps.cuda())
    rep.rows.append(("recon loss after 41 replayed steps on a fixed batch", float(last[1]), first, True))
    assert float(last[1]) < 0.8 * first, (first, float(last[1]))
    rep.finish()


def test_sinkhorn_full_size_properties_and_oracle(A):
    """Config 3 size (1024x1024, eps=0.05, 50 iterations): marginal property + agreement with the CPU oracle."""
    import otvae_oracle as O
    rep = Report("sinkhorn 1024x1024 fp32: properties and oracle")
    z, p = normal((1024, 128), 11), normal((1024, 128), 12)
    C = A.sq_euclidean_cost(z.cuda(), p.cuda())
    rep.check("sq_euclidean_cost vs oracle", C, O.sq_euclidean_cost(z, p))
    a = torch.full((1024,), 1 / 1024)
    Cn = C / C.max()
    pi, u, v, iters = A.sinkhorn_log_potentials(a.cuda(), a.cuda(), Cn, reg=0.05, max_iter=50, threshold=0.0)
    assert int(iters) == 50
    want = O.sinkhorn_log(a, a, Cn.cpu(), reg=0.05, max_iter=50, threshold=0.0)
    rep.check("pi vs oracle", pi, want)
    # after the u-update the row marginals are exact up to the +1e-8 inside the log
    rep.check("row marginals == a", pi.sum(-1), a, tol=2e-4)
    rep.check("OT cost vs oracle", A.ot_cost(C, pi), (C.cpu() * want).sum())
    rep.finish()


def test_empirical_cov_streaming_matches_full(A):
    """Reference tests/test_empirical_cov.py: streaming (n, sum, sum xxT) equals the full-batch mean/cov and their
    W2 distance is ~0 (D=128, 1e4 samples, batches of 100)."""
    D, N = 128, 10000
    g = torch.Generator().manual_seed(1)
    m = torch.randn(D, D, generator=g, dtype=torch.float64)
    zs = (torch.randn(N, D, generator=g, dtype=torch.float64) @ m.T + torch.randn(D, generator=g, dtype=torch.float64)).cuda()
    model = A.GaussianModel(D, dtype=torch.double).cuda()
    model.reset()
    for b in range(N // 100):
        model.update(zs[b * 100:(b + 1) * 100])
    mean, cov = A.mean_cov(model._running_sum, model._running_sum_cov, model._n_obs)
    mean_all = zs.mean(0)
    cov_all = (zs - mean_all).T @ (zs - mean_all) / N
    assert (mean - mean_all).norm() / mean_all.norm() < 1e-8
    assert (cov - cov_all).norm() / cov_all.norm() < 1e-8
    assert A.w2_gaussian(mean_all, mean, cov_all, cov, make_pd=True).abs().item() < 1e-4


@pytest.mark.parametrize("D", [129, 200, 512, 1024])
def test_block_jacobi_eigh_large_D(A, D):
    """eigh-based matrix functions for D > 128 (tests/test_latent_transport.py:70 of the
    reference transports 64x4x4 = 1024-dimensional latents).  Checker: torch.linalg.eigh on the CPU in fp64."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    rep = Report(f"block-Jacobi eigh D={D} vs torch.linalg.eigh (CPU, fp64)")
    g = torch.Generator().manual_seed(100 + D)
    x = torch.randn(3 * D, D, generator=g, dtype=torch.float64) * torch.linspace(0.2, 3.0, D, dtype=torch.float64)
    cov = (x.T @ x) / x.shape[0]
    lam, vec = torch.linalg.eigh(cov)
    ev, sq = MU.eigvals_and_fn(cov.cuda(), 1)
    rep.check("eigenvalues (sorted)", torch.sort(ev.cpu())[0], lam, tol=1e-10)
    want = (vec * lam.sqrt()) @ vec.T
    rep.check("sqrtm", sq, want, tol=1e-9)
    rep.check("sqrtm @ sqrtm == cov", (sq @ sq), cov, tol=1e-9)
    isq = MU.invsqrtm(cov.cuda())
    rep.check("invsqrtm", isq, (vec / lam.sqrt()) @ vec.T, tol=1e-8)
    # lower triangle is what counts (torch.linalg.eigh UPLO='L'): garbage in the strict upper triangle is ignored
    junk = cov.clone()
    junk[torch.triu(torch.ones(D, D, dtype=torch.bool), 1)] = 7.0
    rep.check("UPLO='L'", MU.eigvals_and_fn(junk.cuda(), 1)[1], want, tol=1e-9)
    rep.finish()


@pytest.mark.parametrize("D,nb", [(136, 3), (200, 2), (320, 3), (520, 1)])
def test_block_jacobi_eigh_batches_and_hard_spectra(A, D, nb):
    """The multi-workgroup solver (128 < D <= 1024) on batches (the eigenvector update then takes one row slab per block pair),
    indefinite input (no Cholesky factor: the columns of A itself are iterated) and rank-deficient input (columns shrinking to
    zero: the tracked squared norms are recounted): eigenvalues, V diag(lambda) V^T and V V^T = I against torch.linalg.eigh."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    g = torch.Generator().manual_seed(300 + D)
    x = torch.randn(nb, 2 * D, D, generator=g, dtype=torch.float64) * torch.linspace(0.1, 4.0, D, dtype=torch.float64)
    cov = x.transpose(-1, -2) @ x / x.shape[-2]
    sym = torch.randn(nb, D, D, generator=g, dtype=torch.float64)
    sym = sym + sym.transpose(-1, -2)
    low = x[:, : D // 3].transpose(-1, -2) @ x[:, : D // 3] / D
    eye = torch.eye(D, dtype=torch.float64)
    for name, m, tol in (("cov", cov, 1e-10), ("indefinite", sym, 1e-10), ("rank-deficient", low, 1e-10)):
        lam = torch.linalg.eigvalsh(m)
        ev, vt = MU.eigh_vectors(m.cuda())
        ev, vt = ev.cpu(), vt.cpu()
        scale = lam.abs().max()
        assert float((torch.sort(ev, dim=-1)[0] - lam).abs().max() / scale) < tol, (name, "eigenvalues")
        recon = vt.transpose(-1, -2) @ (ev.unsqueeze(-1) * vt)
        assert float((recon - m).abs().max() / scale) < tol, (name, "reconstruction")
        assert float((vt @ vt.transpose(-1, -2) - eye).abs().max()) < tol, (name, "orthonormality")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_library_matmul_and_row_softmax_vs_float64(A, dtype):
    """``matrix_utils.mm`` / ``softmax_rows``: the weighted sums (weights @ atoms, probs^T @ samples) and assignment distributions
    (softmax(energy / T), F.gumbel_softmax) of the mixture / codebook models on the library's own kernels, values and gradients
    against float64 torch arithmetic; batched, a 2-D operand shared by the batch, broadcast leading dimensions."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    tol = 2e-6 if dtype == torch.float32 else 1e-13
    rep = Report(f"mm / softmax_rows ({dtype}) vs float64")
    g = torch.Generator().manual_seed(12)
    cases = [((3, 37, 50), (3, 50, 19)), ((37, 50), (50, 19)), ((2, 4, 9, 33), (33, 5)), ((5, 1, 64), (5, 64, 300)),
             ((2, 1, 7, 16), (1, 3, 16, 8)), ((300, 1024), (1024, 16))]
    for sa, sb in cases:
        a = torch.randn(*sa, generator=g, dtype=torch.float64).to(dtype).cuda().requires_grad_(True)
        b = torch.randn(*sb, generator=g, dtype=torch.float64).to(dtype).cuda().requires_grad_(True)
        out = MU.mm(a, b)
        ar, br = a.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
        ref = ar @ br
        assert out.shape == ref.shape and out.dtype == dtype
        go = torch.randn(ref.shape, generator=g, dtype=torch.float64).cuda()
        ga, gb = torch.autograd.grad(out, (a, b), go.to(dtype))
        ra, rb = torch.autograd.grad(ref, (ar, br), go)
        rep.check(f"mm {sa} x {sb}", out, ref.detach(), tol)
        rep.check(f"mm {sa} x {sb}: da", ga, ra, tol)
        rep.check(f"mm {sa} x {sb}: db", gb, rb, tol)
    for shape, scale in (((4, 33, 10), 1.0), ((700, 1024), 1 / 0.3), ((5, 1), 2.0), ((2, 3, 4, 130), 0.05)):
        x = (torch.randn(*shape, generator=g, dtype=torch.float64) * 3).to(dtype).cuda().requires_grad_(True)
        y = MU.softmax_rows(x, scale)
        xr = x.detach().double().requires_grad_(True)
        ref = torch.softmax(xr * scale, dim=-1)
        go = torch.randn(shape, generator=g, dtype=torch.float64).cuda()
        (gx,) = torch.autograd.grad(y, x, go.to(dtype))
        (rx,) = torch.autograd.grad(ref, xr, go)
        rep.check(f"softmax {shape}", y, ref.detach(), tol)
        rep.check(f"softmax {shape}: dx", gx, rx, 5 * tol)
    with pytest.raises(RuntimeError):
        MU.mm(torch.zeros(2, 2), torch.zeros(2, 2))
    rep.finish()


@pytest.mark.parametrize("D", [16, 65, 128])
def test_warm_started_eigensolver_gives_the_same_decomposition(D):
    """``otvae_eigh_fn_warm``: the one-sided solver started from an orthonormal basis (the eigenvectors of a nearby matrix, or any
    orthonormal matrix) instead of the identity -- same eigenvalues, a valid eigenbasis of the SAME matrix; with the device flag at
    0 the call is the cold solver bit for bit."""
    import ctypes as C
    from ot_vae_lightning_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(40 + D)
    x = torch.randn(3 * D, D, generator=g, dtype=torch.float64) * torch.linspace(0.3, 2.0, D, dtype=torch.float64)
    a0 = (x.T @ x / x.shape[0])
    pert = torch.randn(D, D, generator=g, dtype=torch.float64) * 0.02
    a1 = a0 + (pert + pert.T) * 0.5 + 0.1 * torch.eye(D, dtype=torch.float64)

    def solve(a, basis=None, flag=1):
        ad = a.reshape(1, D, D).contiguous().cuda()
        ev, vt = torch.empty((1, D), dtype=torch.float64, device="cuda"), torch.empty((1, D, D), dtype=torch.float64, device="cuda")
        ws = torch.empty(lib.otvae_eigh_ws(1, D), dtype=torch.uint8, device="cuda")
        if basis is None:
            L.check(lib.otvae_eigh_fn(L.ptr(ad), 1, D, 3, L.ptr(vt), L.ptr(ev), L.ptr(ws), L.stream()), "otvae_eigh_fn")
        else:
            g0 = torch.empty((1, D, D), dtype=torch.float64, device="cuda")
            warm = torch.full((1,), flag, dtype=torch.int32, device="cuda")
            L.check(lib.otvae_eigh_fn_warm(L.ptr(ad), L.ptr(basis.contiguous()), L.ptr(warm), 1, D, 3, L.ptr(vt), L.ptr(ev), L.ptr(ws),
                                           L.ptr(g0), L.stream()), "otvae_eigh_fn_warm")
        return ev[0].cpu(), vt[0].cpu()

    _, v0 = solve(a0)
    lam = torch.linalg.eigvalsh(a1)
    eye = torch.eye(D, dtype=torch.float64)
    for name, basis in (("previous eigenvectors", v0.unsqueeze(0).cuda()),
                        ("random orthonormal", torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))[0].unsqueeze(0).cuda())):
        ev, vt = solve(a1, basis)
        assert float((torch.sort(ev)[0] - lam).abs().max() / lam.max()) < 1e-12, name
        assert float((vt.T @ (ev.unsqueeze(-1) * vt) - a1).abs().max() / lam.max()) < 1e-12, name
        assert float((vt @ vt.T - eye).abs().max()) < 1e-12, name
    cold = solve(a1)
    flagged_off = solve(a1, v0.unsqueeze(0).cuda(), flag=0)
    assert torch.equal(cold[0], flagged_off[0]) and torch.equal(cold[1], flagged_off[1])


def test_stochastic_transport_operator_vs_reference_golden(A):
    """eq. 19 of Freirich et al. (reference ot/w2_utils.py:391-458,732-786): the stochastic operator (T, Cw) for degenerate /
    nearly degenerate sources, diagonal and full (pseudo-inverse and the three functions of the target covariance from
    eigendecompositions), and ``apply_transport`` with noise: the reference's draw is recovered from its noisy output
    (eps = chol(Cw)^-1 noise, resp. noise / Cw) and injected, so the device Cholesky factor is what is being compared."""
    from ot_vae_lightning_amd.ot import w2_utils as W
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    G = load_golden("stochastic.npz")
    rep = Report("stochastic transport operator (eq. 19) vs reference golden")
    d_, f_ = group(G, "diag"), group(G, "full")
    T, Cw = W.compute_transport_operators(d_["cs"].cuda(), d_["ct"].cuda(), stochastic=True, diag=True, pg_star=0.2, make_pd=True)
    rep.check("diag: T", T, d_["T"], 1e-12)
    rep.check("diag: Cw", Cw, d_["Cw"], 1e-9)
    ms, mt = d_["ms"].cuda(), d_["mt"].cuda()
    eps = (d_["moved_noisy"] - d_["moved"]) / d_["Cw_used"].unsqueeze(-2)
    moved = W.apply_transport(d_["x"].cuda(), ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-2), d_["Cw_used"].cuda().unsqueeze(-2),
                              diag=True, noise_eps=eps.cuda())
    rep.check("diag: noisy transport", moved, d_["moved_noisy"], 1e-12)
    T, Cw = W.compute_transport_operators(f_["cs"].cuda(), f_["ct"].cuda(), stochastic=True, diag=False, pg_star=0.1, make_pd=True)
    rep.check("full: T", T, f_["T"], 1e-7)
    # Cw = Ct^1/2 (I - Ct^1/2 T* Cs^+ T* Ct^1/2) Ct^1/2 cancels to O(1) from terms of size 1e6 at this nearly singular source: a 1e-15
    # relative perturbation of Cs moves the REFERENCE's own Cw by 1.4e-4 ... 3.6e-4 (measured with the oracle on the CPU)
    rep.check("full: Cw", Cw, f_["Cw"], 1e-3)
    L_ref = torch.linalg.cholesky(f_["Cw_used"])
    rep.check("cholesky kernel", MU.cholesky(f_["Cw_used"].cuda()), L_ref, 1e-13)
    noise = f_["moved_noisy"] - f_["moved"]                                        # [2, 9, 6]
    eps = torch.linalg.solve_triangular(L_ref.unsqueeze(-3), noise.unsqueeze(-1), upper=False).squeeze(-1)
    moved = W.apply_transport(f_["x"].cuda(), ms.unsqueeze(-2), mt.unsqueeze(-2), f_["T"].cuda().unsqueeze(-3),
                              f_["Cw_used"].cuda().unsqueeze(-3), diag=False, make_pd=True, noise_eps=eps.cuda())
    rep.check("full: noisy transport", moved, f_["moved_noisy"], 1e-11)
    free = W.apply_transport(f_["x"].cuda(), ms.unsqueeze(-2), mt.unsqueeze(-2), f_["T"].cuda().unsqueeze(-3),
                             f_["Cw_used"].cuda().unsqueeze(-3), diag=False, make_pd=True)       # drawn on the device
    assert free.shape == moved.shape and torch.isfinite(free).all() and not torch.equal(free, moved)
    # well-conditioned full-rank sources: Cw is analytically zero.  Diagonal: it comes out NEGATIVE at 1e-9 size and
    # the reference's allclose(Cw, 0) test (ot/w2_utils.py:507) makes the transport deterministic -- no "valid variance" error.
    # Full: the 1e-8 regularisation leaves Cw ~ 1.9e-8 I, above that test's tolerance: the reference draws noise (recorded)
    wd, wf = group(G, "wc_diag"), group(G, "wc_full")
    T, Cw = W.compute_transport_operators(wd["cs"].cuda(), wd["ct"].cuda(), stochastic=True, diag=True, pg_star=0.2, make_pd=True)
    rep.check("well-conditioned diag: T", T, wd["T"], 1e-12)
    rep.check("well-conditioned diag: Cw (1e-9 size, negative)", Cw, wd["Cw"], 1e-6, floor=1e-8)
    assert float(Cw.max()) < 0 and float(Cw.abs().max()) <= 1e-8
    for mk in (False, True):
        moved = W.apply_transport(wd["x"].cuda(), ms.unsqueeze(-2), mt.unsqueeze(-2), T.unsqueeze(-2), Cw.unsqueeze(-2), diag=True, make_pd=mk)
        rep.check(f"well-conditioned diag: transport is deterministic (make_pd={mk})", moved, wd["moved"], 1e-13)
    T, Cw = W.compute_transport_operators(wf["cs"].cuda(), wf["ct"].cuda(), stochastic=True, diag=False, pg_star=0.1, make_pd=True)
    rep.check("well-conditioned full: T", T, wf["T"], 1e-10)
    rep.check("well-conditioned full: Cw (~1.9e-8 I)", Cw, wf["Cw"], 2e-2, floor=1e-8)
    L_ref = torch.linalg.cholesky(wf["Cw"])
    eps = torch.linalg.solve_triangular(L_ref.unsqueeze(-3), (wf["moved"] - wf["moved_quiet"]).unsqueeze(-1), upper=False).squeeze(-1)
    moved = W.apply_transport(wf["x"].cuda(), ms.unsqueeze(-2), mt.unsqueeze(-2), wf["T"].cuda().unsqueeze(-3),
                              wf["Cw"].cuda().unsqueeze(-3), diag=False, make_pd=False, noise_eps=eps.cuda())
    rep.check("well-conditioned full: noisy transport (the reference's draw)", moved, wf["moved"], 1e-11)
    # through the operator class: a stochastic GaussianTransport computes and transports
    op = A.GaussianTransport(6, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double),
                             transport_cfg=dict(diag=False, stochastic=True, pg_star=0.1, make_pd=True, verbose=False, dtype=torch.double)).cuda()
    g = torch.Generator().manual_seed(5)
    op.update(source_samples=(torch.randn(400, 6, generator=g, dtype=torch.double) * torch.tensor([1, 1, 1, 1e-3, 1e-3, 1e-3])).cuda(),
              target_samples=torch.randn(400, 6, generator=g, dtype=torch.double).cuda() + 1.0)
    assert torch.isfinite(op.compute()).all() and op.cov_stochastic_noise is not None
    out = op.transport(torch.randn(16, 6, generator=g, dtype=torch.double).cuda())
    assert out.shape == (16, 6) and torch.isfinite(out).all()
    rep.finish()


@pytest.mark.parametrize("solver", ["one_sided", "two_sided"])
def test_small_eigh_solvers_vs_lapack(A, solver, monkeypatch):
    """The D <= 128 eigensolvers (one-sided Hestenes Jacobi with replayed rotations, the default; the first-generation two-sided
    Jacobi behind OTVAE_EIGH_TWOSIDED=1) against torch.linalg.eigh on the CPU: covariance-like, indefinite, rank-deficient,
    diagonal, identity and 1 x 1 matrices, odd sizes, batches, garbage in the strict upper triangle."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    if solver == "two_sided":
        monkeypatch.setenv("OTVAE_EIGH_TWOSIDED", "1")
    rep = Report(f"small eigh ({solver}) vs torch.linalg.eigh (CPU, fp64)")
    g = torch.Generator().manual_seed(77)
    for D in (1, 2, 3, 8, 17, 33, 64, 100, 127, 128):
        x = torch.randn(4, 3 * D + 2, D, generator=g, dtype=torch.float64) * torch.linspace(0.2, 3.0, D, dtype=torch.float64)
        cov = x.transpose(-1, -2) @ x / x.shape[-2]
        sym = torch.randn(4, D, D, generator=g, dtype=torch.float64)
        sym = sym + sym.transpose(-1, -2)                                       # indefinite
        low = x[:, : max(1, D // 2)].transpose(-1, -2) @ x[:, : max(1, D // 2)]      # rank-deficient (zero eigenvalues)
        for name, m in (("cov", cov), ("indefinite", sym), ("rank-deficient", low), ("identity", torch.eye(D, dtype=torch.float64).expand(2, D, D)),
                        ("diagonal", torch.diag_embed(torch.linspace(-1.0, 2.0, D, dtype=torch.float64)).expand(2, D, D))):
            lam, vec = torch.linalg.eigh(m)
            ev, vt = MU.eigh_vectors(m.cuda())
            ev, vt = ev.cpu(), vt.cpu()
            scale = lam.abs().max().clamp(min=1e-300)
            assert float((torch.sort(ev, dim=-1)[0] - lam).abs().max() / scale) < 1e-11, (D, name)
            recon = vt.transpose(-1, -2) @ (ev.unsqueeze(-1) * vt)               # V diag(lambda) V^T
            assert float((recon - m).abs().max() / scale) < 1e-11, (D, name, "reconstruction")
            eye = torch.eye(D, dtype=torch.float64)
            assert float((vt @ vt.transpose(-1, -2) - eye).abs().max()) < 1e-11, (D, name, "orthonormality")
        want = (vec_ := torch.linalg.eigh(cov))[1] @ torch.diag_embed(vec_[0].sqrt()) @ vec_[1].transpose(-1, -2)
        rep.check(f"D={D}: sqrtm", MU.sqrtm(cov.cuda()), want, tol=1e-11)
        rep.check(f"D={D}: invsqrtm", MU.invsqrtm(cov.cuda()), torch.linalg.inv(want), tol=1e-9)
        rep.check(f"D={D}: min_eig (indefinite)", MU.min_eig(sym.cuda()), torch.linalg.eigh(sym)[0][..., 0], tol=1e-11)
        junk = cov.clone()
        junk[:, torch.triu(torch.ones(D, D, dtype=torch.bool), 1)] = 7.0
        rep.check(f"D={D}: UPLO='L'", MU.sqrtm(junk.cuda()), want, tol=1e-11)
    rep.finish()


def test_eigh_graded_spectra_and_capture(A):
    """Round 3: positive definite input is factored first and the Jacobi iteration runs on the Cholesky factor's columns (in LDS
    for D <= 128, on the columns of L for the block solver), eigenvectors = the normalised final columns.  (1) graded spectra up
    to condition 1e12, which the iteration on A itself could not finish inside its sweep budget at D >= 96 (NaN by contract):
    residual and orthonormality at fp64 rounding, eigenvalues to the accuracy the matrix defines them (eps * cond).  (2) the block
    solver's host-following sweep loop (not capturable) against the blind 24-sweep budget recorded under a hipGraph capture: the
    launches skipped are no-ops, so the two agree bit for bit."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    g = torch.Generator().manual_seed(5)
    for D in (8, 31, 64, 96, 128, 200, 256):
        for cond in (1e1, 1e6, 1e12):
            q, _ = torch.linalg.qr(torch.randn(D, D, generator=g, dtype=torch.float64))
            lam = torch.logspace(0, -math.log10(cond), D, dtype=torch.float64)
            cov = (q * lam) @ q.T
            cov = 0.5 * (cov + cov.T)
            ev, vt = MU.eigh_vectors(cov.cuda()[None])
            ev, v = ev[0].cpu(), vt[0].cpu().T
            assert bool(torch.isfinite(ev).all()), (D, cond)
            assert float((cov @ v - v * ev).norm() / cov.norm()) < 1e-13, (D, cond, "residual")
            assert float((v.T @ v - torch.eye(D, dtype=torch.float64)).abs().max()) < 1e-12, (D, cond, "orthonormality")
            rel = ((ev.sort().values - lam.sort().values).abs() / lam.sort().values).max()
            assert float(rel) < 64 * D * 2.3e-16 * cond, (D, cond, float(rel))
    x = torch.randn(2, 700, 256, generator=g, dtype=torch.float64)
    cov = (x.transpose(-1, -2) @ x / 700).cuda()
    ev0, vt0 = MU.eigh_vectors(cov)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ev1, vt1 = MU.eigh_vectors(cov)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(ev0, ev1) and torch.equal(vt0, vt1)


def test_eigh_plus_minus_lambda_pairs(A):
    """indefinite matrices whose diagonal is NON-negative never triggered the up-front |A|_inf shift of the one-sided
    solver, and inside a +-lambda pair (a double eigenvalue of A^2) the iteration stops at any mixture of the two eigenvectors:
    [[0,1],[1,0]] came back as V = I, lambda = (+1, +1) and passed for positive definite.  The finish kernel now tests
    |v . g| = |g| per pair and repeats the matrix on the shifted one.  Cases: the 2x2 swap, bipartite adjacency matrices (spectrum
    symmetric about 0), random symmetric matrices with the diagonal replaced by its absolute value, batches mixing such a matrix
    with a covariance (only the failing matrix is repeated).  D > 128 (no second pass there): NaN eigenvalues, not a wrong answer."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    g = torch.Generator().manual_seed(123)
    cases = [("swap", torch.tensor([[0., 1.], [1., 0.]], dtype=torch.float64)[None])]
    for n1, n2 in ((3, 3), (5, 8), (32, 32), (64, 63)):
        Bm = (torch.rand(n1, n2, generator=g, dtype=torch.float64) < 0.4).double()
        adj = torch.zeros(n1 + n2, n1 + n2, dtype=torch.float64)
        adj[:n1, n1:] = Bm
        adj[n1:, :n1] = Bm.T
        cases.append((f"bipartite {n1}+{n2}", adj[None]))
    for D in (4, 17, 64, 128):
        sym = torch.randn(3, D, D, generator=g, dtype=torch.float64)
        sym = sym + sym.transpose(-1, -2)
        sym.diagonal(dim1=-1, dim2=-2).abs_()
        cases.append((f"|diag| symmetric D={D}", sym))
    x = torch.randn(40, 6, generator=g, dtype=torch.float64)
    mixed = torch.stack([x.T @ x / 40, torch.block_diag(torch.tensor([[0., 2.], [2., 0.]], dtype=torch.float64), torch.eye(4, dtype=torch.float64)),
                         torch.eye(6, dtype=torch.float64)])
    cases.append(("batch: covariance, swap block, identity", mixed))
    for name, m in cases:
        lam = torch.linalg.eigvalsh(m)
        ev, vt = MU.eigh_vectors(m.cuda())
        ev, vt = ev.cpu(), vt.cpu()
        scale = lam.abs().max()
        assert float((torch.sort(ev, dim=-1)[0] - lam).abs().max() / scale) < 1e-11, (name, ev, lam)
        recon = vt.transpose(-1, -2) @ (ev.unsqueeze(-1) * vt)
        assert float((recon - m).abs().max() / scale) < 1e-11, (name, "reconstruction")
        assert float((MU.min_eig(m.cuda()).cpu() - lam[..., 0]).abs().max() / scale) < 1e-11, (name, "min_eig")
        assert not bool(MU.is_pd(m.cuda())[lam[..., 0] < 0].any()), name
    # the block solver has no second pass: a +-lambda pair must be loud (NaN), never a confident wrong spectrum
    n1 = 80
    Bm = (torch.rand(n1, n1, generator=g, dtype=torch.float64) < 0.3).double()
    adj = torch.zeros(2 * n1, 2 * n1, dtype=torch.float64)
    adj[:n1, n1:], adj[n1:, :n1] = Bm, Bm.T
    ev = MU.eigh_vectors(adj.cuda()[None])[0].cpu()
    lam = torch.linalg.eigvalsh(adj)
    ok = torch.isfinite(ev).all() and float((torch.sort(ev[0])[0] - lam).abs().max() / lam.abs().max()) < 1e-10
    assert ok or torch.isnan(ev).all(), "D = 160 bipartite: neither the right spectrum nor NaN"


@pytest.mark.parametrize("diag", [False, True])
def test_gaussian_model_update_with_autograd(A, diag):
    """VERDICT r2 #7 / reference tests/test_distribution_models.py:150-168: ``GaussianModel(update_with_autograd=True)`` -- mean and
    the ExpScaleTril-parametrised scale as nn.Parameters trained through -log_prob.  (1) log-density and its gradients with
    respect to samples, mean and the raw scale parameter against torch.distributions under autograd on the CPU (what the
    reference evaluates); (2) the reference's own test loop (AdamW, cosine schedule, 10 epochs): W2 to the true Gaussian < 0.1."""
    import torch.distributions as D
    rep = Report(f"GaussianModel(update_with_autograd=True, diag={diag}) vs torch.distributions autograd (CPU, fp64)")
    g = torch.Generator().manual_seed(17)
    for lead, d, b in (((), 6, 33), ((2,), 17, 40), ((), 128, 300)):
        model = A.GaussianModel(*lead, d, update_with_autograd=True, dtype=torch.double, w2_cfg=dict(diag=diag, make_pd=True)).cuda()
        raw = model.parametrizations.cov.original
        with torch.no_grad():
            raw.copy_(torch.randn(raw.shape, generator=g, dtype=torch.double) * 0.3)
            model.mean.copy_(torch.randn(model.mean.shape, generator=g, dtype=torch.double))
        x = torch.randn(*lead, b, d, generator=g, dtype=torch.double)
        w = torch.randn(*lead, b, generator=g, dtype=torch.double)
        xg = x.cuda().requires_grad_(True)
        lp = model(xg)
        (lp * w.cuda()).sum().backward()
        # the reference's arithmetic: ExpScaleTril, then MultivariateNormal(scale_tril=) / Independent(Normal(scale = cov ** 0.5))
        mean_c = model.mean.detach().cpu().requires_grad_(True)
        raw_c = raw.detach().cpu().requires_grad_(True)
        xc = x.clone().requires_grad_(True)
        if diag:
            dist = D.Independent(D.Normal(mean_c.unsqueeze(-2), (raw_c.exp().unsqueeze(-2)) ** 0.5), 1)
        else:
            tril = raw_c.tril(-1) + torch.diag_embed(raw_c.diagonal(dim1=-1, dim2=-2).exp())
            dist = D.MultivariateNormal(mean_c.unsqueeze(-2), scale_tril=tril.unsqueeze(-3))
        want = dist.log_prob(xc)
        (want * w).sum().backward()
        tag = f"lead={lead} D={d} B={b}"
        rep.check(f"{tag}: log_prob", lp, want, 1e-11)
        rep.check(f"{tag}: d/d samples", xg.grad, xc.grad, 1e-10)
        rep.check(f"{tag}: d/d mean", model.mean.grad, mean_c.grad, 1e-10)
        rep.check(f"{tag}: d/d raw scale parameter", raw.grad, raw_c.grad, 1e-10)
    # the reference's training loop on one known Gaussian
    d, n, bs, epochs = 8, 2000, 200, 10
    a = torch.randn(d, d, generator=g, dtype=torch.double) / d ** 0.5
    true_cov = a @ a.T + 0.5 * torch.eye(d, dtype=torch.double)
    if diag:
        true_cov = torch.diag(true_cov.diagonal())
    true_mean = torch.randn(d, generator=g, dtype=torch.double)
    samples = (torch.randn(n, d, generator=g, dtype=torch.double) @ torch.linalg.cholesky(true_cov).T + true_mean).cuda()
    model = A.GaussianModel(d, update_with_autograd=True, dtype=torch.double, w2_cfg=dict(diag=diag, make_pd=True)).cuda()
    optim = torch.optim.AdamW(model.parameters(), lr=0.1, betas=(0., 0.99), weight_decay=1e-2)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=epochs * n // bs, eta_min=1e-5)
    for _ in range(epochs):
        for i in range(0, n, bs):
            optim.zero_grad()
            nll = -model(samples[i:i + bs]).mean()
            nll.backward()
            optim.step()
            sched.step()
    w2 = A.w2_gaussian(model.mean.detach(), true_mean.cuda(), torch.diag_embed(model.variances.detach()) if diag else model.variances.detach(),
                       true_cov.cuda(), make_pd=True)
    rep.rows.append(("W2^2 to the true Gaussian after the reference's training loop (< 0.1)", float(w2), 0.1, float(w2) < 0.1))
    assert float(w2) < 0.1, float(w2)
    rep.finish()


@pytest.mark.parametrize("tag", ["fit_diag", "fit_full"])
def test_gmm_forward_is_the_mixture_log_density(A, tag):
    """`GaussianMixtureModel(...)(samples)` in eval mode: the reference class resolves `predict` to GaussianModel.predict (its first
    base), the log-density of the MixtureSameFamily -- golden from the reference's own fitted model (gmm_autograd.npz)."""
    g = group(load_golden("gmm_autograd.npz"), tag)
    K, d, B, diag = (int(v) for v in g["cfg"])
    lead = tuple(g["mean"].shape[:-2])
    model = A.GaussianMixtureModel(*lead, d, mixture_cfg=dict(n_components=K), w2_cfg=dict(diag=bool(diag), make_pd=True),
                                   dtype=torch.double).cuda()
    with torch.no_grad():  # the values stored BEHIND the parametrisations (reading `cov` adds the strict positive-definite shift)
        model.mean.copy_(g["mean"].cuda())
        model.parametrizations.cov.original.copy_(g["raw_cov"].cuda())
        model.parametrizations._weights.original.copy_(g["raw_weights"].cuda())
    model.eval()
    rep = Report(f"GaussianMixtureModel forward ({tag}) vs the reference class")
    rep.check("cov as read", model.cov, g["cov"], 1e-12)
    rep.check("weights as read", model.weights, g["weights"], 1e-12)
    rep.check("log-density", model(g["probe"].cuda()), g["log_prob"], 1e-10)
    rep.finish()


@pytest.mark.parametrize("tag", ["auto_diag", "auto_diag_lead", "auto_full", "auto_full_lead"])
def test_gmm_update_with_autograd(A, tag):
    """`GaussianMixtureModel(update_with_autograd=True)` (gassian_mixture_model.py:53-58): log-density and its gradients with
    respect to samples, means, the raw ExpScaleTril parameter and the raw soft-max weights against the reference class under
    torch.autograd (golden gmm_autograd.npz), plus the derived `variances` / `weights` attributes."""
    g = group(load_golden("gmm_autograd.npz"), tag)
    K, d, B, diag = (int(v) for v in g["cfg"])
    lead = tuple(g["mean"].shape[:-2])
    model = A.GaussianMixtureModel(*lead, d, mixture_cfg=dict(n_components=K), w2_cfg=dict(diag=bool(diag), make_pd=True),
                                   update_with_autograd=True, dtype=torch.double).cuda()
    raw_cov, raw_w = model.parametrizations.cov.original, model.parametrizations._weights.original
    assert model.mean.requires_grad and raw_cov.requires_grad and raw_w.requires_grad
    with torch.no_grad():
        model.mean.copy_(g["mean"].cuda())
        raw_cov.copy_(g["raw_cov"].cuda())
        raw_w.copy_(g["raw_weights"].cuda())
    x = g["x"].cuda().requires_grad_(True)
    model.train()
    lp = model(x)
    (lp * g["seed"].cuda()).sum().backward()
    rep = Report(f"GaussianMixtureModel(update_with_autograd=True) ({tag}) vs the reference class")
    rep.check("log-density", lp, g["log_prob"], 1e-10)
    rep.check("d/d samples", x.grad, g["g_x"], 1e-9)
    rep.check("d/d mean", model.mean.grad, g["g_mean"], 1e-9)
    rep.check("d/d raw cov parameter", raw_cov.grad, g["g_raw_cov"], 1e-9)
    rep.check("d/d raw weight parameter", raw_w.grad, g["g_raw_weights"], 1e-9)
    rep.check("variances", model.variances, g["variances"], 1e-12)
    rep.check("weights", model.weights, g["weights"], 1e-12)
    rep.finish()
    with pytest.raises(RuntimeError):
        model.update(x.detach())


@pytest.mark.parametrize("tag", ["mean", "mean_lead", "argmax"])
def test_codebook_update_with_autograd(A, tag):
    """`CodebookModel(update_with_autograd=True)` (codebook_model.py:89-93): predictions, assignment probabilities, entropy and the
    gradients of a scalar made of all three with respect to the samples AND the codebook (otvae_codebook_probs_bwd_atoms +
    the library GEMM's backward) against the reference class under torch.autograd (golden codebook_autograd.npz)."""
    g = group(load_golden("codebook_autograd.npz"), tag)
    K, d, B = (int(v) for v in g["cfg"][:3])
    T = float(g["cfg"][3])
    mode = "argmax" if tag == "argmax" else "mean"
    lead = tuple(g["codebook"].shape[:-2])
    model = A.CodebookModel(*lead, d, mixture_cfg=dict(n_components=K, temperature=T, training_mode=mode, inference_mode=mode),
                            update_with_autograd=True).cuda()
    assert model.codebook.requires_grad and not hasattr(model, "_n_obs")
    with torch.no_grad():
        model.codebook.copy_(g["codebook"].cuda())
    x = g["x"].cuda().requires_grad_(True)
    model.train()
    preds, _, dist = model(x)
    ent = dist.entropy()
    ((preds * g["s_pred"].cuda()).sum() + (dist.probs * g["s_prob"].cuda()).sum() + (ent * g["s_ent"].cuda()).sum()).backward()
    rep = Report(f"CodebookModel(update_with_autograd=True) ({tag}) vs the reference class")
    rep.check("predictions", preds, g["preds"], 1e-5)
    rep.check("assignment probabilities", dist.probs, g["probs"], 5e-5)
    rep.check("entropy", ent, g["entropy"], 5e-5, floor=1e-3)
    rep.check("d/d samples", x.grad, g["g_x"], 5e-4)
    rep.check("d/d codebook", model.codebook.grad, g["g_codebook"], 5e-4)
    rep.finish()
    with pytest.raises(RuntimeError):
        model.update(x.detach())


CODEBOOK_OPTION_CASES = [("cos_p2_mean", "cosine", 2.0, None, "mean", 0.5), ("cos_p1_argmax", "cosine", 1.0, None, "argmax", 1.0),
                         ("cos_p05_mean_top3", "cosine", 0.5, 3, "mean", 0.7), ("euc_p1_mean", "euclidean", 1.0, None, "mean", 0.6),
                         ("euc_p05_argmax", "euclidean", 0.5, None, "argmax", 1.0), ("euc_p3_mean_top2", "euclidean", 3.0, 2, "mean", 0.8),
                         ("euc_p2_mean_top3", "euclidean", 2.0, 3, "mean", 0.5), ("euc_p2_argmax_top1", "euclidean", 2.0, 1, "argmax", 1.0),
                         ("cos_p2_sample_top1", "cosine", 2.0, 1, "sample", 1.0)]


@pytest.mark.parametrize("case", CODEBOOK_OPTION_CASES, ids=[c[0] for c in CODEBOOK_OPTION_CASES])
def test_codebook_metric_p_topk(A, case):
    """`CodebookModel` with `metric='cosine'`, p != 2 and `topk` (reference base.py:166-235, codebook_model.py:150-168) on
    `otvae_codebook_energy/_bwd`: energies, predictions, assignment probabilities and the gradients of a seeded scalar with respect to
    samples and (trained) codebook against the reference's own class (codebook_options.npz)."""
    tag, metric, p, topk, mode, T = case
    g = group(load_golden("codebook_options.npz"), tag)
    K, d = g["codebook"].shape[-2:]
    model = A.CodebookModel(2, d, mixture_cfg=dict(n_components=K, metric=metric, p=p, topk=topk, temperature=T, training_mode=mode,
                                                   inference_mode=mode), update_with_autograd=True).cuda()
    with torch.no_grad():
        model.codebook.copy_(g["codebook"].cuda())
    x = g["x"].cuda().requires_grad_(True)
    model.train()
    energy = model.energy(x)
    preds, _, dist = model(x)
    ((preds * g["s_pred"].cuda()).sum() + (dist.probs * g["s_prob"].cuda()).sum()).backward()
    rep = Report(f"CodebookModel(metric={metric}, p={p}, topk={topk}, mode={mode}) vs the reference class")
    rep.check("energy", energy, g["energy"], 2e-5)
    rep.check("predictions", preds, g["preds"], 5e-5)
    rep.check("assignment probabilities", dist.probs, g["probs"], 1e-4)
    rep.check("d/d samples", x.grad, g["g_x"], 1e-3)
    rep.check("d/d codebook", model.codebook.grad, g["g_codebook"], 1e-3)
    rep.finish()


@pytest.mark.parametrize("tag,topk,mode", [("gmm_top2_mean", 2, "mean"), ("gmm_top1_sample", 1, "sample"), ("gmm_top3_argmax", 3, "argmax")])
def test_gmm_topk_assignment(A, tag, topk, mode):
    """`GaussianMixtureModel(topk=k)`: MixtureMixin.assign restricted to the k most likely components (base.py:217-220,228)."""
    g = group(load_golden("codebook_options.npz"), tag)
    model = A.GaussianMixtureModel(3, mixture_cfg=dict(n_components=5, topk=topk, temperature=0.9, training_mode=mode, inference_mode=mode),
                                   w2_cfg=dict(diag=True, make_pd=True, dtype=torch.double), dtype=torch.double).cuda()
    with torch.no_grad():
        model.mean.copy_(g["mean"].cuda())
        model.parametrizations.cov.original.copy_(g["cov"].cuda())
    model.eval()
    w, _, dist = model.assign(g["x"].cuda())
    rep = Report(f"GaussianMixtureModel(topk={topk}, mode={mode}).assign vs the reference class")
    rep.check("assignment probabilities", dist.probs, g["probs"], 1e-9)
    rep.check("weights", w, g["weights"], 1e-9)
    rep.finish()


@pytest.mark.parametrize("name", ["g_reparam2", "g_reparam_last_4d", "g_reparam2_empirical", "c_empirical", "c_fixed_var", "c_fixed_empirical",
                                  "c_reparam2", "c_reparam2_ema"])
def test_prior_corners_vs_reference_golden(A, name):
    """GaussianPrior with `reparam_dim` != 1 and ConditionalGaussianPrior with the options it inherits (empirical_kl, fixed_var,
    reparam_dim; prior/gaussian.py:58-96, prior/conditional_gaussian.py:44-93) against the reference's own classes (prior_corners.npz):
    z, loss, `out_size`, the gradients with respect to the input and the class embeddings; EMA variant: the buffers after the step."""
    from test_oracle_vs_golden import PRIOR_CORNERS
    g = group(load_golden("prior_corners.npz"), name)
    cls, kw = PRIOR_CORNERS[name]
    prior = getattr(A, cls)(**kw).cuda().train()
    cond = cls == "ConditionalGaussianPrior"
    if cond:
        with torch.no_grad():
            prior._mu.weight.copy_(g["init/_mu.weight"].cuda())
            prior._log_std.weight.copy_(g["init/_log_std.weight"].cuda())
    x = g["x"].cuda().requires_grad_(True)
    extra = {"labels": g["labels"].cuda()} if cond else {}
    z, loss, _ = prior(x, step=0, eps=g["eps"].cuda(), **extra)
    ((z * g["gz"].cuda()).sum() + (loss * g["gl"].cuda()).sum()).backward()
    rep = Report(f"{cls}({kw}) vs the reference class")
    assert list(prior.out_size(torch.Size(tuple(x.shape[1:])))) == [int(v) for v in g["out_size"]] and tuple(z.shape) == tuple(g["z"].shape)
    rep.check("z", z, g["z"], 1e-6)
    rep.check("loss", loss, g["loss"], 1e-5)
    rep.check("d/d input", x.grad, g["gx"], 1e-5)
    if cond:
        for k, p in prior.named_parameters():
            if f"grad/{k}" in g:
                rep.check(f"d/d {k}", p.grad, g[f"grad/{k}"], 1e-5)
        for k, v in prior.state_dict().items():
            rep.check(f"state {k}", v, g[f"state/{k}"], 1e-5)
    rep.finish()


def test_codebook_prior_with_trained_codebook(A):
    """CodebookPrior(loss='kl', soft mode) over a CodebookModel(update_with_autograd=True): encodings, loss and the gradients of
    (seeded encodings + loss) with respect to the latent and the codebook against the reference classes."""
    g = group(load_golden("codebook_autograd.npz"), "prior")
    K, T = int(g["cfg"][0]), float(g["cfg"][1])
    prior = A.CodebookPrior((8, 2, 2), embed_dims=(1,), loss="kl", loss_coeff=1.0, update_with_autograd=True,
                            mixture_cfg=dict(n_components=K, temperature=T, training_mode="mean", inference_mode="mean")).cuda()
    with torch.no_grad():
        prior.codebook_model.codebook.copy_(g["codebook"].cuda())
    z = g["z"].cuda().requires_grad_(True)
    prior.train()
    enc, loss, _ = prior.encode(z)
    ((enc * g["s_z"].cuda()).sum() + loss.sum()).backward()
    rep = Report("CodebookPrior over a trained codebook vs the reference classes")
    rep.check("encodings", enc, g["enc"], 1e-5)
    rep.check("loss", loss, g["loss"], 5e-5)
    rep.check("d/d latent", z.grad, g["g_z"], 5e-4)
    rep.check("d/d codebook", prior.codebook_model.codebook.grad, g["g_codebook"], 5e-4)
    rep.finish()


def test_matrix_utils_function_by_function_vs_reference_golden(A):
    """ot/matrix_utils.py:59-158 function by function on a zoo of matrices (definite, semi-definite, indefinite, slightly asymmetric,
    batched) against the reference's own functions (matrix_utils.npz): is_symmetric, min_eig, is_pd / is_spd, make_psd in all flag
    combinations (also for diagonal 'matrices' given as vectors), sqrtm / invsqrtm, mean_cov; and for invalid arguments of w2_gaussian /
    compute_transport_operators (ot/w2_utils.py:605-708) the same exception type -- or the same value where the reference accepts."""
    from ot_vae_lightning_amd.ot import matrix_utils as MU
    z = load_golden("matrix_utils.npz")
    g = {k: torch.from_numpy(z[k]) for k in z.files if not k.endswith("/error")}
    rep = Report("matrix_utils function by function vs the reference")
    for name in ("pd", "low", "indef", "asym", "nearly", "neg", "batch"):
        m = g[f"zoo/{name}"].cuda()
        assert torch.equal(MU.is_symmetric(m).cpu().to(torch.int64).reshape(-1), g[f"{name}/is_symmetric"].reshape(-1)), name
        if name == "asym":
            continue
        scale = float(m.abs().max())   # (a semi-definite matrix's smallest eigenvalue is rounding noise of either sign: absolute comparison)
        rep.check(f"{name}: min_eig", MU.min_eig(m), g[f"{name}/min_eig"], 1e-10, floor=scale)
        for strict in (0, 1):
            # (semi-definite `low`: its zero eigenvalues come out as +-1e-16 on either side; the verdict there is rounding in the reference too)
            if name not in ("low", "batch"):
                assert torch.equal(MU.is_pd(m, strict=bool(strict)).cpu().to(torch.int64).reshape(-1), g[f"{name}/is_pd/strict{strict}"].reshape(-1)), (name, strict)
                assert torch.equal(MU.is_spd(m, strict=bool(strict)).cpu().to(torch.int64).reshape(-1), g[f"{name}/is_spd/strict{strict}"].reshape(-1)), (name, strict)
            fixed, corr = MU.make_psd(m.clone(), strict=bool(strict), return_correction=True)
            rep.check(f"{name}: make_psd(strict={strict})", fixed, g[f"{name}/make_psd/strict{strict}/out"], 1e-10)
            rep.check(f"{name}: make_psd(strict={strict}) correction", corr, g[f"{name}/make_psd/strict{strict}/corr"], 1e-9, floor=scale)
    rep.check("sqrtm(pd)", MU.sqrtm(g["zoo/pd"].cuda()), g["pd/sqrtm"], 1e-10)
    rep.check("invsqrtm(pd)", MU.invsqrtm(g["zoo/pd"].cuda()), g["pd/invsqrtm"], 1e-9)
    # (sqrtm of the rank-deficient matrix is NaN or not by the sign of a 1e-16 eigenvalue, in the reference as here: not compared)
    for strict in (0, 1):
        fixed, corr = MU.make_psd(g["zoo/diag_vec"].cuda(), strict=bool(strict), return_correction=True, diag=True)
        rep.check(f"make_psd(diag, strict={strict})", fixed, g[f"make_psd_diag/strict{strict}/out"], 1e-12)
        rep.check(f"make_psd(diag, strict={strict}) correction", corr, g[f"make_psd_diag/strict{strict}/corr"], 1e-12, floor=1e-12)
    x = g["mean_cov/x"].cuda()
    n = torch.tensor([30.0, 30.0], dtype=torch.double, device="cuda")
    mean, cov = MU.mean_cov(x.sum(-2), x.transpose(-1, -2) @ x, n)
    rep.check("mean_cov: mean", mean, g["mean_cov/mean"], 1e-12)
    rep.check("mean_cov: cov", cov, g["mean_cov/cov"], 1e-11)
    mean_d, var_d = MU.mean_cov(x.sum(-2), (x ** 2).sum(-2), n, diag=True)
    rep.check("mean_cov(diag): mean", mean_d, g["mean_cov/mean_diag"], 1e-12)
    rep.check("mean_cov(diag): var", var_d, g["mean_cov/var_diag"], 1e-11)
    # argument errors
    pd, indef, asym = g["zoo/pd"].cuda(), g["zoo/indef"].cuda(), g["zoo/asym"].cuda()
    m5 = torch.zeros(5, dtype=torch.double, device="cuda")
    calls = {
        "w2_indef_source": lambda: A.w2_gaussian(m5, m5, indef, pd),
        "w2_indef_target": lambda: A.w2_gaussian(m5, m5, pd, indef),
        "w2_asym": lambda: A.w2_gaussian(m5, m5, asym, pd),
        "w2_shape": lambda: A.w2_gaussian(m5, torch.zeros(4, dtype=torch.double, device="cuda"), pd, pd),
        "w2_indef_make_pd": lambda: A.w2_gaussian(m5, m5, indef, pd, make_pd=True),
        "ops_indef": lambda: A.compute_transport_operators(indef, pd, stochastic=False, diag=False, pg_star=0.0),
        "ops_pg_star_range": lambda: A.compute_transport_operators(pd, pd, stochastic=False, diag=False, pg_star=1.5),
        "ops_diag_negative": lambda: A.compute_transport_operators(torch.tensor([1.0, -1.0], dtype=torch.double, device="cuda"),
                                                                   torch.ones(2, dtype=torch.double, device="cuda"), stochastic=False, diag=True, pg_star=0.0),
    }
    for name, fn in calls.items():
        if f"err/{name}/error" in z.files:
            want = bytes(z[f"err/{name}/error"].astype("uint8")).decode()
            with pytest.raises(Exception) as info:
                fn()
            assert type(info.value).__name__ == want, (name, type(info.value).__name__, want)
        else:
            res = fn()
            rep.check(f"{name}: value", res[0] if isinstance(res, tuple) else res, g[f"err/{name}/value"], 1e-8)
    rep.finish()


GT_CASES = {
    "lead3_full": ((3, 6), dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)),
    "lead3_diag": ((3, 6), dict(diag=True, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)),
    "lead2x2_full_pg": ((2, 2, 5), dict(diag=False, stochastic=False, pg_star=0.3, make_pd=True, verbose=False, dtype=torch.double)),
    "nolead_diag_pg": ((7,), dict(diag=True, stochastic=False, pg_star=0.6, make_pd=True, verbose=False, dtype=torch.double)),
    "lead3_full_ema": ((3, 6), dict(diag=False, stochastic=False, pg_star=0.0, make_pd=True, verbose=False, dtype=torch.double)),
}


@pytest.mark.parametrize("name", sorted(GT_CASES))
def test_gaussian_transport_leading_shapes_vs_reference_golden(A, name):
    """`GaussianTransport` with leading (per-position) shapes, diagonal models, pg_star > 0 and EMA statistics -- what
    `LatentTransport(common_operator=False)` builds (ot/transport/gaussian_transport.py:41-95) -- against the reference's own class
    (gaussian_transport_shapes.npz): distance, operator, fitted moments, transport of [*, B, D] and of [*, D] inputs."""
    g = group(load_golden("gaussian_transport_shapes.npz"), name)
    size, tcfg = GT_CASES[name]
    cfg = dict(update_decay=0.8 if name.endswith("_ema") else None, dtype=torch.double)
    op = A.GaussianTransport(*size, source_cfg=cfg, target_cfg=cfg, transport_cfg=tcfg).cuda()
    for a, b in zip(g["src"].cuda(), g["tgt"].cuda()):
        op.update(source_samples=a, target_samples=b)
    dist = op.compute()
    rep = Report(f"GaussianTransport{size} {name} vs the reference class")
    rep.check("W2^2", dist, g["w2"], 1e-8)
    rep.check("operator", op.transport_operator, g["T"], 1e-8)
    rep.check("source mean", op.source_model.mean, g["src_mean"], 1e-11)
    rep.check("source cov", op.source_model.cov, g["src_cov"], 1e-10)
    probe = g["src"][0][..., :5, :].cuda()
    rep.check("transport [*, B, D]", op.transport(probe), g["moved_batch"], 1e-8)
    rep.check("transport [*, D]", op.transport(probe[..., 0, :]), g["moved_single"], 1e-8)
    rep.finish()


def test_edge_calls_vs_reference_golden(A):
    """Edge cases of the OT helpers and of QKVAttention against the reference (edge_calls.npz, the calls are oracle/detfill.py:
    edge_calls): batch_ot_gmm's weight / variance validation, sinkhorn_log with 0 / 1 iterations, a threshold that stops at once,
    float32 inputs, apply_transport's shape check / zero noise / diagonal operator, a width the attention heads do not divide --
    the value the reference returns, or an exception of the type it raises (a shape mismatch the reference only notices inside a
    torch product, as a RuntimeError, may be refused earlier here, as a ValueError)."""
    from detfill import edge_calls
    from ot_vae_lightning_amd.ot import w2_utils as W2
    from ot_vae_lightning_amd.networks import nets_utils as NU
    z = load_golden("edge_calls.npz")
    rep = Report("edge calls vs the reference")
    for name, fn in edge_calls(W2, NU, dev="cuda").items():
        if f"{name}/error" in z.files:
            want = bytes(z[f"{name}/error"].astype("uint8")).decode()
            with pytest.raises(Exception) as info:
                fn()
            got = type(info.value)
            ok = any(c.__name__ == want for c in got.__mro__) or (want == "RuntimeError" and issubclass(got, ValueError))
            assert ok, (name, got.__name__, want)
            continue
        res = fn()
        for i, v in enumerate(res if isinstance(res, tuple) else (res,)):
            want_v = torch.from_numpy(z[f"{name}/value{i}"])
            tol = 2e-5 if want_v.dtype == torch.float32 else 1e-8
            rep.check(f"{name}[{i}]", v, want_v, tol)
    rep.finish()


def test_gaussian_transport_1024_dims_vs_oracle(A):
    """W2 + transport operator at the reference's latent-transport test size (D = 1024, transport_dims (1,2,3))."""
    import otvae_oracle as O
    rep = Report("GaussianTransport D=1024 vs CPU oracle (fp64)")
    D, B = 1024, 3000
    g = torch.Generator().manual_seed(9)
    mix = torch.randn(D, D, generator=g, dtype=torch.float64) / D ** 0.5
    src = torch.randn(B, D, generator=g, dtype=torch.float64) @ mix * 1.5 + 0.3
    tgt = torch.randn(B, D, generator=g, dtype=torch.float64)
    op = A.GaussianTransport(D, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double),
                             transport_cfg=dict(make_pd=True)).cuda()
    op.update(source_samples=src.cuda(), target_samples=tgt.cuda())
    w2 = op.compute()
    n_s, sx_s, sxx_s = O.gaussian_stats(src)
    n_t, sx_t, sxx_t = O.gaussian_stats(tgt)
    ms, cs = O.gaussian_fit(n_s, sx_s, sxx_s)
    mt, ct = O.gaussian_fit(n_t, sx_t, sxx_t)
    rep.check("W2^2", w2, O.w2_gaussian(ms, mt, cs, ct, make_pd=True), tol=1e-8)
    T = O.transport_operator_full(cs, ct)
    probe = src[:64]
    rep.check("transported samples", op.transport(probe.cuda()), O.apply_transport(probe, ms, mt, T), tol=1e-7)
    rep.finish()


def test_sinkhorn_persistent_equals_multilaunch(A):
    """The one-launch solvers (rows in registers, the potential in LDS: the flag-in-data exchange that fp32 problems with
    a fixed iteration count take by default, the grid-barrier exchange (OTVAE_SK_BARRIER, and whenever an early exit is
    tracked), each rows-per-wave variant, the first-generation 256-thread shape) and the launch-per-half-iteration solver
    run the same arithmetic per row: plan, potentials and iteration count must be identical bits, including the early
    exit.  Problems the persistent shapes cannot hold (rows > 1024, more rows than 128 workgroups take) fall back to
    launches by themselves."""
    import os
    from ot_vae_lightning_amd.ot import w2_utils as W
    cases = [((), 1024, 1024, torch.float32, 0.05, 50, 0.0),      # bench shape: 1 row per wave, 128 workgroups
             ((), 256, 256, torch.float32, 0.05, 50, 0.0),         # CIFAR per-GPU batch: 1 row per wave
             ((3,), 200, 300, torch.float32, 0.05, 7, 0.0),        # batched, non-square, fixed count: flag-in-data, 2 rows/wave
             ((), 64, 1000, torch.float32, 0.05, 0, 0.0),          # no iteration, flag-in-data plan
             ((), 100, 37, torch.float64, 0.1, 200, 1e-9),         # non-square, early exit
             ((2, 3), 33, 65, torch.float32, 0.05, 300, 1e-3),     # batched: min-over-batch early exit
             ((), 1024, 1024, torch.float64, 0.05, 10, 0.0),       # fp64 at the bench shape: 128 workgroups
             ((), 1500, 700, torch.float32, 0.05, 20, 0.0),        # rows > 1024: not register-resident
             ((), 7, 5, torch.float64, 1.0, 0, 0.0)]               # no iteration at all
    variants = [{}, {"OTVAE_SK_RPW": "2"}, {"OTVAE_SK_RPW": "4"}, {"OTVAE_SK_BARRIER": "1"}, {"OTVAE_SK_BARRIER": "1", "OTVAE_SK_RPW": "2"},
                {"OTVAE_SK_PERSISTENT": "256"}]
    for lead, n, m, dt, reg, it, thr in cases:
        a, b, C = _sinkhorn_problem(lead, n, m, dt, seed=n + m)
        a, b, C = a.cuda(), b.cuda(), C.cuda()
        if dt == torch.float32 and reg < 0.1:
            C = C / C.max()
        os.environ["OTVAE_SK_MULTILAUNCH"] = "1"
        try:
            ref = W.sinkhorn_log_potentials(a, b, C, reg=reg, max_iter=it, threshold=thr)
        finally:
            del os.environ["OTVAE_SK_MULTILAUNCH"]
        for env in variants:
            os.environ.update(env)
            try:
                got = W.sinkhorn_log_potentials(a, b, C, reg=reg, max_iter=it, threshold=thr)
            finally:
                for k in env:
                    del os.environ[k]
            torch.cuda.synchronize()
            assert int(got[3]) == int(ref[3]) and int(got[3]) >= 0, (n, m, env, int(got[3]), int(ref[3]))
            for g, r, name in zip(got[:3], ref[:3], ("pi", "u", "v")):
                assert torch.equal(g, r), (n, m, env, name, float((g - r).abs().max()))
            assert torch.isfinite(got[0]).all()


# ------------------------------------------------------------------------------------------------ G9 codebook k-means
@pytest.mark.parametrize("tag", ["sum", "ema"])
def test_codebook_model_kmeans_vs_reference_golden(A, tag):
    """CodebookModel.update x 6 -> fit -> predict / distribution / w2 against the reference's own class
    (same vec_init through load_state_dict, same host randperm seed for the first-call initialisation)."""
    g = group(load_golden("codebook_kmeans.npz"), tag)
    rep = Report(f"CodebookModel streaming k-means ({tag}) vs reference golden")
    K, d, B, decay = g["cfg"].tolist()
    K, d, B = int(K), int(d), int(B)
    decay = None if decay < 0 else float(decay)
    batches = g["batches"]
    lead = tuple(batches.shape[1:-2])
    model = A.CodebookModel(*lead, d, update_decay=decay,
                            mixture_cfg=dict(n_components=K, training_mode="argmax", inference_mode="argmax"))
    sd = model.state_dict()
    sd["vec_init"], sd["mat_init"], sd["codebook"] = g["vec_init"], g["mat_init"], g["vec_init"].clone()
    model.load_state_dict(sd)
    model = model.cuda().train()
    for step in range(batches.shape[0]):
        if step == 0:
            torch.manual_seed(1234)
        model.update(batches[step].cuda())
        rep.check(f"step{step}/n_obs", model._n_obs, g[f"step{step}/n_obs"], 1e-6)
        rep.check(f"step{step}/running_sum", model._running_sum, g[f"step{step}/running_sum"], 1e-5)
        rep.check(f"step{step}/codebook", model.codebook, g[f"step{step}/codebook"], 1e-5)
    model.fit()
    rep.check("fit/codebook", model.codebook, g["fit/codebook"], 1e-5)
    model.eval()
    probe = batches[-1].cuda()
    preds, sampled, dist = model.predict(probe)
    rep.check("predict/preds", preds, g["predict/preds"], 1e-5)
    # the weights are softmax(1/distance): a distance error e moves the exponent by e/distance^2.  The reference's
    # torch.cdist takes the |x|^2 + |c|^2 - 2 x.c route for more than 25 rows, whose cancellation error (~1e-6 at these
    # magnitudes) is amplified ~10-100x for samples within 0.1-0.3 of an atom; the kernel differences coordinates directly.
    rep.check("predict/probs", dist.probs, g["predict/probs"], 5e-4)
    _, ent = model.assignment_probs(probe, with_entropy=True)
    rep.check("predict/entropy", ent, g["predict/entropy"], 5e-4, floor=1e-3)
    assert sampled.shape == probe.shape[:-1] and int(sampled.min()) >= 0 and int(sampled.max()) < K
    enc, idx = model.nearest(probe)
    assert torch.equal(idx.cpu(), g["predict/probs"].argmax(-1))
    rep.check("weights", model.weights, g["weights"], 1e-6)
    other = A.CategoricalEmbeddings(g["centres"].cuda(), probs=(torch.ones(*lead, K) / K).cuda())
    rep.check("w2", model.w2(other), g["w2"], 1e-4)
    rep.finish()


@pytest.mark.parametrize("D,lead", [(24, (3,)), (128, ()), (200, ())])
def test_transport_compute_from_one_decomposition_per_covariance(A, D, lead):
    """``GaussianTransport.compute`` forms W2^2 and the eq. 17 operator from ONE eigendecomposition per covariance
    (``GaussianModel.cov_spectrum`` + ``w2_and_transport_operator``); the public ``w2_gaussian`` /
    ``compute_transport_operators`` on the ``cov`` attributes (ten decompositions) must give the same numbers, for the
    LDS Jacobi (D <= 128), the block Jacobi (D = 200) and batched operators."""
    from ot_vae_lightning_amd.ot.matrix_utils import eigh_vectors, spectral_fn
    rep = Report(f"transport compute, single decomposition, D={D} lead={lead}")
    g = torch.Generator().manual_seed(11 + D)
    B = 3 * D
    mix = torch.randn(*lead, D, D, generator=g, dtype=torch.float64) / D ** 0.5
    src = (torch.randn(*lead, B, D, generator=g, dtype=torch.float64) @ mix * 1.3 + 0.2).cuda()
    tgt = torch.randn(*lead, B, D, generator=g, dtype=torch.float64).cuda()
    op = A.GaussianTransport(*lead, D, source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double),
                             transport_cfg=dict(make_pd=True)).cuda()
    op.update(source_samples=src, target_samples=tgt)
    w2 = op.compute()
    s, t = op.source_model, op.target_model
    cov, lam, vt = s.cov_spectrum()
    rep.check("cov_spectrum: cov is the parametrised attribute", cov, s.cov, 1e-15)
    rep.check("cov_spectrum: V diag(lambda) V^T = cov", spectral_fn(lam, vt), cov, 1e-12)
    rep.check("W2^2 vs w2_gaussian", w2, op.w2_gaussian(s.mean, t.mean, s.cov, t.cov), 1e-10)
    T_ref, _ = op.compute_transport_operators(s.cov, t.cov)
    rep.check("operator vs compute_transport_operators", op.transport_operator, T_ref, 1e-9)
    assert op.cov_stochastic_noise.shape == T_ref.shape and not bool(op.cov_stochastic_noise.any())
    lam2, vt2 = eigh_vectors(cov)
    gram = vt2 @ vt2.transpose(-1, -2)
    rep.check("eigenvector rows are orthonormal", gram, torch.eye(D, dtype=torch.float64, device="cuda").expand_as(gram), 1e-11)
    rep.finish()


def test_codebook_model_recovers_mixture_centres(A):
    """The reference's own acceptance test (tests/test_distribution_models.py:190-211): stream batches of a mixture
    through update(), fit(), then the entropic W2 to the true atoms must be small."""
    torch.manual_seed(0)
    K, d, N = 8, 2, 8000
    g = torch.Generator().manual_seed(3)
    centres = torch.randn(K, d, generator=g) * 6
    which = torch.randint(0, K, (N,), generator=g)
    samples = centres[which] + 0.2 * torch.randn(N, d, generator=g)
    model = A.CodebookModel(d, mixture_cfg=dict(n_components=K, training_mode="argmax")).cuda().train()
    for i in range(0, N, 100):
        model.update(samples[i:i + 100].cuda())
    model.fit()
    w2 = model.w2(A.CategoricalEmbeddings(centres.cuda(), probs=(torch.ones(K) / K).cuda()))
    # k-means from a random initialisation may merge two clusters; the streaming estimate must still be close
    assert float(w2) < 3.0, float(w2)
    enc, idx = model.nearest(samples[:512].cuda())
    assert float((enc - samples[:512].cuda()).norm(dim=-1).mean()) < 2.0


@pytest.mark.parametrize("T,H,C", [(512, 1, 1), (512, 2, 2), (260, 2, 1), (260, 1, 2), (48, 3, 2), (36, 2, 1), (1024, 2, 2),
                                   (5, 1, 1), (128, 3, 1), (768, 1, 2)])
def test_attention_slice_layouts_vs_float64(A, T, H, C):
    """Head widths 1 and 2 (the kernels that carry the forward key moments) at token counts that put 1, 2, 4 ... slices
    in a workgroup, slices that straddle waves (T/4 = 65), non-power-of-two lane groups (48, 36) and a last workgroup
    with fewer slices than the others (3 images): forward and the three gradients against the float64 formula
    (reference networks/nets_utils.py:63-82) through autograd."""
    import otvae_oracle as oracle
    rep = Report(f"attention slice layouts T={T} H={H} C={C}")
    N = 3
    g = torch.Generator().manual_seed(7 * T + H + C)
    qkv = torch.randn(N, 3 * H * C, T, generator=g)
    qkv[:, 2 * H * C:] += 3.0            # values far from zero: the shifted-data moments must not lose the spread
    gout = torch.randn(N, H * C, T, generator=g)
    ref_in = qkv.double().requires_grad_(True)
    ref = oracle.qkv_attention(ref_in, H)
    ref.backward(gout.double())
    x = qkv.cuda().requires_grad_(True)
    out = A.QKVAttention(H)(x)
    out.backward(gout.cuda())
    rep.check("out", out, ref.detach(), 1e-5)
    w = 2 * H * C
    # the same formula in fp32 (what the reference computes) for the evidence-based bound: round 3 held this gradient to 2e-5, a
    # constant 1 % above the measured 1.965e-5
    ref32_in = qkv.clone().requires_grad_(True)
    oracle.qkv_attention(ref32_in, H).backward(gout)
    rep.check_vs_truth("d/dq, d/dk", x.grad[:, :w], ref32_in.grad[:, :w], ref_in.grad[:, :w])
    rep.check("d/dv", x.grad[:, w:], ref_in.grad[:, w:], 1e-5)
    rep.finish()


@pytest.mark.parametrize("T,H,C", [(64, 2, 8), (16, 8, 4), (66, 4, 32), (4, 8, 8)])
def test_attention_peaked_rows_vs_float64(A, T, H, C):
    """Round 4: softmax rows that are PEAKED (scores of order +-30: one key takes almost all the weight) are where the backward's
    delta = gout . out must be consistent with the p and dP it recomputes (csrc/attention.hip phase A): the true dP* - delta is
    (1 - p*) x something, an inconsistent delta leaves an error eps |dP| there, the same sign for every key of the row.  Found through
    the ViT step's LayerNorm gradients (profiles/r04_vit_attention_delta.txt); pinned here on the kernel itself: every gradient within
    the fp32 contract or 1.5x of what torch's own fp32 softmax backward leaves against the float64 truth."""
    import otvae_oracle as oracle
    rep = Report(f"attention, peaked rows T={T} H={H} C={C}")
    N = 5
    g = torch.Generator().manual_seed(31 * T + H + C)
    qkv = torch.randn(N, 3 * H * C, T, generator=g)
    qkv[:, :2 * H * C] *= 6.0 * C ** 0.25            # q and k: scores q.k / sqrt(C) of order +-36
    qkv[:, 2 * H * C:] += 2.0
    gout = torch.randn(N, H * C, T, generator=g)
    truth_in = qkv.double().requires_grad_(True)
    truth = oracle.qkv_attention(truth_in, H)
    truth.backward(gout.double())
    ref_in = qkv.clone().requires_grad_(True)
    oracle.qkv_attention(ref_in, H).backward(gout)
    x = qkv.cuda().requires_grad_(True)
    out = A.QKVAttention(H)(x)
    out.backward(gout.cuda())
    pmax = torch.softmax((truth_in[:, :H * C].reshape(N * H, C, T).transpose(1, 2) @ truth_in[:, H * C:2 * H * C].reshape(N * H, C, T))
                         / C ** 0.5, -1).amax(-1)
    assert float(pmax.median()) > 0.9, "the rows of this test are meant to be peaked"
    w = H * C
    rep.check("out", out, truth.detach(), 1e-5)
    for name, sl in (("d/dq", slice(0, w)), ("d/dk", slice(w, 2 * w)), ("d/dv", slice(2 * w, 3 * w))):
        rep.check_vs_truth(name, x.grad[:, sl], ref_in.grad[:, sl], truth_in.grad[:, sl])
        rep.check_vs_truth(name + " (rel L2)", x.grad[:, sl], ref_in.grad[:, sl], truth_in.grad[:, sl], l2=True)
    rep.finish()


@pytest.mark.parametrize("T,H,C", [(1024, 1, 1), (256, 4, 2), (64, 4, 4), (1, 16, 16)])
def test_attention_full_batch_properties(A, T, H, C):
    """Batch 1024 (BASELINE size), the (T, heads, width) shapes of the MNIST network: properties that need no restatement.
    (a) the attention weights of a row sum to one: v == const  =>  out == const, and then d out / d(q, k) == 0 while the
    v-gradient of every (image, head, channel) sums to the sum of the output gradient; (b) images never mix: permuting the
    batch permutes out and the gradient (bit for bit); (c) against the float64 formula on 4 images (reference
    networks/nets_utils.py:63-82)."""
    rep = Report(f"attention properties at batch 1024, T={T} H={H} C={C}")
    N = 1024
    g = torch.Generator().manual_seed(T + H + C)
    qkv = torch.randn(N, 3 * H * C, T, generator=g).cuda()     # [N, 3*H*C, T] as QKVAttention takes it
    gout = torch.randn(N, H * C, T, generator=g).cuda()
    attn = A.QKVAttention(H)
    # (a)
    const = qkv.clone()
    const[:, 2 * H * C:] = 0.75
    const.requires_grad_(True)
    out = attn(const)
    # fp32 accumulation of T weighted terms and of their normaliser: <= ~T^(1/2) ulp typical, T/4 ulp worst case
    rep.check("v const -> out const", out, torch.full_like(out, 0.75), max(2e-6, T * 1.5e-8))
    out.backward(gout)
    gq = const.grad
    rep.check("v const -> d/dq, d/dk = 0", gq[:, :2 * H * C], torch.zeros_like(gq[:, :2 * H * C]), 1e-5, floor=float(gout.abs().max()))
    rep.check("sum_s dv_s = sum_t gout_t", gq[:, 2 * H * C:].double().sum(2), gout.double().sum(2), 1e-5)
    # (b)
    x1 = qkv.clone().requires_grad_(True)
    o1 = attn(x1)
    o1.backward(gout)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(2)).cuda()
    x2 = qkv[perm].clone().requires_grad_(True)
    o2 = attn(x2)
    o2.backward(gout[perm])
    assert torch.equal(o2, o1[perm]) and torch.equal(x2.grad, x1.grad[perm])
    # (c) float64 formula on the first 4 images
    q, k, v = qkv[:4].double().cpu().reshape(4, 3, H, C, T).unbind(1)   # chunk(3) along channels, then head-major
    sc = 1.0 / C ** 0.5  # the reference scales q AND k by C^-1/2 (nets_utils.py:72-73)
    w = torch.einsum("nhct,nhcs->nhts", q * sc, k * sc).softmax(-1)
    ref = torch.einsum("nhts,nhcs->nhct", w, v).reshape(4, H * C, T)
    rep.check("float64 formula (4 images)", o1[:4], ref, 1e-5)
    rep.finish()


# ------------------------------------------------------------------------------------------------ G10 discrete transport
def _zero_init(model):
    sd = model.state_dict()
    sd["codebook"] = sd["vec_init"].clone()
    model.load_state_dict(sd)
    return model


def test_codebook_model_mean_mode_vs_reference_golden(A):
    """Soft ('mean') assignment: three streaming updates and the inference prediction against the reference's class.
    Temperature 0.5 keeps softmax(1 / distance / T) insensitive to the rounding of the reference's cdist."""
    g = group(load_golden("discrete.npz"), "mean")
    rep = Report("CodebookModel 'mean' mode vs reference golden")
    K, d, B, temp = g["cfg"].tolist()
    K, d = int(K), int(d)
    batches = g["batches"]
    lead = tuple(batches.shape[1:-2])
    model = _zero_init(A.CodebookModel(*lead, d, mixture_cfg=dict(n_components=K, training_mode="mean", inference_mode="mean",
                                                                  temperature=temp))).cuda().train()
    for step in range(batches.shape[0]):
        if step == 0:
            torch.manual_seed(77)
        model.update(batches[step].cuda())
        rep.check(f"step{step}/n_obs", model._n_obs, g[f"step{step}/n_obs"], 2e-5)
        rep.check(f"step{step}/codebook", model.codebook, g[f"step{step}/codebook"], 2e-5)
    model.eval()
    preds, sampled, dist = model.predict(batches[-1].cuda())
    rep.check("predict/probs", dist.probs, g["predict/probs"], 5e-5)
    rep.check("predict/preds", preds, g["predict/preds"], 2e-5)
    rep.finish()


@pytest.mark.parametrize("ttype", ["mean", "argmax"])
def test_discrete_transport_vs_reference_golden(A, ttype):
    """DiscreteTransport (reference ot/transport/discrete_transport.py; configuration of tests/test_latent_transport.py:
    92-101: soft training mode at temperature 1e-2): three updates per side, compute (atom-to-atom cost, Sinkhorn plan,
    total), transport of a probe batch.  At temperature 1e-2 the soft weights are softmax(100 / distance): the ~1e-6
    cancellation error of the reference's cdist (|x|^2 + |c|^2 - 2 x.c for more than 25 rows) moves an exponent by up to
    1e-2 for samples close to an atom, so the fitted codebooks are compared at 2e-3; everything downstream is compared
    at tight tolerance GIVEN the reference's codebooks (loaded into the operator)."""
    import otvae_oracle as O
    g = group(load_golden("discrete.npz"), f"dt_{ttype}")
    rep = Report(f"DiscreteTransport transport_type={ttype} vs reference golden")
    K, d = g["source_codebook"].shape
    mix = dict(n_components=K, training_mode="mean", inference_mode="argmax", temperature=1e-2)
    op = A.DiscreteTransport(d, source_cfg=dict(mixture_cfg=mix), target_cfg=dict(mixture_cfg=mix), transport_type=ttype,
                             sinkhorn_reg=1e-2, sinkhorn_max_iter=200, sinkhorn_threshold=1e-9)
    _zero_init(op.source_model)
    _zero_init(op.target_model)
    op = op.cuda().train()
    for i in range(g["src"].shape[0]):
        if i == 0:
            torch.manual_seed(78)
        op.update(source_samples=g["src"][i].cuda())
        if i == 0:
            torch.manual_seed(178)
        op.update(target_samples=g["tgt"][i].cuda())
    cost = op.compute()
    rep.check("fitted source codebook", op.source_model.codebook, g["source_codebook"], 2e-3)
    rep.check("fitted target codebook", op.target_model.codebook, g["target_codebook"], 2e-3)
    rep.check("source probs", op.source_distribution.probs, g["source_probs"], 2e-3)
    rep.check("total cost (own codebooks)", cost, g["cost"], 5e-3)
    # the same updates in float64 on the CPU (the formula without cdist's shortcut): the kernels must sit on it
    for side, model, seed in (("src", op.source_model, 78), ("tgt", op.target_model, 178)):
        B = g[side].shape[-2]
        torch.manual_seed(seed)
        idx = torch.randperm(B)[:K]
        st = {"codebook": torch.zeros(K, d, dtype=torch.float64), "vec_init": torch.zeros(K, d, dtype=torch.float64),
              "n_obs": torch.zeros(K, dtype=torch.float64), "running_sum": torch.zeros(K, d, dtype=torch.float64)}
        for i in range(g[side].shape[0]):
            st = O.codebook_update(st, g[side][i].double(), None, rand_indices=idx if i == 0 else None, temperature=1e-2, mode="mean")
        rep.check(f"{side}: codebook vs float64 formula", model.codebook, O.codebook_fit(st)["codebook"], 2e-5)
    # downstream of the reference's own codebooks
    with torch.no_grad():
        op.source_model.codebook.copy_(g["source_codebook"].cuda())
        op.target_model.codebook.copy_(g["target_codebook"].cuda())
        op.source_model._n_obs.copy_(g["source_probs"].cuda() * op.source_model._n_obs.sum())
        op.target_model._n_obs.copy_(g["target_probs"].cuda() * op.target_model._n_obs.sum())
        op.source_model._running_sum.copy_(op.source_model.codebook * 0)   # fit() must leave the loaded atoms alone
        op.target_model._running_sum.copy_(op.target_model.codebook * 0)
    op.source_model.kmeans_iter = op.target_model.kmeans_iter = 0
    cost = op.compute()
    rep.check("plan", op.transport_matrix, g["plan"], 1e-5)
    rep.check("total cost", cost, g["cost"], 1e-5)
    moved = op.transport(g["probe"].cuda())
    rep.check("transported probe", moved, g["moved"], 1e-5)
    assert op.training
    op.reset()
    assert op.transport_matrix is None
    with pytest.raises(RuntimeError):
        op.transport(g["probe"].cuda())
    rep.finish()


def test_codebook_prior_vs_reference_golden(A):
    """CodebookPrior (reference prior/codebook.py) in the one-hot mode: two training steps (streaming k-means update of
    the shared codebook, quantised latents, 'l2' loss with cosine annealing, straight-through gradient) and the
    evaluation-mode entropy losses."""
    g = group(load_golden("discrete.npz"), "prior")
    rep = Report("CodebookPrior vs reference golden")
    size, K = tuple(g["step0/x"].shape[1:]), g["step0/codebook"].shape[-2]
    prior = A.CodebookPrior(size, (1, 2, 3), loss="l2", loss_coeff=0.5, annealing_steps=10,
                            mixture_cfg=dict(n_components=K, training_mode="argmax", inference_mode="argmax"))
    _zero_init(prior.codebook_model)
    prior = prior.cuda().train()
    w = g["w"].cuda()
    for step in range(2):
        x = g[f"step{step}/x"].cuda().requires_grad_(True)
        if step == 0:
            torch.manual_seed(79)
        z, loss, art = prior(x, step=3 + step)
        ((z * w).sum() + loss.sum()).backward()
        rep.check(f"step{step}/z", z, g[f"step{step}/z"], 1e-6)
        rep.check(f"step{step}/loss", loss, g[f"step{step}/loss"], 1e-5)
        rep.check(f"step{step}/dx", x.grad, g[f"step{step}/gx"], 1e-5)
        rep.check(f"step{step}/probs", art["distribution"].probs, g[f"step{step}/probs"], 5e-4)
        rep.check(f"step{step}/codebook", prior.codebook_model.codebook, g[f"step{step}/codebook"], 1e-5)
        assert art["indices"].shape == (x.shape[0], 1)
    prior.eval()
    xe = g["eval/x"].cuda()
    # fp64 TRUTH of the two entropy losses: the same formula -- distances by exact differences, energy =
    # 1 / (|x - c| + 1e-8), softmax(energy / T), loss_coeff (log K - entropy) -- in float64 on the CPU from the eval batch and the
    # codebook the golden recorded.
    with torch.no_grad():   # both sides evaluate on exactly the reference's codebook (the trained one agrees with it to 1e-5)
        prior.codebook_model.codebook.copy_(g["step1/codebook"].cuda().reshape(prior.codebook_model.codebook.shape))
    x3 = prior.permute_and_flatten(xe).double().cpu()                                  # [P, B, d]
    cb = prior.codebook_model.codebook.detach().double().cpu()
    dist = (x3.unsqueeze(-2) - cb.unsqueeze(-3)).square().sum(-1).sqrt()               # [P, B, K]
    pr = torch.softmax((1.0 / (dist + 1e-8)) / float(prior.codebook_model.temperature), dim=-1)
    ent = -(pr * pr.clamp_min(1e-300).log()).sum(-1)                                   # [P, B]
    truth = {"kl": 0.5 * (math.log(K) - ent).sum(0), "first_kl": 0.5 * (math.log(K) - ent)[0]}
    for kind in ("kl", "first_kl"):
        prior.loss = kind
        z, loss, _ = prior(xe, step=100)
        # MEASURED (round 3): neither side is "the closer one".  loss = log K - H is a cancellation (H within a few per cent of
        # log K on this batch), so any fp32 evaluation of H carries eps * H / loss ~ 3e-4 of relative error into it: the
        # reference's fp32 value sits 3.0e-4 from the fp64 truth, the HIP value 3.5e-4, on opposite sides -- hence up to 7e-4
        # between them and the 2e-3 bound against the golden.  1e-4 is not attainable for this quantity in fp32 by either.
        e_hip, e_ref = rel_err(loss.double().cpu(), truth[kind]), rel_err(g[f"eval/loss_{kind}"].double(), truth[kind])
        # the rule of VERDICT r3 #7: the fp32 contract or 1.5x the reference's own fp32 error against the fp64 truth (round 3 had
        # "< 1e-3 and <= 2x the reference's", and 2e-3 against the golden)
        tol = max(TOL32, 1.5 * e_ref)
        rep.rows.append((f"eval/loss_{kind}: HIP vs fp64 truth [reference fp32 vs truth: {e_ref:.2e}]", e_hip, tol, e_hip <= tol))
        assert e_hip <= tol, (kind, e_hip, e_ref)
        # against the golden itself nothing tighter than the two errors' sum can hold (they sit on opposite sides of the truth)
        rep.check(f"eval/loss_{kind} vs reference golden", loss, g[f"eval/loss_{kind}"], 1.05 * (tol + e_ref))
    rep.check("eval/z", z, g["eval/z"], 1e-6)
    xg = xe.clone().requires_grad_(True)                      # the entropy losses are differentiable (otvae_codebook_probs_bwd)
    prior(xg, step=100)[1].sum().backward()
    assert torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0
    assert prior.sample((4, *size), "cuda").shape == (4, *size)
    with pytest.raises(ValueError):
        A.CodebookPrior(size, (4,), mixture_cfg=dict(n_components=K))
    rep.finish()


def test_soft_codebook_prior_and_gumbel_modes_vs_reference_golden(A):
    """SURVEY 8f-2, beyond one-hot assignments: CodebookPrior in the soft 'mean' training mode with the 'kl' entropy loss -- the
    gradient reaches the encoder through ``otvae_codebook_probs_bwd`` -- and the 'gumbel-softmax' / 'gumbel-hardmax' assignment
    modes of CodebookModel / GaussianMixtureModel with the reference's own Gumbel draws injected (tests/golden/mixture_modes.npz)."""
    G = load_golden("mixture_modes.npz")
    rep = Report("soft CodebookPrior + Gumbel assignment modes vs reference golden")
    sp = group(G, "soft_prior")
    size, K = tuple(sp["step0/x"].shape[1:]), sp["step0/codebook"].shape[-2]
    prior = A.CodebookPrior(size, (1, 2, 3), loss="kl", loss_coeff=0.7,
                            mixture_cfg=dict(n_components=K, training_mode="mean", inference_mode="mean", temperature=0.5))
    _zero_init(prior.codebook_model)
    prior = prior.cuda().train()
    w = sp["w"].cuda()
    for step in range(2):
        x = sp[f"step{step}/x"].cuda().requires_grad_(True)
        if step == 0:
            torch.manual_seed(179)
        z, loss, art = prior(x, step=step)
        ((z * w).sum() + loss.sum()).backward()
        rep.check(f"soft prior step{step}/codebook", prior.codebook_model.codebook, sp[f"step{step}/codebook"], 1e-5)
        rep.check(f"soft prior step{step}/z", z, sp[f"step{step}/z"], 1e-4)
        rep.check(f"soft prior step{step}/loss", loss, sp[f"step{step}/loss"], 1e-4)
        rep.check(f"soft prior step{step}/dx", x.grad, sp[f"step{step}/gx"], 1e-4)
        rep.check(f"soft prior step{step}/probs", art["distribution"].probs, sp[f"step{step}/probs"], 5e-4)
    for mode in ("gumbel-softmax", "gumbel-hardmax"):
        g = group(G, f"codebook/{mode}")
        m = A.CodebookModel(5, mixture_cfg=dict(n_components=6, training_mode=mode, temperature=0.7)).cuda().train()
        with torch.no_grad():
            m.codebook.copy_(g["codebook"].cuda())
        x = g["x"].cuda().requires_grad_(True)
        m.gumbel_noise = g["gumbel"].cuda()
        wts, _, _ = m.assign(x)
        (wts * g["w"].cuda()).sum().backward()
        rep.check(f"codebook {mode}: weights", wts, g["weights"], 1e-5)
        rep.check(f"codebook {mode}: dx", x.grad, g["gx"], 1e-4)
        assert m.gumbel_noise is None
        free, _, _ = m.assign(x.detach())                    # drawn on the device: a valid assignment
        assert torch.allclose(free.sum(-1), torch.ones_like(free.sum(-1)), atol=1e-5)
        g = group(G, f"gmm/{mode}")
        mm = A.GaussianMixtureModel(5, w2_cfg={"diag": True}, dtype=torch.double,
                                    mixture_cfg=dict(n_components=4, training_mode=mode, temperature=1.3)).cuda().train()
        with torch.no_grad():
            mm.mean.copy_(g["mean"].cuda())
            mm.cov = g["var"].cuda()
            mm._weights = torch.tensor([0.1, 0.4, 0.3, 0.2], dtype=torch.double).cuda()
        mm.gumbel_noise = g["gumbel"].cuda()
        # 1e-7: the variances read back through the strictly-positive parametrisation (+1e-8) on both sides
        rep.check(f"gmm {mode}: weights", mm.assign(g["x"].cuda())[0], g["weights"], 1e-7)
    rep.finish()


# ------------------------------------------------------------------------------------------------ G11 Gaussian mixtures
_GMM_W2 = dict(diag=True, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)
_GMM_MIX = dict(metric="euclidean", p=2., topk=None, temperature=1., training_mode="argmax", inference_mode="argmax")


@pytest.mark.parametrize("tag", ["sum", "ema"])
def test_gaussian_mixture_model_vs_reference_golden(A, tag):
    """GaussianMixtureModel with diagonal covariances (reference gassian_mixture_model.py): three streaming updates (with
    and without EMA decay, with a leading dimension), fit, the mixture energy / assignment of a batch and w2 against
    another mixture -- golden vectors from the reference's own class, double precision."""
    g = group(load_golden("gmm.npz"), tag)
    rep = Report(f"GaussianMixtureModel ({tag}) vs reference golden")
    K, d, B, decay = g["cfg"].tolist()
    K, d = int(K), int(d)
    decay = None if decay < 0 else float(decay)
    batches = g["batches"]
    lead = tuple(batches.shape[1:-2])
    model = A.GaussianMixtureModel(*lead, d, mixture_cfg={**_GMM_MIX, "n_components": K}, w2_cfg=_GMM_W2, update_decay=decay,
                                   dtype=torch.double).cuda().train()
    for step in range(batches.shape[0]):
        if step == 0:
            torch.manual_seed(81)
        model.update(batches[step].cuda())
        rep.check(f"step{step}/n_obs", model._n_obs, g[f"step{step}/n_obs"], 1e-12)
        rep.check(f"step{step}/mean", model.mean, g[f"step{step}/mean"], 1e-12)
        rep.check(f"step{step}/cov", model.cov, g[f"step{step}/cov"], 1e-11)
        rep.check(f"step{step}/weights", model.weights, g[f"step{step}/weights"], 1e-12)
    model.fit()
    rep.check("fit/mean", model.mean, g["fit/mean"], 1e-12)
    rep.check("fit/cov", model.cov, g["fit/cov"], 1e-11)
    rep.check("fit/weights", model.weights, g["fit/weights"], 1e-12)
    model.eval()
    x = batches[-1].cuda()
    rep.check("energy", model.energy(x), g["energy"], 1e-12)
    weights, sampled, dist = model.assign(x)
    rep.check("assignment one-hot", weights, g["assign_onehot"], exact=True)
    rep.check("assignment probabilities", dist.probs, g["assign_probs"], 1e-10)
    assert sampled.shape == x.shape[:-1]
    centres = g["centres"].cuda()
    other = torch.distributions.MixtureSameFamily(
        torch.distributions.Categorical((torch.ones(*lead, K, dtype=torch.double) / K).cuda()),
        torch.distributions.Independent(torch.distributions.Normal(centres, torch.full_like(centres, 0.4)), 1))
    rep.check("w2", model.w2(other), g["w2"], 1e-8)
    model.reset()
    assert float(model._n_obs.sum()) == 0 and torch.allclose(model.mean, model.vec_init)
    rep.finish()


_GMM_W2_FULL = dict(diag=False, stochastic=False, pg_star=0., make_pd=True, verbose=False, dtype=torch.double)


def test_full_covariance_mixture_functions_vs_reference_golden(A):
    """SURVEY 8f-2, full covariances: ``batch_w2_dissimilarity_gaussian`` (N x M Gaussian W2 through one batched eigensolver
    call), ``batch_ot_gmm(diag=False)`` and ``gaussian_barycenter`` (diagonal closed form; full fixed point from the index the
    reference drew) against the reference's outputs (tests/golden/gmm_full.npz)."""
    from ot_vae_lightning_amd.ot import w2_utils as W
    f = group(load_golden("gmm_full.npz"), "fn")
    rep = Report("full-covariance mixture functions vs reference golden")
    c = {k: v.cuda() for k, v in f.items() if hasattr(v, "cuda")}
    rep.check("batch_w2_dissimilarity_gaussian", W.batch_w2_dissimilarity_gaussian(c["ms"], c["mt"], c["cs"], c["ct"], make_pd=True),
              f["dissimilarity"], 1e-9)
    total, plan = W.batch_ot_gmm(c["ms"], c["mt"], c["cs"], c["ct"], diag=False, weight_source=c["ws"], weight_target=c["wt"],
                                 max_iter=100)
    rep.check("batch_ot_gmm total", total, f["ot_total"], 1e-8)
    rep.check("batch_ot_gmm coupling", plan, f["ot_coupling"], 1e-7)
    mb, vb = W.gaussian_barycenter(c["ms"], torch.diagonal(c["cs"], dim1=-2, dim2=-1), c["ws"], diag=True)
    rep.check("barycenter (diag) mean", mb, f["bary_diag_mean"], 1e-12)
    rep.check("barycenter (diag) var", vb, f["bary_diag_var"], 1e-12)
    mb, cb = W.gaussian_barycenter(c["ms"], c["cs"], c["ws"], diag=False, n_iter=100, init_index=int(f["bary_init_index"]))
    rep.check("barycenter (full) mean", mb, f["bary_full_mean"], 1e-12)
    rep.check("barycenter (full) cov", cb, f["bary_full_cov"], 1e-8)
    with pytest.raises(ValueError):
        W.gaussian_barycenter(c["ms"], c["cs"], c["ws"] * 2, diag=False)
    rep.finish()


@pytest.mark.parametrize("tag", ["sum", "ema"])
def test_full_covariance_gaussian_mixture_model_vs_reference_golden(A, tag):
    """GaussianMixtureModel with FULL covariances (reference gassian_mixture_model.py with w2_cfg diag=False): streaming updates,
    fit, energy (one batched eigendecomposition instead of MultivariateNormal's Cholesky), assignment, predict_mean_var and
    w2 against another full-covariance mixture."""
    g = group(load_golden("gmm_full.npz"), tag)
    rep = Report(f"GaussianMixtureModel, full covariances ({tag}) vs reference golden")
    K, d, B, decay = g["cfg"].tolist()
    K, d = int(K), int(d)
    decay = None if decay < 0 else float(decay)
    batches = g["batches"]
    lead = tuple(batches.shape[1:-2])
    model = A.GaussianMixtureModel(*lead, d, mixture_cfg={**_GMM_MIX, "n_components": K}, w2_cfg=_GMM_W2_FULL, update_decay=decay,
                                   dtype=torch.double).cuda().train()
    assert model.cov.shape == (*lead, K, d, d)
    for step in range(batches.shape[0]):
        if step == 0:
            torch.manual_seed(83)
        model.update(batches[step].cuda())
        rep.check(f"step{step}/n_obs", model._n_obs, g[f"step{step}/n_obs"], 1e-12)
        rep.check(f"step{step}/mean", model.mean, g[f"step{step}/mean"], 1e-11)
        rep.check(f"step{step}/cov", model.cov, g[f"step{step}/cov"], 1e-9)
        rep.check(f"step{step}/weights", model.weights, g[f"step{step}/weights"], 1e-12)
    model.fit()
    rep.check("fit/mean", model.mean, g["fit/mean"], 1e-11)
    rep.check("fit/cov", model.cov, g["fit/cov"], 1e-9)
    rep.check("fit/weights", model.weights, g["fit/weights"], 1e-12)
    model.eval()
    x = batches[-1].cuda()
    rep.check("energy", model.energy(x), g["energy"], 1e-8)
    weights, sampled, dist = model.assign(x)
    rep.check("assignment one-hot", weights, g["assign_onehot"], exact=True)
    rep.check("assignment probabilities", dist.probs, g["assign_probs"], 1e-7)
    pm, pv = model.predict_mean_var(weights)
    rep.check("predict_mean_var: mean", pm, g["pred_mean"], 1e-11)
    rep.check("predict_mean_var: cov", pv, g["pred_cov"], 1e-9)
    centres, oc = g["centres"].cuda(), g["other_cov"].cuda()
    other = torch.distributions.MixtureSameFamily(
        torch.distributions.Categorical((torch.ones(*lead, K, dtype=torch.double) / K).cuda()),
        torch.distributions.MultivariateNormal(centres, covariance_matrix=oc))
    rep.check("w2", model.w2(other), g["w2"], 1e-7)
    rep.finish()


def test_gmm_transport_barycenter_and_full_covariances(A):
    """GMMTransport beyond the reference's tested configuration: transport_type='barycenter' (the reference hands the coupled
    assignment to gaussian_barycenter un-normalised, which its own validation rejects; here it is normalised) and full
    covariances.  Invariants: finite outputs of the input's shape; for a one-component target the barycentre IS that
    component, so 'barycenter' and 'argmax' transport agree; the full-covariance map pushes the source's component mean to
    the target's."""
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(91)
    mix = {**_GMM_MIX}
    for diag in (True, False):
        w2_cfg = dict(_GMM_W2, diag=diag)
        cfg = lambda k: dict(update_decay=None, update_with_autograd=False, dtype=torch.double, mixture_cfg={**mix, "n_components": k})  # noqa: E731
        d = 3
        src = torch.randn(400, d, generator=g, dtype=torch.double) * torch.tensor([1.0, 0.5, 2.0], dtype=torch.double)
        tgt = torch.randn(400, d, generator=g, dtype=torch.double) * 0.7 + 2.0
        outs = {}
        for ttype in ("barycenter", "argmax"):
            torch.manual_seed(11)
            op = A.GMMTransport(d, transport_type=ttype, transport_cfg=w2_cfg, source_cfg=cfg(2), target_cfg=cfg(1)).cuda().train()
            op.update(source_samples=src.cuda())
            op.update(target_samples=tgt.cuda())
            cost = op.compute()
            assert torch.isfinite(cost).all()
            op.eval()
            op.barycenter_init = 0
            outs[ttype] = op.transport(src[:64].cuda())
            assert outs[ttype].shape == (64, d) and torch.isfinite(outs[ttype]).all()
        assert rel_err(outs["barycenter"], outs["argmax"]) < 1e-7, diag
        moved_mean = outs["argmax"].mean(0).cpu()
        assert float((moved_mean - tgt.mean(0)).abs().max()) < 0.35, (diag, moved_mean)
        op3 = A.GMMTransport(d, transport_type="barycenter", transport_cfg=w2_cfg, source_cfg=cfg(2), target_cfg=cfg(3)).cuda().train()
        torch.manual_seed(12)
        op3.update(source_samples=src.cuda())
        op3.update(target_samples=torch.cat([tgt, tgt * 0.5 - 3.0]).cuda())
        op3.compute()
        op3.eval()
        op3.barycenter_init = 1
        moved = op3.transport(src[:16].cuda())
        assert moved.shape == (16, d) and torch.isfinite(moved).all()


def test_gmm_transport_vs_reference_golden(A):
    """GMMTransport (reference ot/transport/gmm_transport.py, configured as tests/test_latent_transport.py:80-91): three
    updates per side, compute (component coupling from the HIP Sinkhorn solver on the diagonal-Gaussian W2 cost),
    transport of a probe batch ('argmax': likeliest source component -> most coupled target component -> closed-form
    diagonal map)."""
    g = group(load_golden("gmm.npz"), "tr")
    rep = Report("GMMTransport vs reference golden")
    K, d = g["source_mean"].shape
    cfg = dict(update_decay=None, update_with_autograd=False, dtype=torch.double, mixture_cfg={**_GMM_MIX, "n_components": K})
    op = A.GMMTransport(d, transport_type="argmax", transport_cfg=_GMM_W2, source_cfg=cfg, target_cfg=cfg).cuda().train()
    for i in range(g["src"].shape[0]):
        if i == 0:
            torch.manual_seed(82)
        op.update(source_samples=g["src"][i].cuda())
        if i == 0:
            torch.manual_seed(182)
        op.update(target_samples=g["tgt"][i].cuda())
    total = op.compute()
    for side, m in (("source", op.source_model), ("target", op.target_model)):
        rep.check(f"{side} means", m.mean, g[f"{side}_mean"], 1e-12)
        rep.check(f"{side} variances", m.cov, g[f"{side}_cov"], 1e-11)
        rep.check(f"{side} weights", m.weights, g[f"{side}_weights"], 1e-12)
    rep.check("coupling", op.transport_matrix, g["coupling"], 1e-8)
    rep.check("total cost", total, g["total"], 1e-8)
    op.eval()
    moved = op.transport(g["probe"].cuda())
    assert moved.dtype == torch.float32
    rep.check("transported probe", moved, g["moved"], 1e-5)
    op.reset()
    with pytest.raises(RuntimeError):
        op.transport(g["probe"].cuda())
    with pytest.raises(NotImplementedError):
        A.GMMTransport(d, transport_type="nearest", transport_cfg=_GMM_W2, source_cfg=cfg, target_cfg=cfg)
    rep.finish()


# ------------------------------------------------------------------------------------------------ G12 ViT
@pytest.mark.parametrize("tag", ["d32", "d128"])
@pytest.mark.parametrize("role", ["enc", "dec"])
def test_vit_vs_reference_golden(A, tag, role):
    """The reference's ViT (networks/vit.py) as encoder and as decoder, configured like tests/test_conditional_vit_vae.py:
    41-67 with dropout 0: same constructor arguments, the reference's state_dict keys filled with the closed-form weights,
    forward output, input gradient and every parameter gradient against the golden vectors.  d32: head width 8, 22
    tokens, class token; d128: head width 32 (the attention kernel's widest instantiation), 18 tokens."""
    from detfill import fill_vit_state_dict
    from test_oracle_vs_golden import VIT_CASES, VIT_ROLES, check_vit_grads, vit_param_shapes
    g = load_golden("vit.npz")
    g = {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "/")}
    rep = Report(f"ViT {tag} {role} vs reference golden")
    cfg = VIT_CASES[tag]
    net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., **cfg, **VIT_ROLES[role])
    want = vit_param_shapes(cfg, role)
    sd = net.state_dict()
    assert set(sd.keys()) == set(want.keys()) and all(tuple(sd[k].shape) == tuple(s) for k, s in want.items())
    ordered = {k: torch.zeros(s) for k, s in want.items()}          # the reference's key order decides the fill phases
    fill_vit_state_dict(ordered)
    net.load_state_dict(ordered)
    net = net.cuda().train()
    x = g[f"{role}/x"].cuda().requires_grad_(True)
    labels = g["labels"].cuda() if "labels" in g else None
    y = net(x, labels=labels)
    y.backward(g[f"{role}/gy"].cuda())
    rep.check("output", y, g[f"{role}/y"], 1e-5)
    rep.check("input gradient", x.grad, g[f"{role}/gx"], 1e-4)
    rep.finish()
    check_vit_grads(g, role, {k: p.grad for k, p in net.named_parameters()}, 5e-4)  # column sums with cancellation
    assert tuple(net.out_size) == tuple(y.shape[1:])


VIT_VARIANTS = {   # oracle/gen_golden.py: VIT_VARIANTS
    "out_embed_and_class": dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=2,
                                output_tokens=["embed", "class"], num_classes=5),
    "patches_in_patches_out": dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=1,
                                   output_tokens="input", embed_to_patch=True),
    "default_patch_rect_image": dict(image_size=(8, 16), dim=16, depth=1, heads=4, channels=1, n_embed_tokens=1),
    "preprocess_identity": dict(image_size=8, patch_size=4, dim=16, depth=1, preprocess_depth=0, heads=2, mlp_dim=32, channels=2,
                                n_embed_tokens=2, output_tokens="embed", num_classes=3),
    "class_only": dict(image_size=8, patch_size=4, dim=16, depth=2, heads=2, mlp_dim=32, channels=2, n_embed_tokens=1,
                       output_tokens="class", num_classes=3),
    "tokens_in_embed_none": dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=None,
                                 n_input_tokens=3, patch_to_embed=False, output_tokens="embed"),
    "no_embed_tokens": dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, mlp_dim=32, channels=2, n_embed_tokens=0,
                            output_tokens="input"),
    "time_output_without_time": dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, channels=2, output_tokens="time"),
    "bad_patch": dict(image_size=10, patch_size=4, dim=16, depth=1, heads=2, channels=2),
    "bad_output_token": dict(image_size=8, patch_size=4, dim=16, depth=1, heads=2, channels=2, output_tokens="latent"),
}


@pytest.mark.parametrize("name", sorted(VIT_VARIANTS))
def test_vit_constructor_corners_vs_reference_golden(A, name):
    """The ViT at corners of its constructor against the reference's own class (vit_variants.npz): several output token types,
    patches in -> patches out, the default patch size on a rectangular image, `preprocess_depth=0`, the class token as the only
    output, token inputs with `n_embed_tokens=None`, no embed tokens, an empty output selection; state_dict keys and shapes, `out_size`,
    output, input gradient and every parameter gradient; where the reference raises, the same exception type."""
    from detfill import fill_vit_state_dict
    z = load_golden("vit_variants.npz")
    kw = VIT_VARIANTS[name]
    if f"{name}/error" in z.files:
        want = bytes(z[f"{name}/error"].astype("uint8")).decode()
        with pytest.raises(Exception) as info:
            net = A.ViT(dropout=0.0, emb_dropout=0., **kw)
        assert type(info.value).__name__ == want, (type(info.value).__name__, want)
        return
    names = [str(n) for n in z[f"{name}/param_names"]]
    shapes = [tuple(int(d) for d in str(sh).split(";") if d) for sh in z[f"{name}/param_shapes"]]
    g = {k[len(name) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "/") and "param_" not in k}
    net = A.ViT(dropout=0.0, emb_dropout=0., **kw)
    sd = net.state_dict()
    assert list(sd.keys()) == names or set(sd.keys()) == set(names), (sorted(set(sd) ^ set(names)))
    assert all(tuple(sd[k].shape) == sh for k, sh in zip(names, shapes))
    ordered = {k: torch.zeros(sh) for k, sh in zip(names, shapes)}      # the reference's key order decides the fill phases
    fill_vit_state_dict(ordered)
    net.load_state_dict(ordered)
    net = net.cuda().train()
    assert list(net.out_size) == [int(v) for v in g["out_size"]]
    x = g["x"].cuda().requires_grad_(True)
    labels = g["labels"].cuda() if "labels" in g else None
    y = net(x, labels=labels)
    rep = Report(f"ViT corner `{name}` vs reference golden")
    assert tuple(y.shape) == tuple(g["y"].shape)
    if y.numel():
        y.backward(g["gy"].cuda())
        rep.check("output", y, g["y"], 1e-5)
        rep.check("input gradient", x.grad, g["gx"], 2e-4)
        gscale = max(v.abs().max().item() for k, v in g.items() if k.startswith("grad/"))
        for k, p in net.named_parameters():
            got = p.grad if p.grad is not None else torch.zeros_like(p)
            rep.check(f"grad/{k}", got, g[f"grad/{k}"], tol=1e-3, floor=1e-2 * gscale)
    rep.finish()


@pytest.mark.parametrize("role", ["enc", "dec"])
def test_vit_with_causal_mask_vs_reference_golden(A, role):
    """``ViT(causal_mask=True)`` (reference networks/vit.py:215-217,225: every layer's attention restricted to tokens <= t)
    on the d32 configuration: output, input gradient and the L2 norm of every parameter gradient against golden vectors
    of the reference class (``tests/golden/vit_causal.npz``)."""
    from detfill import fill_vit_state_dict
    from test_oracle_vs_golden import VIT_CASES, VIT_ROLES, vit_param_shapes
    g = {k: torch.from_numpy(v) for k, v in load_golden("vit_causal.npz").items()}
    rep = Report(f"ViT d32 {role} with causal mask vs reference golden")
    cfg = VIT_CASES["d32"]
    net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., causal_mask=True, **cfg, **VIT_ROLES[role])
    ordered = {k: torch.zeros(s) for k, s in vit_param_shapes(cfg, role).items()}
    fill_vit_state_dict(ordered)
    net.load_state_dict(ordered)
    net = net.cuda().train()
    x = g[f"{role}/x"].cuda().requires_grad_(True)
    y = net(x, labels=g["labels"].cuda())
    y.backward(g[f"{role}/gy"].cuda())
    rep.check("output", y, g[f"{role}/y"], 1e-5)
    rep.check("input gradient", x.grad, g[f"{role}/gx"], 1e-4)
    rep.check("grad_l2 (all parameters)", torch.tensor([p.grad.double().norm().item() for _, p in net.named_parameters()]),
              g[f"{role}/grad_l2"], 5e-4)
    rep.finish()


@pytest.mark.parametrize("tag", ["enc_p1", "dec_p0_causal"])
def test_vit_cross_attention_variant_vs_reference_golden(A, tag):
    """``ViT(preprocess_depth=...)`` (reference networks/vit.py:171-181,240-244): the output tokens are the target of
    nn.TransformerDecoder layers (self-attention, cross-attention over the other tokens as memory -- pre-processed by encoder layers
    or not -- feed-forward).  d32 configuration: 2 embed tokens over 17 memory tokens after one encoder layer; 16 patch tokens with the
    causal mask over 2 memory tokens.  Output, input gradient and every parameter gradient against the reference class."""
    from detfill import fill_vit_state_dict
    from test_oracle_vs_golden import VIT_CASES, VIT_CROSS_CASES, VIT_ROLES, check_vit_grads, vit_param_shapes
    g = load_golden("vit_cross.npz")
    names = [str(n) for n in g[f"{tag}/param_names"]]
    labels = torch.from_numpy(g["labels"]).cuda()
    g = {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "/") and not k.endswith("param_names")}
    role, pre, causal = VIT_CROSS_CASES[tag]
    cfg = VIT_CASES["d32"]
    rep = Report(f"cross-attention ViT {tag} vs reference golden")
    net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., preprocess_depth=pre, causal_mask=causal, **cfg, **VIT_ROLES[role])
    want = vit_param_shapes(cfg, role, preprocess_depth=pre)
    sd = net.state_dict()
    assert list(want) == names and set(sd.keys()) == set(names) and all(tuple(sd[k].shape) == tuple(s) for k, s in want.items())
    ordered = {k: torch.zeros(s) for k, s in want.items()}          # the reference's key order decides the fill phases
    fill_vit_state_dict(ordered)
    net.load_state_dict(ordered)
    net = net.cuda().train()
    x = g["x"].cuda().requires_grad_(True)
    y = net(x, labels=labels)
    y.backward(g["gy"].cuda())
    rep.check("output", y, g["y"], 1e-5)
    rep.check("input gradient", x.grad, g["gx"], 1e-4)
    rep.finish()
    check_vit_grads(g, "", {k: p.grad for k, p in net.named_parameters()}, 5e-4, sep="")
    assert tuple(net.out_size) == tuple(y.shape[1:])
    # the same network trains through the captured engine: two steps, finite and moving
    from ot_vae_lightning_amd.networks.vit import TokenDecoderLayer
    assert any(isinstance(m, TokenDecoderLayer) for m in net.modules())


def test_autoregressive_vit_vs_reference_golden(A):
    """``AutoRegressive(vocab_size, **vit_kwargs)`` (reference networks/vit.py:249-260): 7 token ids -> vocabulary embedding -> causal
    ViT over the input tokens (class-conditioned) -> logits; output and every parameter gradient against the reference class."""
    from test_oracle_vs_golden import AR_CFG, AR_ROLE, autoregressive_state
    g = load_golden("vit_autoregressive.npz")
    rep = Report("AutoRegressive ViT vs reference golden")
    net = A.AutoRegressive(vocab_size=13, output_tokens="input", causal_mask=True, dropout=0.0, emb_dropout=0., **AR_CFG, **AR_ROLE)
    state = autoregressive_state(g)
    sd = net.state_dict()
    assert set(sd.keys()) == set(state.keys()) and all(tuple(sd[k].shape) == tuple(v.shape) for k, v in state.items())
    net.load_state_dict(state)
    net = net.cuda().train()
    y = net(torch.from_numpy(g["ids"]).cuda(), labels=torch.from_numpy(g["labels"]).cuda())
    y.backward(torch.from_numpy(g["gy"]).cuda())
    rep.check("logits", y, torch.from_numpy(g["y"]), 1e-5)
    for k, p in net.named_parameters():
        rep.check("grad " + k, p.grad, torch.from_numpy(g[f"grad/{k}"]), 5e-4)
    rep.finish()


@pytest.mark.parametrize("n,tq,tk,heads,c,p", [(3, 2, 17, 4, 8, 0.0), (4, 16, 2, 4, 8, 0.0), (2, 40, 100, 2, 16, 0.2), (5, 7, 7, 1, 32, 0.5),
                                                (2, 200, 31, 3, 4, 0.1), (3, 1, 1, 2, 2, 0.0)])
def test_cross_attention_kernels_vs_float64(n, tq, tk, heads, c, p):
    """``otvae_attn_cross_fwd/_bwd`` through ``functional.cross_attention_tokens``: nn.MultiheadAttention(x, memory, memory) up to
    out_proj, queries and keys of different token counts, with and without dropout on the probabilities (the reference is
    handed the kernel's own mask), against float64 torch arithmetic -- output and the gradients of x, memory, weight, bias."""
    import ot_vae_lightning_amd.functional as HF
    d = heads * c
    rep = Report(f"cross attention N={n} Tq={tq} Tk={tk} H={heads} C={c} p={p}")
    x = normal((n, tq, d), 810 + tq).cuda().requires_grad_(True)
    mem = normal((n, tk, d), 811 + tk).cuda().requires_grad_(True)
    w = HF.new_linear_weight(3 * d, d, device="cuda")
    with torch.no_grad():
        w.copy_(normal((3 * d, d), 812).cuda() / d ** 0.5)
    w.requires_grad_(True)
    b = (normal((3 * d,), 813).cuda() * 0.1).requires_grad_(True)
    gout = normal((n, tq, d), 814).cuda()
    key = HF.new_dropout_key(x.device, seed=77) if p > 0 else None
    out = HF.cross_attention_tokens(x, mem, w, b, heads, p, key, stream_id=5)
    gx, gm, gw, gb = torch.autograd.grad(out, (x, mem, w, b), gout)
    keep = HF.attention_cross_mask(HF._CrossAttentionFn.last_used, n, tq, tk, heads, p).double() if p > 0 else 1.0
    if p > 0:
        frac = float(keep.mean())
        assert abs(frac - (1 - p)) < 6 * (p * (1 - p) / keep.numel()) ** 0.5 + 1e-3, (frac, 1 - p)
    xr, mr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, mem, w, b))
    q = torch.nn.functional.linear(xr, wr[:d], br[:d]).reshape(n, tq, heads, c).transpose(1, 2)
    k = torch.nn.functional.linear(mr, wr[d:2 * d], br[d:2 * d]).reshape(n, tk, heads, c).transpose(1, 2)
    v = torch.nn.functional.linear(mr, wr[2 * d:], br[2 * d:]).reshape(n, tk, heads, c).transpose(1, 2)
    prob = torch.softmax(q @ k.transpose(-1, -2) / c ** 0.5, dim=-1)
    ref = ((prob * keep / (1 - p)) @ v).transpose(1, 2).reshape(n, tq, d)
    rx, rm, rw, rb = torch.autograd.grad(ref, (xr, mr, wr, br), gout.double())
    rep.check("out", out, ref.detach(), tol=2e-5)
    rep.check("dx", gx, rx, tol=5e-5)
    rep.check("dmemory", gm, rm, tol=5e-5)
    rep.check("dweight", gw, rw, tol=5e-5)
    rep.check("dbias", gb, rb, tol=5e-5)
    rep.finish()


# ------------------------------------------------------------------------------------------------ G13 conditional prior, ViT VAE
def test_conditional_gaussian_prior_vs_reference_golden(A):
    """ConditionalGaussianPrior (reference prior/conditional_gaussian.py): learned class embeddings (z, loss with cosine
    annealing, gradients of the input and of both embeddings) and the EMA-tracked variant over two training steps."""
    z_ = load_golden("vit_vae.npz")
    g, e = group(z_, "prior"), group(z_, "ema")
    rep = Report("ConditionalGaussianPrior vs reference golden")
    C, n = g["mu_weight"].shape
    prior = A.ConditionalGaussianPrior(dim=(1, n), num_classes=C, loss_coeff=0.3, annealing_steps=10)
    with torch.no_grad():
        prior._mu.weight.copy_(g["mu_weight"])
        prior._log_std.weight.copy_(g["log_std_weight"])
    prior = prior.cuda().train()
    x = g["x"].cuda().requires_grad_(True)
    labels = g["labels"].cuda()
    z, loss, art = prior(x, step=4, labels=labels, eps=g["eps"].cuda())
    ((z * g["w"].cuda()).sum() + loss.sum()).backward()
    rep.check("z", z, g["z"], 1e-6)
    rep.check("loss", loss, g["loss"], 1e-5)
    rep.check("d/dx", x.grad, g["gx"], 1e-5)
    rep.check("d/d mu embedding", prior._mu.weight.grad, g["g_mu"], 1e-5)
    rep.check("d/d log_std embedding", prior._log_std.weight.grad, g["g_log_std"], 1e-5)
    assert art["prior"].mean.shape == z.shape and prior.sample((6, 1, n), "cuda", labels=labels).shape == (6, 1, n)
    ema = A.ConditionalGaussianPrior(dim=(1, n), num_classes=C, loss_coeff=1.0, embedding_ema_decay=0.9)
    with torch.no_grad():
        ema._mu.weight.copy_(e["mu_weight0"])
        ema._log_std.weight.copy_(e["log_std_weight0"])
    ema = ema.cuda().train()
    assert not ema._mu.weight.requires_grad
    for step in range(2):
        z, loss, _ = ema(e[f"step{step}/x"].cuda(), step=0, labels=labels, eps=e[f"step{step}/eps"].cuda())
        rep.check(f"ema step{step}/z", z, e[f"step{step}/z"], 1e-6)
        rep.check(f"ema step{step}/loss", loss, e[f"step{step}/loss"], 1e-5)
        rep.check(f"ema step{step}/size", ema._size, e[f"step{step}/size"], 1e-6)
        rep.check(f"ema step{step}/mu", ema._mu.weight, e[f"step{step}/mu"], 1e-5)
        rep.check(f"ema step{step}/log_std", ema._log_std.weight, e[f"step{step}/log_std"], 1e-5)
    rep.finish()


def test_conditional_vit_vae_nelbo_vs_reference_golden(A):
    """VAE.nelbo of the conditional ViT VAE (reference tests/test_conditional_vit_vae.py:41-85 at dim 32, dropout 0):
    ViT encoder -> ConditionalGaussianPrior -> ViT decoder, class labels to all three, through the reference-named VAE
    module: the three losses, reconstructions, latents, and the gradient norm of every parameter."""
    from detfill import fill_vit_state_dict
    from test_oracle_vs_golden import VIT_ROLES, VIT_VAE_CFG, vit_param_shapes
    z_ = load_golden("vit_vae.npz")
    g = group(z_, "vae")
    rep = Report("conditional ViT VAE nelbo vs reference golden")
    nets = {}
    for role in ("enc", "dec"):
        net = A.ViT(output_tokens="embed", dropout=0.0, emb_dropout=0., **VIT_VAE_CFG, **VIT_ROLES[role])
        ordered = {k: torch.zeros(s) for k, s in vit_param_shapes(VIT_VAE_CFG, role).items()}
        fill_vit_state_dict(ordered)
        net.load_state_dict(ordered)
        nets[role] = net
    prior = A.ConditionalGaussianPrior(dim=(1, 32), num_classes=10, loss_coeff=0.1, empirical_kl=False, reparam_dim=1,
                                       annealing_steps=1000)
    with torch.no_grad():
        prior._mu.weight.copy_(g["mu_weight"])
        prior._log_std.weight.copy_(g["log_std_weight"])
    model = A.VAE(encoder=nets["enc"], decoder=nets["dec"], prior=prior, conditional=True).cuda().train()
    assert tuple(model.latent_size) == (1, 32)
    x = g["x"].cuda()
    batch = {"samples": x, "target": x, "kwargs": {"labels": g["labels"].cuda(), "eps": g["eps"].cuda()}}
    loss, logs, art = model.nelbo(batch, 0)
    loss.backward()
    rep.check("loss [total, recon, prior]", torch.stack([logs["train/loss/total"], logs["train/loss/recon"],
                                                           logs["train/loss/prior"]]), g["loss"], 1e-5)
    rep.check("preds", art["preds"], g["preds"], 1e-5)
    rep.check("latents", art["latents"], g["latents"], 1e-5)
    grads = {}
    for pre, net in (("encoder.", model.encoder), ("decoder.", model.decoder), ("prior.", model.prior)):
        for k, p in net.named_parameters():
            grads[pre + k] = p.grad
    names = [str(n) for n in z_["vae/param_names"]]
    assert set(names) == set(grads)
    l2 = torch.tensor([grads[n].double().norm().item() if grads[n] is not None else 0.0 for n in names])
    rep.check("gradient norms (all parameters)", l2, g["grad_l2"], 5e-4)
    rep.finish()


@pytest.mark.parametrize("n,t,heads,c,p,causal", [(3, 65, 8, 32, 0.1, False), (5, 17, 4, 8, 0.5, False), (2, 100, 2, 16, 0.25, False),
                                                  (7, 9, 3, 4, 0.1, False), (2, 244, 1, 32, 0.1, False), (4, 33, 2, 1, 0.3, False),
                                                  (3, 65, 8, 32, 0.1, True), (2, 50, 2, 16, 0.0, True)])
def test_attention_with_dropout_matches_reference_given_its_own_mask(n, t, heads, c, p, causal):
    """``otvae_attn_dropout_fwd/_bwd`` (nn.MultiheadAttention's dropout on the attention probabilities, reference
    networks/vit.py:157-172 in training mode) against plain torch arithmetic that is handed the kernel's own keep mask
    (``otvae_attn_dropout_mask``): out = (softmax(q k^T / sqrt(C)) o keep / (1-p)) v and all three input gradients."""
    import ot_vae_lightning_amd.functional as HF
    rep = Report(f"attention with dropout N={n} T={t} H={heads} C={c} p={p} causal={causal}")
    qkv = normal((n, t, 3 * heads * c), 700 + t).cuda().requires_grad_(True)
    gout = normal((n, t, heads * c), 701 + t).cuda()
    key = HF.new_dropout_key(qkv.device, seed=1234)
    out, used = HF.mha_attention_tokens(qkv, heads, p, key, stream_id=3, return_used=True, causal=causal)
    (gq,) = torch.autograd.grad(out, qkv, gout)
    keep = HF.attention_dropout_mask(used, n, t, heads, p)                      # [N, H, T, T]
    frac = keep.float().mean().item()
    sigma = (p * (1 - p) / keep.numel()) ** 0.5
    assert abs(frac - (1 - p)) < 6 * sigma + 1e-3, (frac, 1 - p)                # Bernoulli(1-p), not merely "some mask"
    assert abs(keep.float().mean(dim=(0, 1, 2)) - (1 - p)).max().item() < 0.2  # no key column is favoured
    ref_in = qkv.detach().double().requires_grad_(True)
    q, k, v = (x.reshape(n, t, heads, c).transpose(1, 2) for x in ref_in.chunk(3, dim=-1))      # [N, H, T, C]
    scores = q @ k.transpose(-1, -2) / c ** 0.5
    if causal:                                                                   # the ViT's `causal_mask` (networks/vit.py:215-217)
        scores = scores + torch.nn.Transformer.generate_square_subsequent_mask(t).to(scores)
    prob = torch.softmax(scores, dim=-1)
    ref = ((prob * keep.double() / (1 - p)) @ v).transpose(1, 2).reshape(n, t, heads * c)
    (gref,) = torch.autograd.grad(ref, ref_in, gout.double())
    rep.check("out", out, ref.detach(), tol=2e-5)
    rep.check("dq", gq[..., :heads * c], gref[..., :heads * c], tol=5e-5)
    rep.check("dk", gq[..., heads * c:2 * heads * c], gref[..., heads * c:2 * heads * c], tol=5e-5)
    rep.check("dv", gq[..., 2 * heads * c:], gref[..., 2 * heads * c:], tol=5e-5)
    # same key, same call site: the same mask; another call site or a later counter: another mask
    out2, used2 = HF.mha_attention_tokens(qkv, heads, p, key, stream_id=3, return_used=True, causal=causal)
    assert torch.equal(out2, out) and torch.equal(used2, used)
    if p == 0:
        rep.finish()
        return
    other_site = HF.mha_attention_tokens(qkv, heads, p, key, stream_id=4, return_used=True)[1]
    key[1:].add_(1)
    later = HF.mha_attention_tokens(qkv, heads, p, key, stream_id=3, return_used=True)[1]
    masks = [HF.attention_dropout_mask(u, n, t, heads, p) for u in (used, other_site, later)]
    for a in range(3):
        for b in range(a + 1, 3):
            agree = (masks[a] == masks[b]).float().mean().item()
            assert abs(agree - (p * p + (1 - p) ** 2)) < 0.05, (a, b, agree)    # independent masks agree by chance only
    rep.finish()


def test_attention_dropout_zero_probability_is_the_plain_kernel_and_bad_arguments_are_refused():
    import ot_vae_lightning_amd.functional as HF
    from ot_vae_lightning_amd import _lib as L
    qkv = normal((3, 20, 3 * 2 * 8), 711).cuda()
    key = HF.new_dropout_key(qkv.device, seed=7)
    lib = L.load()
    out = torch.empty(3, 20, 16, device="cuda")
    lse = torch.empty(3, 2, 20, device="cuda")
    used = torch.empty(1, dtype=torch.int64, device="cuda")
    rc = lib.otvae_attn_dropout_fwd(L.ptr(qkv), 3, 20, 2, 8, 8 ** -0.5, 0.0, 0, L.ptr(key), 0, L.ptr(out), L.ptr(lse), L.ptr(used), L.stream())
    assert rc == 0
    plain = HF.mha_attention_tokens(qkv, 2)
    assert (out - plain).abs().max().item() < 1e-6
    assert HF.attention_dropout_mask(used, 3, 20, 2, 0.0).all()
    for bad in (dict(p=1.0), dict(p=-0.1), dict(t=300), dict(c=5)):
        t, c, p = bad.get("t", 20), bad.get("c", 8), bad.get("p", 0.1)
        rc = lib.otvae_attn_dropout_fwd(L.ptr(qkv), 1, t, 1, c, 1.0, p, 0, L.ptr(key), 0, L.ptr(out), L.ptr(lse), L.ptr(used), L.stream())
        assert rc != 0, bad
    with pytest.raises(ValueError):
        HF.mha_attention_tokens(qkv, 2, 0.1)


@pytest.mark.parametrize("m,d,p", [(1235, 256, 0.1), (77, 32, 0.5), (300, 100, 0.25), (64, 1000, 0.1)])
def test_layernorm_with_fused_dropout_matches_reference_given_its_own_mask(m, d, p):
    """``otvae_layernorm_dropout_fwd/_bwd``: y = LayerNorm(res + dropout(x)), the post-norm block of a training-mode
    nn.TransformerEncoderLayer (reference networks/vit.py:157-172), against torch arithmetic handed the kernel's mask."""
    import ot_vae_lightning_amd.functional as HF
    rep = Report(f"LayerNorm(res + dropout(x)) M={m} D={d} p={p}")
    x = normal((m, d), 720 + d).cuda().requires_grad_(True)
    res = normal((m, d), 721 + d).cuda().requires_grad_(True)
    gamma = (1 + 0.1 * normal((d,), 722)).cuda().requires_grad_(True)
    beta = (0.1 * normal((d,), 723)).cuda().requires_grad_(True)
    gy = normal((m, d), 724 + d).cuda()
    key = HF.new_dropout_key(x.device, seed=99)
    y, used = HF.layer_norm_tokens(x, gamma, beta, 1e-5, res, p, key, stream_id=1027, return_used=True)
    grads = torch.autograd.grad(y, (x, res, gamma, beta), gy)
    keep = HF.layer_norm_dropout_mask(used, m, d, p)
    frac = keep.float().mean().item()
    assert abs(frac - (1 - p)) < 6 * (p * (1 - p) / keep.numel()) ** 0.5 + 1e-3, (frac, 1 - p)
    ref_in = [t.detach().double().requires_grad_(True) for t in (x, res, gamma, beta)]
    s = ref_in[1] + ref_in[0] * keep.double() / (1 - p)
    ref = torch.nn.functional.layer_norm(s, (d,), ref_in[2], ref_in[3], 1e-5)
    gref = torch.autograd.grad(ref, ref_in, gy.double())
    rep.check("y", y, ref.detach(), tol=2e-5)
    for name, got, want in zip(("dx (thinned operand)", "dres", "dgamma", "dbeta"), grads, gref):
        rep.check(name, got, want, tol=5e-5)
    # a different call site draws an independent mask; without dropout the plain kernel is used
    other = HF.layer_norm_tokens(x, gamma, beta, 1e-5, res, p, key, stream_id=1028, return_used=True)[1]
    agree = (HF.layer_norm_dropout_mask(other, m, d, p) == keep).float().mean().item()
    assert abs(agree - (p * p + (1 - p) ** 2)) < 0.05
    with pytest.raises(ValueError):
        HF.layer_norm_tokens(x, gamma, beta, 1e-5, None, p, key)
    rep.finish()


@pytest.mark.parametrize("relu", [True, False])
def test_elementwise_dropout_kernels_match_reference_given_their_own_mask(relu):
    """``otvae_dropout_fwd/_bwd``: dropout(relu(x)) of the feed-forward half / the embedding dropout (reference
    networks/vit.py:54-58,157-172 in training mode); bit-exact against torch arithmetic with the kernel's mask."""
    import ot_vae_lightning_amd.functional as HF
    m, d, p = 999, 512, 0.1
    x = normal((3, 333, d), 740).cuda().requires_grad_(True)
    gy = normal((3, 333, d), 741).cuda()
    key = HF.new_dropout_key(x.device, seed=5)
    y, used = HF.dropout_tokens(x, p, key, stream_id=2049, relu=relu, return_used=True)
    (gx,) = torch.autograd.grad(y, x, gy)
    keep = HF.layer_norm_dropout_mask(used, m, d, p).reshape(3, 333, d)
    assert abs(keep.float().mean().item() - (1 - p)) < 5e-3
    inv = torch.tensor(1.0, device="cuda") / (1.0 - torch.tensor(p, device="cuda"))      # the kernel's fp32 1/(1-p)
    act = torch.relu(x.detach()) if relu else x.detach()
    gate = keep & (x.detach() > 0) if relu else keep
    assert torch.equal(y, torch.where(keep, act * inv, torch.zeros_like(act)))
    assert torch.equal(gx, torch.where(gate, gy * inv, torch.zeros_like(gy)))
    with pytest.raises(ValueError):
        HF.dropout_tokens(normal((4, 6), 1).cuda(), p, key)                               # D % 4 != 0


# ---- the reference's own statistical tests (tests/test_distribution_models.py), same sizes and tolerance -------------------
_REF_LEAD, _REF_DIM, _REF_NCOMP, _REF_SAMPLES, _REF_TOL = (2,), 32, 16, 10000, 1e-1


def _ref_rand_mean_cov(*shape, diag, seed):
    """mean ~ N(0, I); cov = G G^T / dim + 1e-5 I (its diagonal when ``diag``): test_distribution_models.py:31-37"""
    gen = torch.Generator().manual_seed(seed)
    mean = torch.randn(*shape, generator=gen, dtype=torch.double)
    g = torch.randn(*shape, shape[-1], generator=gen, dtype=torch.double)
    cov = g @ g.transpose(-1, -2) / _REF_DIM + 1e-5 * torch.eye(shape[-1], dtype=torch.double)
    return mean, (torch.diagonal(cov, dim1=-1, dim2=-2).clone() if diag else cov)


@pytest.mark.parametrize("diag", [True, False])
def test_gaussian_model_recovers_the_sampling_distribution(A, diag):
    """reference tests/test_distribution_models.py:171-174 (``fit`` and ``update`` modes; its third mode,
    ``update_with_autograd``, is outside the hot path and must say so): 1e4 samples of a known Gaussian per operator,
    leading shape (2,), dim 32, double -- the fitted model must be within W2 < 0.1 of the truth."""
    import torch.distributions as D
    torch.manual_seed(100 + int(diag))
    size = (*_REF_LEAD, _REF_DIM)
    mean, cov = _ref_rand_mean_cov(*size, diag=diag, seed=7 + int(diag))
    truth = D.Independent(D.Normal(mean, cov ** 0.5), 1) if diag else D.MultivariateNormal(mean, cov)
    samples = truth.sample((_REF_SAMPLES,)).permute(1, 0, 2).contiguous().cuda()           # [2, 1e4, 32]
    truth_gpu = D.Independent(D.Normal(mean.cuda(), cov.cuda() ** 0.5), 1) if diag else D.MultivariateNormal(mean.cuda(), cov.cuda())
    kwargs = dict(w2_cfg={"diag": diag}, dtype=torch.double)
    fitted = A.GaussianModel(*size, **kwargs).cuda()
    fitted.fit(samples)
    assert float(fitted.w2(truth_gpu).max()) < _REF_TOL
    streamed = A.GaussianModel(*size, **kwargs, update_decay=None).cuda()
    for batch in samples[:, torch.randperm(_REF_SAMPLES, device="cuda")].split(100, dim=-2):     # shuffled batches of 100
        streamed.update(batch)
    streamed.fit()
    assert float(streamed.w2(truth_gpu).max()) < _REF_TOL
    assert torch.allclose(streamed.mean, fitted.mean, atol=1e-9) and torch.allclose(streamed.cov, fitted.cov, atol=1e-9)
    # the third mode of the reference's test, 'autograd' (:150-168), has its own test: test_gaussian_model_update_with_autograd
    auto = A.GaussianModel(*size, **kwargs, update_with_autograd=True).cuda()
    assert auto.mean.requires_grad and not hasattr(auto, "_running_sum")
    with pytest.raises(RuntimeError):
        auto.update(samples[..., :100, :])


def test_gaussian_mixture_model_on_the_references_recovery_experiment(A):
    """reference tests/test_distribution_models.py:177-181 (diagonal case, 'argmax', fit and update): 1e4 samples of a
    16-component mixture per operator.  The reference's test of this cannot run (it names an undefined variable, :180) and
    its W2 < 0.1 bound is far from what its class reaches (W2 between 1 and 15: k-means from a random start in 32
    dimensions), so the bar here is the reference class itself: the same seeds, the same W2 (``tests/golden/
    gmm_recovery.npz``, recorded by oracle/gen_golden.py::gen_gmm_recovery), through the HIP assignment / k-means /
    Sinkhorn kernels."""
    from detfill import gmm_recovery_inputs
    g = load_golden("gmm_recovery.npz")
    rep = Report("GaussianMixtureModel on the reference's recovery experiment vs the reference class")
    (lead, k, dim, n), mean, var, truth, samples, order = gmm_recovery_inputs()
    check = torch.tensor([samples.sum().item(), samples.square().sum().item(), float(order[:16].sum())], dtype=torch.double)
    rep.check("inputs regenerated from the seeds", check, torch.from_numpy(g["samples_checksum"]), 1e-12)
    import torch.distributions as D
    truth_gpu = D.MixtureSameFamily(D.Categorical(probs=truth.mixture_distribution.probs.cuda()),
                                    D.Independent(D.Normal(mean.cuda(), var.cuda() ** 0.5), 1))
    cfg = dict(w2_cfg={"diag": True}, dtype=torch.double, mixture_cfg={"n_components": k, "training_mode": "argmax", "topk": None})
    torch.manual_seed(103)
    fitted = A.GaussianMixtureModel(*lead, dim, **cfg).cuda().train()
    fitted.fit(samples.cuda())
    rep.check("fit: component means", fitted.mean, torch.from_numpy(g["fit_mean"]), 1e-9)
    rep.check("fit: W2 to the sampling mixture", fitted.w2(truth_gpu), torch.from_numpy(g["w2_fit"]), 1e-6)
    torch.manual_seed(105)
    streamed = A.GaussianMixtureModel(*lead, dim, **cfg, update_decay=None).cuda().train()
    for batch in samples[:, order].cuda().split(100, dim=-2):
        streamed.update(batch)
    streamed.fit()
    rep.check("update: component means", streamed.mean, torch.from_numpy(g["update_mean"]), 1e-9)
    rep.check("update: W2 to the sampling mixture", streamed.w2(truth_gpu), torch.from_numpy(g["w2_update"]), 1e-6)
    with pytest.raises(NotImplementedError):   # an assignment mode the reference does not have either (base.py:236-237)
        A.GaussianMixtureModel(*lead, dim, **{**cfg, "mixture_cfg": {**cfg["mixture_cfg"], "training_mode": "nearest"}})
    rep.finish()
