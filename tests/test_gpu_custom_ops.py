"""GPU: the registered custom operators (``torch.ops.otvae.*``, ot_vae_lightning_amd/ops.py) -- ``torch.library.opcheck``
(schema, fake tensors, autograd registration, AOT dispatch) on real inputs, and parity: the operator route gives the bits of
the module route, its gradients match torch's own autograd of the reference arithmetic."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from detfill import normal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    assert torch.cuda.is_available()
    import ot_vae_lightning_amd as pkg
    return pkg


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def test_opcheck_on_every_differentiable_op(A):
    from torch.library import opcheck
    from ot_vae_lightning_amd import functional as HF
    utils = ("test_schema", "test_autograd_registration", "test_faketensor")
    x = _nhwc(normal((4, 8, 8, 8), 1)).cuda().requires_grad_(True)
    w = HF.hwio_weight(normal((16, 8, 3, 3), 2)).cuda().requires_grad_(True)
    g, b = normal((8,), 3).cuda().requires_grad_(True), normal((8,), 4).cuda().requires_grad_(True)
    with torch.no_grad():
        st = torch.ops.otvae.bn_batch_stats(x, g, b, None, None, None, True)
    opcheck(torch.ops.otvae.conv_bn_act, (x, w, None, g, b, *st, None, 1, 1, 1, True, True), test_utils=utils)
    qkv = _nhwc(normal((2, 24, 8, 8), 5)).cuda().requires_grad_(True)
    opcheck(torch.ops.otvae.qkv_attention, (qkv, 4, 0.5, True), test_utils=utils)
    h = _nhwc(normal((6, 32, 1, 1), 6)).cuda().requires_grad_(True)
    opcheck(torch.ops.otvae.gaussian_prior, (h, normal((6, 16, 1, 1), 7).cuda(), 0.1), test_utils=utils)
    pred = normal((6, 1, 8, 8), 8).cuda().requires_grad_(True)
    opcheck(torch.ops.otvae.nelbo_loss, (pred, normal((6, 1, 8, 8), 9).cuda(), normal((6,), 10).cuda().requires_grad_(True), 64.0),
            test_utils=utils)
    z = normal((64, 16), 11).cuda().requires_grad_(True)
    opcheck(torch.ops.otvae.sinkhorn_prior, (z, normal((64, 16), 12).cuda(), 0.05, 20, 0.0, 1.0), test_utils=utils)
    opcheck(torch.ops.otvae.gaussian_w2_prior, (z, None, None, None, None, None, 1.0), test_utils=utils)
    basis = torch.linalg.qr(normal((16, 16), 13).double())[0].unsqueeze(0).cuda()   # any orthonormal start basis
    opcheck(torch.ops.otvae.gaussian_w2_prior, (z, None, None, None, basis, torch.ones(1, dtype=torch.int32, device="cuda"), 1.0),
            test_utils=utils)


def test_conv_bn_act_operator_equals_the_module_route_and_torch_autograd(A):
    """``functional.conv_bn_act`` (the two registered operators) against (i) the ConvLayer module's packed route: identical bits,
    running statistics included; (ii) torch's own autograd of batch_norm -> relu -> upsample -> conv2d on the GPU: 1e-4."""
    from ot_vae_lightning_amd import functional as HF
    for (cin, cout, hw, up, stride, k) in ((8, 16, 8, 1, 1, 3), (16, 8, 4, 2, 1, 3), (8, 8, 16, 1, 2, 4)):
        torch.manual_seed(0)
        layer = A.ConvLayer(cin, cout, up_sample=up if up > 1 else None, down_sample=2 if stride == 2 else None,
                            normalization="batchnorm", activation="relu").cuda().train()
        x = _nhwc(normal((6, cin, hw, hw), 20 + cin)).cuda()
        gy = None
        outs = []
        for route in ("module", "operator"):
            for p in layer.parameters():
                p.grad = None
            layer._normalization.running_mean.zero_(); layer._normalization.running_var.fill_(1.0)
            xi = x.clone().requires_grad_(True)
            if route == "module":
                y = layer(xi)
            else:
                bn = layer._normalization
                br = layer.branch()
                y = HF.conv_bn_act(xi, layer.weight, layer.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, None, br["stride"], br["pad"], br["up"], True, True)
            gy = torch.sin(torch.arange(y.numel(), device="cuda", dtype=torch.float32)).reshape(y.shape) if gy is None else gy
            y.backward(gy)
            outs.append([y.detach().clone(), xi.grad.clone(), layer.weight.grad.clone(), layer.bias.grad.clone(),
                         layer._normalization.weight.grad.clone(), layer._normalization.bias.grad.clone(),
                         layer._normalization.running_mean.clone(), layer._normalization.running_var.clone()])
        for a, b in zip(*outs):
            assert torch.equal(a, b)
        # torch's autograd of the same arithmetic
        bn = layer._normalization
        br = layer.branch()
        xt = x.clone().contiguous().requires_grad_(True)
        wt, bt = layer.weight.detach().clone().contiguous().requires_grad_(True), layer.bias.detach().clone().requires_grad_(True)
        gt, bet = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
        hcur = F.relu(F.batch_norm(xt, None, None, gt, bet, True, 0.1, 1e-5))
        if br["up"] > 1:
            hcur = F.interpolate(hcur, scale_factor=br["up"], mode="nearest")
        yt = F.conv2d(hcur, wt, bt, stride=br["stride"], padding=br["pad"])
        yt.backward(gy.contiguous())
        y, gx, gw, gb, dg, db = outs[1][:6]
        for name, got, want in (("y", y, yt), ("gx", gx, xt.grad), ("gw", gw, wt.grad), ("gb", gb, bt.grad), ("dgamma", dg, gt.grad),
                                ("dbeta", db, bet.grad)):
            assert rel_err(got, want.detach()) < 1e-4, (cin, cout, name, rel_err(got, want.detach()))


def test_operator_gradients_reach_an_ordinary_training_step(A):
    """An unmodified ``loss.backward()`` through modules whose forward goes through the operators (attention, Gaussian prior,
    nelbo reduction): the autograd graph holds the operators' registered backward nodes."""
    torch.manual_seed(1)
    enc = A.CNN(1, 32, 16, 1, capacity=4, down_sample=True, residual="add")
    dec = A.CNN(16, 1, 1, 16, capacity=4, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).cuda().train()
    x = normal((8, 1, 16, 16), 31).cuda()
    loss, logs, art = model.nelbo({"samples": x, "target": x, "kwargs": {"eps": normal((8, 16, 1, 1), 32).cuda()}}, 0)
    names = set()
    stack, seen = [loss.grad_fn], set()
    while stack:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        names.add(type(fn).__name__)
        stack += [nf for nf, _ in fn.next_functions]
    assert any("nelbo_loss" in n for n in names) and any("qkv_attention" in n for n in names) and any("gaussian_prior" in n for n in names), names
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.optim_parameters())
