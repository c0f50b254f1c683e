"""CPU: the ``torch.library`` registration of the fused operators (ot_vae_lightning_amd/ops.py) -- every op is defined in the
``otvae`` namespace with the documented schema, has a fake (meta) implementation whose output shapes / strides match the
kernels' contract, an autograd registration, and NO CPU kernel (the product has no CPU path)."""
import pytest
import torch

import ot_vae_lightning_amd  # noqa: F401
from ot_vae_lightning_amd import ops


def test_every_fused_op_is_registered_with_a_backward_op():
    for name in ops.OPS:
        op = getattr(torch.ops.otvae, name).default
        assert op._schema.name == f"otvae::{name}"
        if name != "bn_batch_stats":
            assert hasattr(torch.ops.otvae, name + "_backward"), name
            assert torch._C._dispatch_has_kernel_for_dispatch_key(f"otvae::{name}", "Autograd"), name
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"otvae::{name}", "CUDA"), name
        assert not torch._C._dispatch_has_kernel_for_dispatch_key(f"otvae::{name}", "CPU"), name
    assert "running_mean" in str(torch.ops.otvae.bn_batch_stats.default._schema) and "(a!)" in str(
        torch.ops.otvae.bn_batch_stats.default._schema)


def test_cpu_tensors_are_refused():
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.otvae.qkv_attention(torch.zeros(2, 6, 4, 4), 1, 0.5, False)


def test_fake_implementations_give_the_kernels_shapes():
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x = torch.empty((4, 16, 16, 8), device="cuda").permute(0, 3, 1, 2)                      # logical NCHW on NHWC memory
        w = torch.empty((4, 4, 8, 16), device="cuda").permute(3, 2, 0, 1)                        # logical OIHW on HWIO memory
        g = torch.empty(8, device="cuda")
        mean, invstd, scale, shift = torch.ops.otvae.bn_batch_stats(x, g, g, None, None, None, True)
        assert mean.shape == (8,) and shift.shape == (8,)
        y = torch.ops.otvae.conv_bn_act(x, w, None, g, g, mean, invstd, scale, shift, None, 2, 1, 1, True, True)
        assert y.shape == (4, 16, 8, 8) and y.stride() == (8 * 8 * 16, 1, 8 * 16, 16)           # NHWC memory
        y = torch.ops.otvae.conv_bn_act(x, torch.empty((3, 3, 8, 8), device="cuda").permute(3, 2, 0, 1), None, None, None, None, None,
                                        None, None, None, 1, 1, 2, True, True)
        assert y.shape == (4, 8, 32, 32)
        gx, gw, gb, dg, db = torch.ops.otvae.conv_bn_act_backward(y, x, torch.empty((3, 3, 8, 8), device="cuda").permute(3, 2, 0, 1),
                                                                  None, None, None, None, None, 1, 1, 2, True, False, False, True, True)
        assert gx.shape == x.shape and gw.shape == (8, 8, 3, 3) and gb.numel() == 0 and dg.numel() == 0
        qkv = torch.empty((2, 32, 32, 3), device="cuda").permute(0, 3, 1, 2)
        out, lse, aux = torch.ops.otvae.qkv_attention(qkv, 1, 1.0, True)
        assert out.shape == (2, 1, 32, 32) and lse.shape == (2, 1, 1024) and aux.shape == (2, 1, 1024, 1)
        assert torch.ops.otvae.qkv_attention_backward(out, qkv, out, lse, aux, 1, 1.0).shape == qkv.shape
        h = torch.empty((5, 1, 1, 256), device="cuda").permute(0, 3, 1, 2)
        z, loss = torch.ops.otvae.gaussian_prior(h, torch.empty((5, 128, 1, 1), device="cuda"), 0.1)
        assert z.shape == (5, 128, 1, 1) and loss.shape == (5,)
        assert torch.ops.otvae.nelbo_loss(torch.empty(5, 1, 32, 32, device="cuda"), torch.empty(5, 1, 32, 32, device="cuda"), loss,
                                          1024.0).shape == (3,)
        zz, yy = torch.empty(64, 16, device="cuda"), torch.empty(48, 16, device="cuda")
        cost, pi, iters = torch.ops.otvae.sinkhorn_prior(zz, yy, 0.05, 50, 0.0, 1.0)
        assert cost.shape == (64,) and pi.shape == (64, 48) and iters.dtype == torch.int32
        assert torch.ops.otvae.sinkhorn_prior_backward(cost, None, zz, yy, pi, 1.0).shape == zz.shape
        loss, mu, q, vt = torch.ops.otvae.gaussian_w2_prior(zz, None, None, None, None, None, 1.0)
        assert loss.shape == (64,) and mu.shape == (1, 16) and q.shape == (16, 16) and q.dtype == torch.float64 and vt.shape == (1, 16, 16)


def test_deferred_side_work_bookkeeping():
    """``_PendingReduce.defer_to_side``: work is accepted only while a training engine has opened the window for the device, runs once
    (in line when no fork took it), and an interrupted pass leaves nothing behind."""
    from ot_vae_lightning_amd.functional import _PendingReduce as PR
    dev = torch.device("cpu")
    ran = []
    assert PR.defer_to_side(dev, lambda: ran.append("no")) is False and not PR._side_prologue.get(dev)
    PR._defer[dev] = True
    t = torch.zeros(3)
    try:
        assert PR.defer_to_side(dev, lambda: ran.append("a"), t, None) is True
        assert PR.defer_to_side(dev, lambda: ran.append("b")) is True
    finally:
        PR._defer[dev] = False
    assert ran == [] and any(h is t for h in PR._held[dev])
    PR.run_deferred_inline(dev)
    assert ran == ["a", "b"]
    PR.run_deferred_inline(dev)  # nothing is left
    assert ran == ["a", "b"]
    PR._defer[dev] = True
    PR.defer_to_side(dev, lambda: ran.append("c"))
    PR._defer[dev] = False
    PR._held[dev].clear()
    PR._side_prologue.pop(dev, None)
    assert ran == ["a", "b"]


def test_attention_stage_declines_what_it_cannot_fuse_without_touching_the_device():
    """``functional.attention_stage`` answers None (the caller issues three launches) for CPU tensors, non-fp32 inputs, head counts that do
    not divide the width and for 1x1 layers that are not plain, before it loads the HIP library."""
    from ot_vae_lightning_amd import functional as HF
    w_q, w_p = HF.new_hwio(24, 8, 1, 1), HF.new_hwio(8, 8, 1, 1)
    plain = dict(stride=1, pad=0, up=1, relu=False, act=0, wscale=1.0, bscale=1.0)
    qb, pb = dict(weight=w_q, **plain), dict(weight=w_p, **plain)
    x = torch.zeros(2, 8, 4, 4)
    assert HF.attention_stage(x, qb, 4, pb) is None                      # CPU tensor
    assert HF.attention_stage(x.double(), qb, 4, pb) is None             # not fp32
    assert HF._plain_1x1(qb, 24, 8) and HF._plain_1x1(pb, 8, 8)
    assert not HF._plain_1x1(dict(qb, bias=torch.zeros(24)), 24, 8)      # bias
    assert not HF._plain_1x1(dict(qb, relu=True), 24, 8)                 # activation in front
    assert not HF._plain_1x1(dict(qb, wscale=0.5), 24, 8)                # equalized learning rate
    assert not HF._plain_1x1(dict(qb, film=(x, x)), 24, 8)               # FiLM conditioning
    assert not HF._plain_1x1(dict(qb, stride=2), 24, 8)
    assert not HF._plain_1x1(qb, 16, 8)                                  # not 3 x width outputs
