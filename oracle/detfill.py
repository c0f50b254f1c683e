"""
TEST INFRASTRUCTURE.  Deterministic, closed-form weight / input fills shared by the golden generator
(applied to the real reference's modules), the oracle tests and the GPU parity tests (applied to the product
modules), so that large configurations need no weight files: every value is a function of its flat index
and of the tensor's position in ``state_dict`` order.
"""
import math

import torch


def fill_tensor(t: torch.Tensor, j: int, amp: float, offset: float = 0.0) -> None:
    i = torch.arange(t.numel(), dtype=torch.float64)
    v = offset + amp * torch.sin(0.37 * i + 1.0 + 0.61 * j)
    with torch.no_grad():
        t.copy_(v.reshape(t.shape).to(t.dtype))


def fill_state_dict(sd) -> None:
    """In-place fill of a CNN ``state_dict`` (or a plain dict of tensors with the same keys)."""
    for j, (k, t) in enumerate(sd.items()):
        if k.endswith("num_batches_tracked") or k.endswith("running_mean") or k.endswith("running_var"):
            continue
        if k.endswith("_normalization.weight"):
            fill_tensor(t, j, 0.2, 1.0)
        elif k.endswith("_normalization.bias"):
            fill_tensor(t, j, 0.1)
        elif k.endswith("bias"):
            fill_tensor(t, j, 0.05)
        elif k.endswith("weight"):
            fan_in = t[0].numel()
            fill_tensor(t, j, 1.7 / math.sqrt(fan_in))
        else:
            raise KeyError(k)


def fill_vit_state_dict(sd) -> None:
    """In-place fill of a ViT ``state_dict``: Linear / in_proj weights ~ 1.7 / sqrt(fan_in), LayerNorm weights around 1,
    biases small, embeddings and the learned tokens of amplitude 0.5."""
    for j, (k, t) in enumerate(sd.items()):
        if k.endswith("bias"):
            fill_tensor(t, j, 0.05)
        elif t.dim() == 1:
            fill_tensor(t, j, 0.2, 1.0)
        elif k.endswith("embed_token") or "position_embeddings" in k or k.startswith("class_token"):
            fill_tensor(t, j, 0.5)
        else:
            fill_tensor(t, j, 1.7 / math.sqrt(t.shape[-1]))


def det_input(shape, phase: float = 0.0, amp: float = 1.0, dtype=torch.float32) -> torch.Tensor:
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.float64)
    return (amp * torch.sin(0.11 * i + phase) + 0.3 * amp * torch.cos(0.0137 * i * i + phase)).reshape(shape).to(dtype)


def mnist_like(batch: int, seed: int = 42) -> torch.Tensor:
    """MNIST-like synthetic batch [B,1,32,32]: ~19% ink, normalised, zero border of 2."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(batch, 1, 28, 28, generator=g)
    m = torch.rand(batch, 1, 28, 28, generator=g)
    x = torch.where(m < 0.81, torch.zeros_like(u), u)
    x = (x - 0.1307) / 0.3081
    return torch.nn.functional.pad(x, (2, 2, 2, 2))


def normal(shape, seed: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=dtype)


def gmm_recovery_inputs():
    """the sampling mixture and samples of the reference's tests/test_distribution_models.py:177-181 experiment (leading
    shape (2,), 16 components, dim 32, 1e4 samples, diagonal covariances), from fixed seeds; shared with the GPU test"""
    import torch.distributions as D
    lead, k, dim, n = (2,), 16, 32, 10000
    gen = torch.Generator().manual_seed(9)
    mean = torch.randn(*lead, k, dim, generator=gen, dtype=torch.double)
    g = torch.randn(*lead, k, dim, dim, generator=gen, dtype=torch.double)
    var = torch.diagonal(g @ g.transpose(-1, -2) / dim + 1e-5 * torch.eye(dim, dtype=torch.double), dim1=-1, dim2=-2).clone()
    weights = torch.ones(*lead, k, dtype=torch.double) / k
    truth = D.MixtureSameFamily(D.Categorical(probs=weights), D.Independent(D.Normal(mean, var ** 0.5), 1))
    torch.manual_seed(102)
    samples = truth.sample((n,)).permute(1, 0, 2).contiguous()
    torch.manual_seed(104)
    order = torch.randperm(n)
    return (lead, k, dim, n), mean, var, truth, samples, order


def edge_calls(w2, nu, dev="cpu"):
    """(name -> thunk) of edge-case calls into the OT helpers and QKVAttention; shared by the generator (reference modules, CPU) and --
    with the package's own modules and dev='cuda' -- by tests/test_gpu_parity.py"""
    g = torch.Generator().manual_seed(501)
    t = lambda *s, **k: torch.randn(*s, generator=g, dtype=torch.double, **k)  # noqa: E731
    ms, mt = t(2, 3, 4), t(2, 5, 4)
    vs, vt = torch.rand(2, 3, 4, generator=g, dtype=torch.double) + 0.3, torch.rand(2, 5, 4, generator=g, dtype=torch.double) + 0.3
    ws = torch.tensor([[0.2, 0.3, 0.5], [0.6, 0.3, 0.1]], dtype=torch.double)
    wt = torch.full((2, 5), 0.2, dtype=torch.double)
    a, b = torch.full((4,), 0.25, dtype=torch.double), torch.full((6,), 1 / 6, dtype=torch.double)
    C = torch.rand(4, 6, generator=g, dtype=torch.double)
    T5 = t(5, 5)
    x5, m5 = t(7, 5), t(5)
    qkv = torch.randn(2, 3 * 4 * 2, 9, generator=g)
    D = lambda v: v.to(dev)  # noqa: E731
    return {
        "gmm_ok_weights": lambda: w2.batch_ot_gmm(D(ms), D(mt), D(vs), D(vt), diag=True, weight_source=D(ws), weight_target=D(wt), max_iter=50),
        "gmm_default_weights": lambda: w2.batch_ot_gmm(D(ms), D(mt), D(vs), D(vt), diag=True, max_iter=50),
        "gmm_weights_not_normalised": lambda: w2.batch_ot_gmm(D(ms), D(mt), D(vs), D(vt), diag=True, weight_source=D(ws * 2), weight_target=D(wt)),
        "gmm_negative_weight": lambda: w2.batch_ot_gmm(D(ms), D(mt), D(vs), D(vt), diag=True, weight_source=D(torch.tensor([[1.2, -0.2, 0.0], [0.6, 0.3, 0.1]], dtype=torch.double)), weight_target=D(wt)),
        "gmm_negative_variance": lambda: w2.batch_ot_gmm(D(ms), D(mt), D(-vs), D(vt), diag=True),
        "gmm_dim_mismatch": lambda: w2.batch_ot_gmm(D(ms), D(mt[..., :3]), D(vs), D(vt[..., :3]), diag=True),
        "sinkhorn_one_iteration": lambda: w2.sinkhorn_log(D(a), D(b), D(C), reg=0.1, max_iter=1),
        "sinkhorn_zero_iterations": lambda: w2.sinkhorn_log(D(a), D(b), D(C), reg=0.1, max_iter=0),
        "sinkhorn_huge_threshold": lambda: w2.sinkhorn_log(D(a), D(b), D(C), reg=0.1, max_iter=100, threshold=1e9),
        "sinkhorn_float32": lambda: w2.sinkhorn_log(D(a.float()), D(b.float()), D(C.float()), reg=0.05, max_iter=30),
        "apply_dim_mismatch": lambda: w2.apply_transport(D(x5[:, :4]), D(m5), D(m5), D(T5), D(torch.zeros(5, 5, dtype=torch.double))),
        "apply_zero_noise": lambda: w2.apply_transport(D(x5), D(m5), D(m5 + 1), D(T5), D(torch.zeros(5, 5, dtype=torch.double))),
        "apply_diag": lambda: w2.apply_transport(D(x5), D(m5), D(m5 + 1), D(torch.rand(5, generator=torch.Generator().manual_seed(5), dtype=torch.double)), D(torch.zeros(5, dtype=torch.double)), diag=True),
        "attention_bad_width": lambda: nu.QKVAttention(5)(D(qkv)),
        "attention_ok": lambda: nu.QKVAttention(4)(D(qkv)),
    }


