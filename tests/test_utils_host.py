"""The small tensor helpers of ``utils`` the hot path calls (reference utils/__init__.py:123-218) against the reference's own functions
(tests/golden/utils_small.npz, recorded by oracle/gen_golden.py): host logic on CPU tensors, no GPU."""
import os

import numpy as np
import torch

from ot_vae_lightning_amd import utils as U

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "utils_small.npz"))
T = lambda k: torch.from_numpy(Z[k])  # noqa: E731


def _same(a, b):
    # NaN == NaN here: 0 / 0 of an all-zero count row is what the reference returns too
    return a.shape == b.shape and torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))


def test_replicate_and_reduce():
    g = torch.Generator().manual_seed(201)
    x = torch.randn(4, 3, 2, generator=g)
    for n in (0, 1, 2, 3):
        r = U.replicate_batch({"a": x, "b": [x[:, 0], 7]}, n)
        assert _same(r["a"], T(f"replicate{n}/a")) and _same(r["b"][0], T(f"replicate{n}/b0")) and r["b"][1] == 7
        e = T(f"reduce{n}/in")
        assert _same(U.mean_replicated_batch(e, n), T(f"reduce{n}/mean"))
        if n > 1:
            assert _same(U.std_replicated_batch(e, n), T(f"reduce{n}/std"))


def test_ema():
    avg, new = T("ema/avg"), T("ema/new")
    for tag, decay in (("none", None), ("d0", 0.0), ("d09", 0.9), ("d1", 1.0)):
        assert _same(U.ema(avg.clone(), new, decay), T(f"ema/{tag}")), tag
        inp = avg.clone()
        U.ema_inplace(inp, new, decay)
        assert _same(inp, T(f"ema_inplace/{tag}")), tag


def test_laplace_smoothing():
    cnt = T("laplace/in")
    assert _same(U.laplace_smoothing(cnt, 4), T("laplace/eps1e-5"))
    assert _same(U.laplace_smoothing(cnt, 4, eps=0.5), T("laplace/eps0.5"))
    assert _same(U.laplace_smoothing(cnt, 4, eps=None), T("laplace/none"))
