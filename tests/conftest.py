import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def group(npz, prefix):
    """All arrays under ``prefix/`` as {rest_of_key: torch tensor}."""
    out = {}
    pre = prefix + "/"
    for k in npz.files:
        if k.startswith(pre):
            a = npz[k]
            out[k[len(pre):]] = torch.from_numpy(a) if a.dtype.kind in "fiu" else a
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max(|b|_inf, tiny): the relative metric used for every fp32 parity bound."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    denom = max(b.abs().max().item(), 1e-30)
    return (a - b).abs().max().item() / denom


@pytest.fixture(scope="session")
def has_gpu():
    return torch.cuda.is_available()
