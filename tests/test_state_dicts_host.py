"""state_dict keys, shapes, dtypes and requires_grad flags of the distribution models, transport operators and priors against the
reference's own classes (tests/golden/state_dicts.npz, recorded by oracle/gen_golden.py): what a checkpoint of the reference holds is what
``load_state_dict`` here accepts.  Construction only: host logic, no GPU."""
import os

import numpy as np
import pytest
import torch

import ot_vae_lightning_amd as A

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "state_dicts.npz"))

CASES = {
    "gaussian_full": (A.GaussianModel, (3, 4), dict(dtype=torch.double)),
    "gaussian_diag": (A.GaussianModel, (4,), dict(dtype=torch.double, w2_cfg=dict(diag=True))),
    "gaussian_autograd": (A.GaussianModel, (4,), dict(dtype=torch.double, update_with_autograd=True)),
    "gaussian_autograd_diag": (A.GaussianModel, (2, 4), dict(dtype=torch.double, update_with_autograd=True, w2_cfg=dict(diag=True))),
    "gmm_diag": (A.GaussianMixtureModel, (2, 3), dict(dtype=torch.double, mixture_cfg=dict(n_components=4), w2_cfg=dict(diag=True))),
    "gmm_full_autograd": (A.GaussianMixtureModel, (3,), dict(dtype=torch.double, mixture_cfg=dict(n_components=4), update_with_autograd=True)),
    "codebook": (A.CodebookModel, (2, 3), dict(mixture_cfg=dict(n_components=5))),
    "codebook_autograd": (A.CodebookModel, (3,), dict(mixture_cfg=dict(n_components=5), update_with_autograd=True)),
    "gaussian_transport": (A.GaussianTransport, (2, 4), dict(source_cfg=dict(dtype=torch.double), target_cfg=dict(dtype=torch.double), store_source=True)),
    "gmm_transport": (A.GMMTransport, (4,), dict(transport_type="argmax", transport_cfg=dict(diag=True, dtype=torch.double),
                      source_cfg=dict(dtype=torch.double, mixture_cfg=dict(n_components=3)), target_cfg=dict(dtype=torch.double, mixture_cfg=dict(n_components=3)))),
    "discrete_transport": (A.DiscreteTransport, (4,), dict(transport_type="argmax", source_cfg=dict(mixture_cfg=dict(n_components=3)),
                           target_cfg=dict(mixture_cfg=dict(n_components=3)))),
    "gaussian_prior": (A.GaussianPrior, (), dict(loss_coeff=0.5)),
    "cond_prior": (A.ConditionalGaussianPrior, (), dict(dim=(2, 3), num_classes=4)),
    "cond_prior_ema": (A.ConditionalGaussianPrior, (), dict(dim=(2, 3), num_classes=4, embedding_ema_decay=0.9)),
    "codebook_prior": (A.CodebookPrior, ((8, 2, 2), (1,)), dict(loss="kl", mixture_cfg=dict(n_components=6))),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_state_dict_layout_equals_the_reference(name):
    cls, args, kw = CASES[name]
    m = cls(*args, **kw)
    got = sorted(f"{k}|{tuple(v.shape)}|{v.dtype}" for k, v in m.state_dict().items())
    want = sorted(str(r) for r in Z[f"{name}/state"] if str(r))
    assert got == want, (sorted(set(got) ^ set(want)))
    got_p = sorted(f"{k}|{int(p.requires_grad)}" for k, p in m.named_parameters())
    want_p = sorted(str(r) for r in Z[f"{name}/params"] if str(r))
    assert got_p == want_p, (sorted(set(got_p) ^ set(want_p)))
