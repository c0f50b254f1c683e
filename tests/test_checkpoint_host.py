"""Checkpoint interop (SURVEY.md §8 f, rank 3): parameter names equal to the reference's (golden list recorded from the
imported reference), ``PartialCheckpoint`` key selection / loading / freezing (reference utils/partial_checkpoint.py:
24-81), ``VisionModule.setup`` and the save/load hooks (model/base.py:192-241).  Construction and ``load_state_dict``
only -- no kernel runs, so no GPU."""
import os

import pytest
import torch

import ot_vae_lightning_amd as A
from ot_vae_lightning_amd.utils import PartialCheckpoint, human_format

from conftest import load_golden


def make_vae(seed, **kw):
    torch.manual_seed(seed)
    enc = A.CNN(1, 256, 32, 1, capacity=8, down_sample=True, residual="add")
    dec = A.CNN(128, 1, 1, 32, capacity=8, up_sample=True, residual="add")
    return A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1), **kw)


def same(a, b):
    return all(torch.equal(a[k].cpu(), b[k].cpu()) for k in a) and a.keys() == b.keys()


def test_parameter_names_are_the_references():
    names = [str(n) for n in load_golden("nelbo_mnist.npz")["add/param_names"]]
    model = make_vae(0)
    assert [n for n, _ in model.named_parameters()] == names
    sd = model.state_dict()
    assert set(names) <= set(sd)
    assert "encoder.0.block.1._normalization.running_mean" in sd      # buffers carry the reference's names as well
    w = sd["encoder.0.block.0.weight"]
    assert w.dim() == 4 and w.shape[2] == w.shape[3]                    # OIHW shape whatever the memory order


def test_partial_checkpoint_loads_one_attribute(tmp_path):
    donor, taker = make_vae(1), make_vae(2)
    path = str(tmp_path / "donor.ckpt")
    donor.export_checkpoint(path)
    blob = torch.load(path, weights_only=True)
    assert set(blob) == {"state_dict", "global_step"} and same(blob["state_dict"], donor.state_dict())

    before_dec = {k: v.clone() for k, v in taker.decoder.state_dict().items()}
    assert not same(taker.encoder.state_dict(), donor.encoder.state_dict())
    part = PartialCheckpoint(path, attr_name="encoder")
    assert all(not k.startswith("encoder.") for k in part.state_dict) and len(part.state_dict) == len(donor.encoder.state_dict())
    part.load_attribute(taker, "encoder", verbose=False)
    assert same(taker.encoder.state_dict(), donor.encoder.state_dict())
    assert same(taker.decoder.state_dict(), before_dec)
    assert all(p.requires_grad for p in taker.encoder.parameters()) and taker.encoder.training


def test_partial_checkpoint_nested_attribute_replace_and_freeze(tmp_path):
    donor, taker = make_vae(3), make_vae(4)
    path = str(tmp_path / "donor.ckpt")
    donor.export_checkpoint(path)
    # a dotted attribute: only encoder.0.* is taken ('encoder.00...' would not match: whole components are compared)
    nested = PartialCheckpoint(path, attr_name="encoder.0", freeze=True)
    assert set(nested.state_dict) == set(donor.encoder[0].state_dict())
    block = nested.load_attribute(taker, "encoder.0", verbose=False)
    assert block is taker.encoder[0] and same(block.state_dict(), donor.encoder[0].state_dict())
    assert not block.training and not any(p.requires_grad for p in block.parameters())
    assert all(p.requires_grad for p in taker.encoder[1].parameters())
    # replace_str re-roots the keys: encoder.* -> decoder.* is a shape mismatch, and says so
    rerooted = PartialCheckpoint(path, attr_name="encoder", replace_str="decoder.")
    assert all(k.startswith("decoder.") for k in rerooted.state_dict)
    with pytest.raises(RuntimeError):
        PartialCheckpoint(path, attr_name="encoder").load_attribute(taker, "decoder", verbose=False)
    # strict=False tolerates the missing prior/decoder keys when the whole file goes into the whole model
    loose = PartialCheckpoint(path, attr_name="no_such_attr", strict=False)
    assert same(loose.state_dict, donor.state_dict())                   # nothing mentions the name: file used whole
    loose.load_attribute(taker, "encoder", verbose=False)               # no key matches; strict=False lets it pass
    with pytest.raises(FileNotFoundError):
        PartialCheckpoint(str(tmp_path / "absent.ckpt"))


def test_bare_state_dict_file_and_setup_hook(tmp_path):
    donor = make_vae(5)
    path = str(tmp_path / "decoder_only.pt")
    torch.save({k: v.detach().clone() for k, v in donor.decoder.state_dict().items()}, path)
    taker = make_vae(6, checkpoints={"decoder": PartialCheckpoint(path, attr_name="decoder", freeze=True)})
    assert not same(taker.decoder.state_dict(), donor.decoder.state_dict())
    taker.setup("fit")
    assert same(taker.decoder.state_dict(), donor.decoder.state_dict())
    assert not any(p.requires_grad for p in taker.decoder.parameters())
    assert all(p is not q for p in taker.optim_parameters() for q in taker.decoder.parameters())


class Scale:                                                            # a picklable inference transform
    def __init__(self, k):
        self.k = k

    def __call__(self, x):
        return x * self.k


def test_inference_transforms_travel_with_the_checkpoint(tmp_path):
    donor = make_vae(7, inference_preprocess=Scale(2.0), inference_postprocess=Scale(0.5))
    path = str(tmp_path / "with_transforms.ckpt")
    donor.export_checkpoint(path)
    with pytest.raises(Exception):                                      # pickled objects: refused unless trusted
        _ = PartialCheckpoint(path, attr_name="encoder").state_dict
    assert len(PartialCheckpoint(path, attr_name="encoder", trusted=True).state_dict) == len(donor.encoder.state_dict())
    taker = make_vae(8)
    taker.on_load_checkpoint(torch.load(path, weights_only=False))
    assert taker.inference_preprocess.k == 2.0 and taker.inference_postprocess.k == 0.5
    taker.inference = True
    assert taker.inference


def test_human_format():
    assert [human_format(n) for n in (0, 999, 1000, 1234567, 1.72e6, 2.5e9, 3e13, 7e16)] == \
        ["0", "999", "1K", "1.23M", "1.72M", "2.5B", "30T", "70000T"]


def test_key_selection_and_human_format_equal_the_reference(tmp_path):
    """``PartialCheckpoint.state_dict`` and ``human_format`` against the reference's own functions (tests/golden/partial_checkpoint.npz,
    recorded by oracle/gen_golden.py from utils/partial_checkpoint.py:10-66): which keys an attribute name selects -- incl. a name that
    is only a substring of some key (nothing selected) and one that is absent altogether (the whole file) -- and their new names."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "partial_checkpoint.npz"))
    keys = ["encoder.0.block.0.weight", "encoder.0.block.0.bias", "encoder.0.skip.weight", "encoder.1.block.0.weight",
            "decoder.0.block.0.weight", "decoder.encoder.weight", "prior._mu.weight", "encoder_extra.weight"]
    cases = [(None, ""), ("encoder", ""), ("encoder", "enc."), ("encoder.0", ""), ("encoder.0.block", "b."), ("decoder", ""),
             ("prior", ""), ("enc", ""), ("missing", ""), ("encoder_extra", ""), ("decoder.encoder", "x.")]
    path = str(tmp_path / "c.ckpt")
    torch.save({"state_dict": {k: torch.full((1,), float(i)) for i, k in enumerate(keys)}}, path)
    for i, (attr, rep) in enumerate(cases):
        got = PartialCheckpoint(path, attr_name=attr, replace_str=rep).state_dict
        want_keys = [str(k) for k in z[f"case{i}/keys"] if str(k)]
        assert list(got.keys()) == want_keys, (attr, rep)
        assert [float(v) for v in got.values()] == [float(v) for v in z[f"case{i}/values"] if v >= 0], (attr, rep)
    for n, text in zip(z["human/nums"], z["human/text"]):
        assert human_format(float(n)) == str(text), n
