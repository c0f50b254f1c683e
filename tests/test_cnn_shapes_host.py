"""Shape inference of the CNN constructor (reference networks/cnn.py:605-672) against the reference's own functions over a grid
(tests/golden/cnn_shapes.npz, recorded by oracle/gen_golden.py): host logic, no GPU."""
import os

import numpy as np

from ot_vae_lightning_amd.networks import cnn as C

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cnn_shapes.npz"))


def test_get_channel_list_grid():
    rows = Z["channel_list"]
    assert len(rows) > 2000
    for r in rows:
        cin, cout, rin, rout, sf, cap, n = (int(v) for v in r[:7])
        feats, res = C.get_channel_list(cin, cout, rin, rout, sf, cap)
        assert list(feats) == [int(v) for v in r[7:7 + n]], (cin, cout, rin, rout, sf, cap)
        assert list(res) == [int(v) for v in r[7 + n:7 + 2 * n]], (cin, cout, rin, rout, sf, cap)


def test_div_sqrt_up_to_600():
    assert [int(C.div_sqrt(n)) for n in range(1, 601)] == [int(v) for v in Z["div_sqrt"]]


def test_get_block_scaling_grid():
    for r in Z["block_scaling"]:
        hi, lo, m, n = (int(v) for v in r[:4])
        assert list(C.get_block_scaling(hi, lo, m)) == [int(v) for v in r[4:4 + n]], (hi, lo, m)
