"""Host logic of ``engine/lifetime.py`` (no GPU): ownership, parking while a capture of the package is open, draining at the next
safe point, finalizers of dropped owners.  The graphs here are stand-ins that count their ``release()`` calls; the real
``torch.cuda.CUDAGraph`` destructor's behaviour is pinned on the GPU (tests/test_gpu_lifetime.py)."""
import gc
import weakref

import pytest
import torch

from ot_vae_lightning_amd.engine import lifetime
from ot_vae_lightning_amd.engine.lifetime import GraphSet, capture_guard


class FakeGraph:
    log = []

    def __init__(self, name):
        self.name = name

    def release(self):
        FakeGraph.log.append(self.name)


@pytest.fixture(autouse=True)
def _no_device_sync(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    FakeGraph.log = []
    assert lifetime.parked() == 0
    yield
    assert lifetime.parked() == 0


def test_release_destroys_in_reverse_capture_order_and_is_idempotent():
    gs = GraphSet(torch.device("cpu"))
    for n in ("fb", "b2", "opt"):
        gs.put(n, FakeGraph(n))
    assert gs and gs.get("b2").name == "b2"
    with pytest.raises(RuntimeError):
        gs.put("fb", FakeGraph("again"))
    gs.release()
    assert FakeGraph.log == ["opt", "b2", "fb"] and not gs and gs.get("fb") is None
    gs.release()
    assert FakeGraph.log == ["opt", "b2", "fb"]


def test_release_inside_a_capture_parks_and_the_next_capture_drains_first():
    a, b = GraphSet(torch.device("cpu")), GraphSet(torch.device("cpu"))
    a.put("fb", FakeGraph("a"))
    b.put("fb", FakeGraph("b"))
    was_enabled = gc.isenabled()
    with capture_guard():
        assert lifetime.capture_open() and not gc.isenabled()      # the cyclic collector is off while a capture is open
        a.release()
        assert FakeGraph.log == [] and lifetime.parked() == 1      # not destroyed inside the capture
        assert lifetime.drain() == 0                                # ... and not drainable there either
        with capture_guard():                                       # nesting keeps the count
            assert lifetime.capture_open()
        assert lifetime.capture_open()
    assert gc.isenabled() == was_enabled and not lifetime.capture_open()
    assert lifetime.parked() == 1
    with capture_guard():                                           # the next capture destroys the parked graphs BEFORE it begins
        assert FakeGraph.log == ["a"] and lifetime.parked() == 0
    b.release()                                                     # outside a capture: at once
    assert FakeGraph.log == ["a", "b"]


def test_a_dropped_owner_releases_through_its_finalizer_also_from_the_collector_inside_a_capture():
    class Owner:
        def __init__(self, name):
            self.gs = GraphSet(torch.device("cpu"))
            self.gs.put("fb", FakeGraph(name))
            weakref.finalize(self, GraphSet.release, self.gs)

    o = Owner("refcount")
    del o                                                           # reference count: released immediately
    assert FakeGraph.log == ["refcount"]
    o = Owner("cycle")
    o.me = o                                                        # only the cyclic collector frees it
    del o
    with capture_guard():
        gc.collect()                                                # ... and it runs inside a capture
        assert FakeGraph.log == ["refcount"] and lifetime.parked() == 1
    assert lifetime.drain() == 1 and FakeGraph.log == ["refcount", "cycle"]


def test_a_graphed_model_deep_copies_onto_its_own_copy():
    """``model.loss = GraphedNelbo(model)`` holds the model weakly (no reference cycle); ``copy.deepcopy(model)`` must bind the copied
    ``loss`` to the COPY, not to the original."""
    import copy
    import ot_vae_lightning_amd as A
    enc = A.CNN(1, 16, 16, 1, capacity=4, down_sample=True, residual="add")
    dec = A.CNN(8, 1, 1, 16, capacity=4, up_sample=True, residual="add")
    model = A.VAE(encoder=enc, decoder=dec, prior=A.GaussianPrior(loss_coeff=0.1)).enable_graphed_step()
    twin = copy.deepcopy(model)
    assert twin.loss is not model.loss and twin.loss.model is twin and model.loss.model is model and twin.loss._cap is None
    with pytest.raises(copy.Error):
        copy.deepcopy(model.loss)
    with pytest.raises(TypeError):
        import pickle
        pickle.dumps(model.loss)
    ref = weakref.ref(model)
    graphed = weakref.ref(model.loss)
    del model
    gc.collect()
    assert ref() is None and graphed() is None     # nothing but the model kept its GraphedNelbo alive


def test_data_parallel_overlap_default_follows_the_gradient_size():
    """The backward cut that overlaps the decoder's all-reduce costs the step 0.18 ms by itself (profiles/r04_dp_host_enqueue.txt): the
    default takes it only for gradient buffers of tens of MB; the environment switch decides outright; one rank never cuts."""
    from ot_vae_lightning_amd.engine import trainer as T
    mnist, big = 1717319 * 4, 64 << 20
    assert T.DP_OVERLAP_MIN_BYTES == 32 << 20
    assert not T.default_dp_overlap(1, big) and not T.default_dp_overlap(1, big, "1")
    assert not T.default_dp_overlap(8, mnist) and T.default_dp_overlap(8, big)
    assert T.default_dp_overlap(8, mnist, "1") and not T.default_dp_overlap(8, big, "0")
