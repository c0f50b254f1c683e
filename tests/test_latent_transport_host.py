"""Host logic of ``LatentTransport`` (reference ot/transport_callback.py:173-287): which latents reach the transport
operator and when, the [B,C,H,W] <-> operator layouts, class filtering and error behaviour.  No GPU: the operator is a
recording stand-in with the ``TransportOperator`` interface."""
import pytest
import torch

from ot_vae_lightning_amd.ot.transport_callback import ConditionalLatentTransport, LatentTransport


class RecordingOperator(torch.nn.Module):
    made = []

    def __init__(self, *size, **kwargs):
        super().__init__()
        self.size, self.kwargs, self.calls = size, kwargs, []
        RecordingOperator.made.append(self)

    def update(self, source_samples=None, target_samples=None):
        self.calls.append(("source", source_samples) if source_samples is not None else ("target", target_samples))

    def reset(self):
        self.calls.append(("reset", None))

    def compute(self):
        return torch.tensor([1.0, 3.0])

    def forward(self, x):
        return 2 * x


class Module:
    training = True

    def __init__(self):
        self.encoded, self.log_calls = [], []

    def encode(self, x, **kw):
        self.encoded.append((x, kw))
        return x[:, :2, ::2, ::2] * 10          # [B, 2, 2, 2] "latents"

    def decode(self, z, **kw):
        return z

    def eval(self):
        self.training = False

    def train(self):
        self.training = True

    def log(self, name, value, **kw):
        self.log_calls.append((name, float(value)))


def make(**kw):
    cfg = dict(size=(2, 2, 2), transport_dims=(1,), transport_operator=RecordingOperator, transformations=lambda x: x + 100.0,
               logging_prefix="t")
    cfg.update(kw)
    return LatentTransport(**cfg)


def batch(i, n=3):
    return {"samples": torch.arange(n * 4 * 4 * 4, dtype=torch.float32).reshape(n, 4, 4, 4) + 1000 * i}


def test_operator_size_and_layouts():
    cb = make()                                    # per-position operators: [H*W, B, C]
    assert cb.transport_operator.size == (2, 2, 2) and cb.dim == 2 and tuple(cb.batch_shape) == (2, 2)
    z = torch.arange(3 * 2 * 2 * 2, dtype=torch.float32).reshape(3, 2, 2, 2)
    flat = cb._permute_and_flatten(z)
    assert flat.shape == (4, 3, 2)
    assert torch.equal(flat[1, 2], z[2, :, 0, 1])  # position (0, 1), image 2 -> its channel vector
    assert torch.equal(cb._unflatten_and_unpermute(flat), z)
    assert torch.equal(cb.transport(z), 2 * z)
    cb = make(common_operator=True)                # one operator: [B*H*W, C]
    assert cb.transport_operator.size == (2,)
    flat = cb._permute_and_flatten(z)
    assert flat.shape == (12, 2) and torch.equal(flat[2 * 4 + 1], z[2, :, 0, 1])
    assert torch.equal(cb._unflatten_and_unpermute(flat), z)
    cb = make(transport_dims=(1, 2, 3), common_operator=True)  # whole latent: [B, C*H*W]
    assert cb.transport_operator.size == (8,) and cb._permute_and_flatten(z).shape == (3, 8)
    assert cb.logging_prefix == "transport/recording_operator/t/"
    with pytest.raises(ValueError):
        make(transport_dims=(4,))


@pytest.mark.parametrize("unpaired", [True, False])
def test_validation_routing(unpaired):
    cb, m = make(unpaired=unpaired), Module()
    cb.on_validation_epoch_start(None, m)
    for i in range(4):
        cb.on_validation_batch_end(None, m, batch(i), None, i)
    kinds = [k for k, _ in cb.transport_operator.calls]
    # unpaired: even batches are target (clean), odd batches source (transformed); paired: both from every batch
    assert kinds == (["reset", "target", "source", "target", "source"] if unpaired else ["reset"] + ["target", "source"] * 4)
    src = [t for k, t in cb.transport_operator.calls if k == "source"][0]
    first_src_batch = 1 if unpaired else 0
    want = m.encode(batch(first_src_batch)["samples"] + 100.0)
    assert torch.equal(src, cb._permute_and_flatten(want))            # the transformation was applied before encoding
    cb.on_validation_epoch_end(None, m)
    assert m.log_calls == [("transport/recording_operator/t/avg_transport_cost", 2.0)]
    assert float(cb.logged["transport/recording_operator/t/avg_transport_cost"]) == 2.0


def test_latents_key_is_used_when_present_and_train_hooks():
    cb, m = make(target_latents_from_train=True), Module()
    z = torch.ones(3, 2, 2, 2)
    cb.on_train_batch_end(None, m, {"latents": z, "samples": batch(0)["samples"]}, None, 0)
    assert [k for k, _ in cb.transport_operator.calls] == ["target"] and not m.encoded
    # validation then only supplies the source side (every batch: the target comes from training)
    cb.on_validation_batch_end(None, m, batch(1), None, 0)
    cb.on_validation_batch_end(None, m, batch(2), None, 1)
    assert [k for k, _ in cb.transport_operator.calls] == ["target", "source", "source"]
    # both sides from training + unpaired: even -> target, odd -> source (encoded in eval mode, training restored)
    cb, m = make(target_latents_from_train=True, source_latents_from_train=True), Module()
    for i in range(4):
        cb.on_train_batch_end(None, m, {"latents": z, "samples": batch(i)["samples"]}, None, i)
    assert [k for k, _ in cb.transport_operator.calls] == ["target", "source", "target", "source"] and m.training
    cb.on_validation_batch_end(None, m, batch(0), None, 0)             # nothing left for validation to feed
    assert len(cb.transport_operator.calls) == 4
    # neither side from training: the train hook is a no-op
    cb, m = make(), Module()
    cb.on_train_batch_end(None, m, batch(0), None, 0)
    assert cb.transport_operator.calls == []


def test_class_filter_errors_and_conditional_fan_out():
    cb, m = make(class_idx=1), Module()
    out = batch(0, n=4)
    out["y"] = torch.tensor([1, 0, 1, 2])
    cb.on_validation_batch_end(None, m, out, None, 0)
    assert m.encoded[0][0].shape[0] == 2 and torch.equal(m.encoded[0][0], out["samples"][[0, 2]])
    with pytest.raises(ValueError):
        cb.on_validation_batch_end(None, m, batch(0), None, 0)          # no condition anywhere
    with pytest.raises(ValueError):
        make().on_validation_batch_end(None, m, [1, 2], None, 0)        # not a dict
    with pytest.raises(ValueError):
        make().on_validation_batch_end(None, m, {"x": 1}, None, 0)      # neither latents nor samples
    with pytest.raises(NotImplementedError):
        make().on_validation_batch_end(None, object(), batch(0), None, 0)
    with pytest.raises(NotImplementedError):
        make().sample(2, "elsewhere")
    RecordingOperator.made.clear()
    cc = ConditionalLatentTransport(3, "c", 9, size=(2, 2, 2), transport_dims=(1,), transport_operator=RecordingOperator,
                                    transformations=lambda x: x)
    assert [t.class_idx for t in cc.transports] == [0, 1, 2] and len(RecordingOperator.made) == 3
    cc.on_validation_epoch_start(None, m)
    assert all(op.calls == [("reset", None)] for op in RecordingOperator.made)


def test_routing_equals_the_reference_callback_over_the_flag_grid():
    """The same scripted run (validation-epoch start, four training batches, four validation batches; even batches carry `latents`) through
    this callback and -- recorded in tests/golden/latent_transport_routing.npz by oracle/gen_golden.py -- through the reference's own
    LatentTransport with a recording operator, for all 32 combinations of source_latents_from_train / target_latents_from_train /
    unpaired / common_operator x transport_dims in {(1,), (1,2,3)} (+ verbose and class_idx cases): the same calls in the same order with the same tensors, the
    same operator size, the same `transport()` result."""
    import numpy as np
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "latent_transport_routing.npz"))
    grid = [(s_, t_, u_, c_, dims, False, None) for s_ in (False, True) for t_ in (False, True) for u_ in (False, True)
            for c_ in (False, True) for dims in ((1,), (1, 2, 3))]
    # + a verbose callback (encode fallback on batch 0 only, as the reference) and class filtering
    grid += [(False, True, False, False, (1,), True, None), (True, True, True, True, (1,), True, None),
             (False, False, True, False, (1,), False, 1), (True, True, False, True, (2, 3), False, 0)]
    labels = torch.tensor([1, 0])
    x = [torch.arange(2 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 4, 4) + 1000.0 * i for i in range(8)]
    for idx, (src, tgt, unp, common, dims, verbose, cls) in enumerate(grid):
        assert list(z[f"{idx}/flags"]) == [int(src), int(tgt), int(unp), int(common), len(dims), int(verbose), -1 if cls is None else cls]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            cb = make(transport_dims=dims, target_latents_from_train=tgt, source_latents_from_train=src, unpaired=unp,
                      common_operator=common, verbose=verbose, class_idx=cls)
        mod = Module()
        cb.on_validation_epoch_start(None, mod)
        for b in range(4):
            o = {"samples": x[b], "kwargs": {}, "y": labels}
            if b % 2 == 0:
                o["latents"] = x[b][:, :2, ::2, ::2] * 10 + 1.0
            cb.on_train_batch_end(None, mod, o, None, b)
        for b in range(4):
            o = {"samples": x[4 + b], "kwargs": {}, "y": labels}
            if b % 2 == 0:
                o["latents"] = x[4 + b][:, :2, ::2, ::2] * 10 + 1.0
            cb.on_validation_batch_end(None, mod, o, None, b, 0)
        calls = cb.transport_operator.calls
        kinds = [{"target": 0, "source": 1, "reset": 2}[k] for k, _ in calls]
        tag = (src, tgt, unp, common, dims, verbose, cls)
        assert kinds == list(z[f"{idx}/kinds"]), tag
        for j, (_, t) in enumerate(calls):
            if t is not None:
                want = torch.from_numpy(z[f"{idx}/call{j}"])
                assert t.shape == want.shape and torch.equal(t, want), (tag, j)
        assert list(cb.transport_operator.size) == list(z[f"{idx}/op_size"]), tag
        got = cb.transport(x[0][:, :2, ::2, ::2] * 10)
        assert torch.equal(got, torch.from_numpy(z[f"{idx}/transported"])), tag


def test_layout_functions_equal_the_reference_over_all_axis_choices():
    """``utils.permute_and_flatten`` / ``unflatten_and_unpermute`` against the reference's functions (tests/golden/layouts.npz, recorded by
    oracle/gen_golden.py from utils/__init__.py:233-311): every choice and order of axes, batch_first, flatten_batch, for 2-, 4- and
    5-dimensional inputs -- the rearranged tensor bit for bit, and the round trip wherever the reference's own round trip holds."""
    import os
    import numpy as np
    from ot_vae_lightning_amd import utils as U
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "layouts.npz"))
    n = int(z["n"][0])
    assert n > 150
    for i in range(n):
        cfg = [int(v) for v in z[f"{i}/cfg"]]
        a = cfg.index(-1)
        b = cfg.index(-1, a + 1)
        shape, perm, bf, fb = tuple(cfg[:a]), tuple(cfg[a + 1:b]), bool(cfg[b + 1]), bool(cfg[b + 2])
        x = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
        y = U.permute_and_flatten(x, perm, batch_first=bf, flatten_batch=fb)
        want = torch.from_numpy(z[f"{i}/y"])
        assert y.shape == want.shape and torch.equal(y, want), (shape, perm, bf, fb)
        if int(z[f"{i}/roundtrip_ok"][0]):
            back = U.unflatten_and_unpermute(y, x.shape, perm, batch_first=bf, flatten_batch=fb)
            assert torch.equal(back, x), (shape, perm, bf, fb, "round trip")
