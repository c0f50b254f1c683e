"""Host time of one captured-step replay (no synchronisation between replays) vs the device time per step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
import ot_vae_lightning_amd as A
from ot_vae_lightning_amd.utils.synthetic import mnist_like

if os.environ.get("SKIP_WGRAD") == "1":  # lower bound of the launch stream's chain: weight-gradient jobs dropped (timing only)
    from ot_vae_lightning_amd import functional as HF

    def _drop(device, ev):
        q = HF._PendingReduce._wq[device]
        q[0].clear()
        q[1] = 0
    HF._PendingReduce.issue = staticmethod(_drop)
if os.environ.get("EMPTY_FORKS") == "1":  # probe: keep the fork / join edges of the side stream but launch no weight-gradient job on it
    from ot_vae_lightning_amd import functional as HF

    def _empty(device, ev):
        q = HF._PendingReduce._wq[device]
        if not q[0]:
            return
        side = HF._PendingReduce.side_stream(device)
        side.wait_event(ev)
        if os.environ.get("TINY_SIDE") == "1":  # one trivial kernel per fork: the graph keeps its second branch
            import ctypes as C
            from ot_vae_lightning_amd import _lib as L
            global _dummy
            if "_dummy" not in globals():
                _dummy = torch.zeros(1, dtype=torch.int32, device=device)
            L.check(L.load().otvae_step_begin(L.ptr(_dummy), C.c_void_p(side.cuda_stream)), "otvae_step_begin")
        q[0].clear()
        q[1] = 0
        HF._PendingReduce._forked[device] = True
    HF._PendingReduce.issue = staticmethod(_empty)
if os.environ.get("SKIP_RANGE"):  # drop the weight-gradient jobs of backward calls lo <= index < hi (timing probe only)
    from ot_vae_lightning_amd import functional as HF
    lo, hi = map(int, os.environ["SKIP_RANGE"].split(":"))
    _orig_issue = HF._PendingReduce.issue
    _orig_flush = HF._PendingReduce.flush
    _cnt = [0]

    def _issue(device, ev):
        i = _cnt[0]
        _cnt[0] += 1
        if lo <= i < hi:
            q = HF._PendingReduce._wq[device]
            q[0].clear()
            q[1] = 0
            return
        _orig_issue(device, ev)

    def _flush(device):
        _orig_flush(device)
        if os.environ.get("SHOW_CALLS") and _cnt[0]:
            print("backward calls:", _cnt[0])
        _cnt[0] = 0
    HF._PendingReduce.issue = staticmethod(_issue)
    HF._PendingReduce.flush = staticmethod(_flush)
if os.environ.get("SKIP_BN_FIN"):  # timing probe: BatchNorm finalize launches dropped (values become garbage)
    from ot_vae_lightning_amd import _lib as L
    lib = L.load()
    which = os.environ["SKIP_BN_FIN"]

    class _Shim:
        def __init__(self, inner):
            self._inner = inner

        def __getattr__(self, name):
            if (name == "otvae_bn_finalize" and "f" in which) or (name == "otvae_bn_bwd_finalize" and "b" in which):
                return lambda *a: 0
            return getattr(self._inner, name)
    shim = _Shim(lib)
    L.load = lambda: shim
model = B.build_model(A, seed=2, workload="gaussian").cuda().train()
tr = A.HipTrainer(model, batch_shape=(1024, 1, 32, 32), data_parallel=False)
x = mnist_like(1024, seed=77).cuda()
for _ in range(5):
    tr.step(x)
torch.cuda.synchronize()
for n in (1, 5, 50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.step(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%d host enqueue %.3f ms/step, until done %.3f ms/step" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
