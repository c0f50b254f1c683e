"""Which destructor aborts when a captured step is left to the garbage collector?

Hypothesis (from torch's graph wrapper as shipped for ROCm >= 6.2): ``at::cuda::CUDAGraph::~CUDAGraph`` ends with
``hipDeviceSynchronize()`` under ``AT_CUDA_CHECK`` ("hipGraphExecDestroy does not free at once, so wait for the launches").  A
device synchronize is an illegal call while the calling thread has a stream capture open: it returns
``hipErrorStreamCaptureUnsupported``, the check throws out of a (noexcept) destructor, ``std::terminate`` -> SIGABRT.  And since
``torch.cuda.graph.__enter__`` no longer runs ``gc.collect()`` (``torch.compiler.config.force_cudagraph_gc`` is False), a graph that
sits in a dead reference cycle is destroyed at whatever allocation makes the cyclic collector run -- inside the NEXT capture if
that is where it happens.

Each scenario runs ONCE, in a child process of its own, stdout + stderr kept under ``gpurun_out/r04_teardown/``:

    python tools/probe/graph_teardown.py            # parent never touches the GPU
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PRELUDE = r"""
import faulthandler, gc, sys, torch
faulthandler.enable()
gc.disable()                       # the collector runs only where a scenario calls it
dev = torch.device("cuda", 0)
a = torch.ones(1 << 20, device=dev)

def small_graph(mode="thread_local"):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
        b = a * 2
    return g, b

class Cycle:                       # a dead reference cycle owning `payload`: only the cyclic collector frees it
    def __init__(self, payload):
        self.payload = payload
        self.me = self

def capture_and_collect(mode):
    g2 = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g2, stream=s, capture_error_mode=mode):
        c = a + 1
        print("collecting inside the capture", flush=True)
        n = gc.collect()
        print("collected", n, flush=True)
        d = c + 1
    g2.replay(); torch.cuda.synchronize()
    print("second graph replayed, d[0] =", d[0].item(), flush=True)
"""

SCENARIOS = {
    # control: graph dropped by reference count with the device idle
    "a_refcount_idle": "g, b = small_graph(); g.replay(); torch.cuda.synchronize(); del g, b; print('ok')",
    # control: graph in a dead cycle, collected with the device idle, no capture open
    "b_cycle_collected_idle": "g, b = small_graph(); g.replay(); c = Cycle((g, b)); del g, b, c; gc.collect(); torch.cuda.synchronize(); print('ok')",
    # control: graph in a dead cycle, collected while its replay may still be in flight (destructor synchronises), no capture open
    "c_cycle_collected_in_flight": "g, b = small_graph(); [g.replay() for _ in range(200)]; c = Cycle((g, b)); del g, b, c; gc.collect(); torch.cuda.synchronize(); print('ok')",
    # THE CASE: graph in a dead cycle, collector runs while this thread captures the next graph (thread_local mode, the engine's)
    "d_cycle_collected_in_capture_thread_local": "g, b = small_graph(); g.replay(); torch.cuda.synchronize(); c = Cycle((g, b)); del g, b, c; capture_and_collect('thread_local')",
    # same under the default global mode
    "e_cycle_collected_in_capture_global": "g, b = small_graph('global'); g.replay(); torch.cuda.synchronize(); c = Cycle((g, b)); del g, b, c; capture_and_collect('global')",
    # same under relaxed mode (the segmented capture's)
    "f_cycle_collected_in_capture_relaxed": "g, b = small_graph('relaxed'); g.replay(); torch.cuda.synchronize(); c = Cycle((g, b)); del g, b, c; capture_and_collect('relaxed')",
    # not the graph: events and a stream object in a dead cycle, collected inside a capture
    "g_events_collected_in_capture": "e = [torch.cuda.Event() for _ in range(8)]; [x.record() for x in e]; s_ = torch.cuda.Stream(); c = Cycle((e, s_)); del e, s_, c; capture_and_collect('thread_local')",
    # not the graph: a plain device tensor in a dead cycle, collected inside a capture (allocator free during capture)
    "h_tensor_collected_in_capture": "t = torch.ones(1 << 22, device=dev); c = Cycle(t); del t, c; capture_and_collect('thread_local')",
    # a tensor of the FIRST graph's private pool in a dead cycle (graph itself already released by refcount), collected inside a capture
    "i_pool_tensor_collected_in_capture": "g, b = small_graph(); g.replay(); torch.cuda.synchronize(); c = Cycle(b); del g, b, c; capture_and_collect('thread_local')",
    # the direct statement of the hypothesis: torch.cuda.synchronize() inside a thread-local capture is a Python exception here
    # (it is the same illegal call, but raised where it can be caught); shows the HIP error text
    "j_device_sync_in_capture": '''
g2 = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.graph(g2, stream=s, capture_error_mode='thread_local'):
        c = a + 1
        torch.cuda.synchronize()
except Exception as e:
    print('exception:', type(e).__name__, str(e)[:400])
''',
}


def main():
    out = os.path.join(ROOT, "gpurun_out", "r04_teardown")
    os.makedirs(out, exist_ok=True)
    only = sys.argv[1:]
    rows = []
    for name, body in SCENARIOS.items():
        if only and name not in only:
            continue
        log = os.path.join(out, name + ".log")
        with open(log, "w") as f:
            r = subprocess.run([sys.executable, "-c", PRELUDE + "\n" + body], stdout=f, stderr=subprocess.STDOUT, timeout=300)
        rows.append((name, r.returncode))
        print(f"{name:48s} exit {r.returncode}", flush=True)
    with open(os.path.join(out, "summary.txt"), "w") as f:
        for name, rc in rows:
            f.write(f"{name:48s} exit {rc}\n")


if __name__ == "__main__":
    main()
