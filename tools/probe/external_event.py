"""probe: do external events (torch.cuda.Event(external=True): event record / wait NODES inside captured graphs) work on this
ROCm, and what do two LINEAR graphs tied by such events cost against one graph with a fork per side kernel?
Chain: N main kernels; after every second one a side kernel that depends on it; everything joined at the end."""
import ctypes
import sys
import time

import torch

# torch refuses Event(external=True) on ROCm ("External events are disallowed in rocm"); the HIP runtime has the calls
def _loaded_hip():
    for line in open("/proc/self/maps"):   # the runtime torch itself loaded (a second copy would not know torch's streams)
        if "libamdhip64" in line:
            return line.split()[-1]
    return "libamdhip64.so"


torch.cuda.init()
_hip = ctypes.CDLL(_loaded_hip())
print("HIP runtime:", _loaded_hip())
_hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
_hip.hipEventRecordWithFlags.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
_hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]


class ExtEvent:
    def __init__(self):
        self.h = ctypes.c_void_p()
        assert _hip.hipEventCreateWithFlags(ctypes.byref(self.h), 0x2) == 0          # hipEventDisableTiming

    def record(self):
        rc = _hip.hipEventRecordWithFlags(self.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), 0x1)   # hipEventRecordExternal
        if rc:
            raise RuntimeError(f"hipEventRecordWithFlags(external) -> {rc}")

    def wait(self, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream()
        rc = _hip.hipStreamWaitEvent(ctypes.c_void_p(st.cuda_stream), self.h, 0x1)   # hipEventWaitExternal
        if rc:
            raise RuntimeError(f"hipStreamWaitEvent(external) -> {rc}")


def main():
    dev = torch.device("cuda")
    n_main, every = 200, 4
    a = torch.zeros(1 << 16, device=dev)
    side_out = [torch.zeros(1 << 16, device=dev) for _ in range(n_main // every)]
    s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()

    def main_kernel():
        a.add_(1.0)

    def side_kernel(i):
        side_out[i].copy_(a)      # reads what the main chain had at the fork

    # ---- variant 1: one forked graph
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, stream=s_main):
        for k in range(n_main):
            main_kernel()
            if k % every == every - 1:
                s_side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s_side):
                    side_kernel(k // every)
        torch.cuda.current_stream().wait_stream(s_side)

    # ---- variant 2: two linear graphs, external events
    ok2 = True
    try:
        evs = [ExtEvent() for _ in range(n_main // every)]
        done = ExtEvent()
        gA, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gA, stream=s_main):
            for k in range(n_main):
                main_kernel()
                if k % every == every - 1:
                    evs[k // every].record()
        with torch.cuda.graph(gB, stream=s_side):
            for i in range(n_main // every):
                evs[i].wait()
                side_kernel(i)
            done.record()
    except Exception as e:  # noqa: BLE001
        ok2 = False
        print("external events under capture failed:", type(e).__name__, e)

    # ---- variant 3: the main chain alone (no side work)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=s_main):
        for k in range(n_main):
            main_kernel()

    def run2():
        with torch.cuda.stream(s_main):
            gA.replay()
        with torch.cuda.stream(s_side):
            gB.replay()
        done.wait(s_main)

    def timed(fn, reps=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return (t2 - t0) / reps * 1e3, (t1 - t0) / reps * 1e3

    def run1():
        with torch.cuda.stream(s_main):
            g1.replay()

    def run3():
        with torch.cuda.stream(s_main):
            g3.replay()

    print("forked graph           : %.3f ms per replay, host enqueue %.3f ms" % timed(run1))
    print("main chain alone       : %.3f ms per replay, host enqueue %.3f ms" % timed(run3))
    if ok2:
        # correctness: side_out[i] must hold the main chain's value at its fork
        a.zero_()
        torch.cuda.synchronize()
        run2()
        torch.cuda.synchronize()
        want = [float((i + 1) * every) for i in range(n_main // every)]
        got = [float(t[0]) for t in side_out]
        print("external events: dependency respected:", got == want, got[:4], want[:4])
        print("two linear graphs + ext: %.3f ms per replay, host enqueue %.3f ms" % timed(run2))
        a.zero_()
        torch.cuda.synchronize()
        for _ in range(20):
            run2()
        torch.cuda.synchronize()
        got = [float(t[0]) for t in side_out]
        want = [float(19 * n_main + (i + 1) * every) for i in range(n_main // every)]
        print("after 20 back-to-back replays still respected:", got == want)


if __name__ == "__main__":
    sys.exit(main())
