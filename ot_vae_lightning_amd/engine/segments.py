"""A captured training step as a CHAIN OF LINEAR hipGraphs plus side graphs, instead of one graph with ~50 forks.

Round 2 moved every layer's weight-gradient jobs onto a second stream inside the captured step.  That paid
(3.10 -> 2.84 ms) but a hipGraph with more than one branch leaves the executor's batched-submission path: the same forks with ONE
trivial kernel each cost 0.15 ms of device time per step and 0.8 ms of host time per replay.  Here the launch stream's work is
captured as a few linear graphs (cut every ``SEGMENT_CALLS`` backward calls), the weight-gradient jobs of a segment as a linear
graph of their own, and the order between the two streams is set by events BETWEEN graph launches:

    launch stream:  G0 (forward ... first backward calls) | G1 | G2 | ... | join | tail (leftover reduction, optimizer)
    side stream  :            wait(G0 done) S0            | wait(G1 done) S1 | ...

Same kernels, same arguments, same per-stream order as the forked capture: bit-identical results
(tests/test_gpu_configs.py::test_segmented_capture_equals_forked_capture).

Memory: all graphs share one private pool.  Whatever a side graph reads or writes is allocated while the launch stream's graphs
are being captured and is HELD until every graph of the step has been captured, so no later allocation of the same step can
land on it; at replay the launch stream joins the side stream before the tail graph and the next step begins behind the tail.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional

import torch

from .. import _lib

__all__ = ["SegmentedStep", "SEGMENT_CALLS"]

# backward calls (ConvBlock stages) per launch-stream graph; 0 (default) = the forked single graph of round 2.
# MEASURED on the MI355X (profiles/r03_segments_ab.txt, batch 1024, ms per step / host enqueue ms per step):
#   forked graph 2.81 / 2.32;  12-call segments (3 graphs) 2.96 / 0.26;  6-call (5 graphs) 3.03 / 0.34;  3-call (10 graphs) 3.11 / 0.51.
# Every graph-to-graph boundary on the launch stream costs the DEVICE ~35 us (more than the ~3 us per fork it removes) even
# though the launches are enqueued ahead; the host cost per replay does fall 7-9x.  The segmented capture therefore stays an A/B
# switch (OTVAE_SEGMENT_CALLS=n): it is the better shape only where the host, not the device, bounds the step.
SEGMENT_CALLS = int(os.environ.get("OTVAE_SEGMENT_CALLS", "0"))


class SegmentedStep:
    """Records a step as [(main graph, side closure | None, join_before)] and replays it.  Capture protocol:

        seg = SegmentedStep(device, side_stream)
        with seg:                       # begins the first launch-stream graph on a capture stream
            ...                         # kernels; seg.cut(side_work) ends the current graph, begins the next one
            seg.cut(join=True)          # the next graph starts behind everything the side stream was given
            ...
        seg.replay()
    """

    def __init__(self, device, side_stream: "torch.cuda.Stream", pool=None, stream: Optional["torch.cuda.Stream"] = None):
        self.device = device
        # the capture stream.  Autograd runs a node's backward on the stream its forward ran on: a backward pass that is cut into
        # segments must be captured on the SAME stream its forward pass was captured on, or the engine's cross-stream hand-over
        # (a fork of the capture) is still open when a cut ends the graph ("capturing stream has unjoined work")
        self._stream = stream
        self.side_stream = side_stream
        self.main: List[torch.cuda.CUDAGraph] = []
        self.side: List[Optional[torch.cuda.CUDAGraph]] = []      # side[k] runs behind main[k]
        self.join_before: List[bool] = []                           # main[k] waits for the side stream first
        self._side_work: List[Optional[Callable[[], None]]] = []
        self._pending_join = False
        self._cur: Optional[torch.cuda.CUDAGraph] = None
        # one private pool for all graphs of the step (None: a fresh one; else the pool of graphs captured before, whose tensors
        # are read here)
        self._pool = pool if pool is not None else torch.cuda.graph_pool_handle()
        self._ctx = None
        self._events: List[torch.cuda.Event] = []
        self._join_event = torch.cuda.Event()
        self._held: list = []   # tensors the side graphs touch, kept until those graphs have been captured
        self.capturing = False

    # ---- capture -----------------------------------------------------------------------------------------------
    def _begin_graph(self):
        g = torch.cuda.CUDAGraph()
        # "relaxed": the autograd engine ends / begins captures from its worker thread (a cut lands inside backward)
        g.capture_begin(pool=self._pool, capture_error_mode="relaxed")
        self._cur = g
        self.join_before.append(self._pending_join)
        self._pending_join = False

    def __enter__(self):
        torch.cuda.synchronize(self.device)
        if self._stream is None:
            self._stream = _lib.fresh_stream(self.device)
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        self._ctx = torch.cuda.stream(self._stream)
        self._ctx.__enter__()
        self.capturing = True
        self._begin_graph()
        return self

    def cut(self, side_work: Optional[Callable[[], None]] = None, join: bool = False) -> None:
        """Ends the current launch-stream graph.  ``side_work`` (a closure that issues kernels on the CURRENT stream) becomes the
        side graph that runs behind it; ``join``: the next launch-stream graph waits for the side stream."""
        self._cur.capture_end()
        self.main.append(self._cur)
        self._side_work.append(side_work)
        self._pending_join = join
        self._begin_graph()

    def __exit__(self, exc_type, exc, tb):
        try:
            if self._cur is not None:
                try:
                    self._cur.capture_end()
                except Exception:  # noqa: BLE001
                    if exc_type is None:
                        raise      # (an exception already under way is the one to report; the capture is invalid either way)
                if exc_type is None:
                    self.main.append(self._cur)
                    self._side_work.append(None)
        finally:
            self._cur = None
            self.capturing = False
            self._ctx.__exit__(exc_type, exc, tb)
        if exc_type is not None:
            return False
        # the side graphs, each a linear capture on the side stream, in the pool of the launch stream's graphs
        torch.cuda.synchronize(self.device)
        for work in self._side_work:
            if work is None:
                self.side.append(None)
                continue
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(self.side_stream):
                g.capture_begin(pool=self._pool, capture_error_mode="relaxed")
                try:
                    work()
                finally:
                    g.capture_end()
            self.side.append(g)
        self._side_work = []
        self._held = []
        self._events = [torch.cuda.Event() for _ in self.main]
        torch.cuda.synchronize(self.device)
        return False

    # ---- replay ------------------------------------------------------------------------------------------------
    def replay(self) -> None:
        cur = torch.cuda.current_stream(self.device)
        side = self.side_stream
        used_side = False
        for k, g in enumerate(self.main):
            if self.join_before[k] and used_side:
                self._join_event.record(side)
                cur.wait_event(self._join_event)
                used_side = False
            g.replay()
            sg = self.side[k]
            if sg is not None:
                ev = self._events[k]
                ev.record(cur)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    sg.replay()
                used_side = True
        if used_side:  # nothing of this step may still run when the caller's next work starts
            self._join_event.record(side)
            cur.wait_event(self._join_event)

    def release(self) -> None:
        self.main, self.side = [], []
