"""Deterministic lifetime of captured hipGraphs.

THE FACT THIS MODULE EXISTS FOR (diagnosed in round 4, ``tools/probe/graph_teardown.py``, ``profiles/r04_graph_teardown.txt``):
torch's graph wrapper for ROCm >= 6.2 ends its destructor with ``hipDeviceSynchronize()`` under ``AT_CUDA_CHECK``
(``at::cuda::CUDAGraph::~CUDAGraph``, HIPGraph.cpp:324 -- hipGraphExecDestroy frees lazily, so the wrapper waits for the
launches).  A device synchronize is illegal while the calling thread has a stream capture open: it returns
``hipErrorStreamCaptureUnsupported``, the check throws out of the destructor and the process dies with SIGABRT
("terminate called after throwing an instance of 'c10::AcceleratorError' ... operation not permitted when stream is
capturing"), in every capture error mode.  ``torch.cuda.graph.__enter__`` no longer collects garbage before a capture
(``torch.compiler.config.force_cudagraph_gc`` is False), so a graph that sits in a dead reference cycle is destroyed at whatever
allocation makes the cyclic collector run -- inside the NEXT capture if that is where the allocation happens.  Events, streams,
plain tensors and tensors of a graph's private pool are harmless there (probe scenarios g, h, i exit 0).

Rules enforced here:

* a step's graphs have exactly ONE strong owner, a ``GraphSet``; nobody else keeps a reference to a ``CUDAGraph`` object;
* graphs are destroyed only by ``_destroy``: the device is synchronised first and the calling thread has no capture open;
* ``GraphSet.release()`` asked for while a capture is open (an owner dropped by reference count or by the collector inside
  somebody's capture) PARKS the graphs; they are destroyed at the next safe point (``drain()``: the start of the next capture,
  the next ``release`` / ``close`` outside a capture);
* every capture of this package runs inside ``capture_guard()``: parked graphs are destroyed before the capture begins and the
  cyclic collector is switched off until it ends, so that no destructor of a foreign object graph (a user's own dropped
  ``CUDAGraph`` in a cycle) can run inside it either;
* owners attach ``weakref.finalize(owner, GraphSet.release, graphset)``: an owner that is dropped without ``close()`` releases
  its graphs through the same ordered path the moment it dies (the finalizers are NOT run at interpreter exit: no capture can be
  open there, and what is still alive then goes down with the process as it always has).
"""
from __future__ import annotations

import gc
import threading
from typing import Dict, List

import torch

__all__ = ["GraphSet", "capture_guard", "capture_open", "drain"]

_lock = threading.RLock()
_parked: List[tuple] = []        # (device, [graph objects]) waiting for a point where destroying them is legal
_open = threading.local()        # .n = captures of this package open on the calling thread


def capture_open() -> bool:
    """True when destroying a graph now could land inside a stream capture of the calling thread."""
    if getattr(_open, "n", 0) > 0:
        return True
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _destroy(device, graphs: list) -> None:
    """The one place graphs die: device idle, no capture open on this thread.  Later captures first (they allocate from the
    first graph's pool)."""
    torch.cuda.synchronize(device)
    while graphs:
        g = graphs.pop()
        rel = getattr(g, "release", None)   # SegmentedStep: a chain of graphs
        if rel is not None:
            rel()
        del g


def drain() -> int:
    """Destroys what ``release`` had to park.  No-op (returns 0) while a capture is open."""
    if capture_open():
        return 0
    with _lock:
        work, _parked[:] = list(_parked), []
    for device, graphs in work:
        _destroy(device, graphs)
    return len(work)


def parked() -> int:
    return len(_parked)


class GraphSet:
    """The single strong owner of one captured step's graphs (``torch.cuda.CUDAGraph`` or ``SegmentedStep``), by name, in capture
    order."""

    def __init__(self, device):
        self.device = device
        self._graphs: Dict[str, object] = {}

    def new(self, name: str):
        g = torch.cuda.CUDAGraph()
        self.put(name, g)
        return g

    def put(self, name: str, graph) -> None:
        if name in self._graphs:
            raise RuntimeError(f"graph `{name}` is still owned: release() first")
        self._graphs[name] = graph

    def get(self, name: str):
        return self._graphs.get(name)

    def __bool__(self) -> bool:
        return bool(self._graphs)

    def release(self) -> None:
        """Idempotent.  Outside a capture: synchronise the device, destroy the graphs in reverse capture order.  Inside one: park
        them for ``drain()``."""
        graphs, self._graphs = list(self._graphs.values()), {}
        if not graphs:
            if not capture_open():
                drain()
            return
        if capture_open():
            with _lock:
                _parked.append((self.device, graphs))
            return
        drain()
        _destroy(self.device, graphs)


class capture_guard:
    """``with capture_guard():`` around every stream capture of this package (see the module docstring)."""

    def __enter__(self):
        drain()
        self._gc = gc.isenabled()
        gc.disable()
        _open.n = getattr(_open, "n", 0) + 1
        return self

    def __exit__(self, *exc):
        _open.n -= 1
        if self._gc:
            gc.enable()
        return False
