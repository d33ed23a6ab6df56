"""Shape-keyed hipGraph capture/replay around a callable.

Counterpart of reference optimizers/cuda/graphs.py:13-153
(`make_dynamic_graphed_callable`): one graph per distinct input signature,
static input/output buffers, outputs handed back as copies, a lock around
creation and replay.  On ROCm `torch.cuda.CUDAGraph` is hipGraph; the HIP
operators launch on torch's current stream, so they are captured like any
torch op.  Differences from the reference: the cache key never contains tensor
*values* (graphs.py:197-199 keys 1-element CPU tensors by value, so a CPU
timestep would capture one graph per step); CPU tensors are instead staged into
static device buffers on every call.
"""
from __future__ import annotations

import functools
import logging
import threading
from typing import Any, Callable

import torch

logger = logging.getLogger(__name__)

_pools = {}
_pools_lock = threading.Lock()


def _pool_for(device_index: int):
    with _pools_lock:
        if device_index not in _pools:
            _pools[device_index] = torch.cuda.graph_pool_handle()
        return _pools[device_index]


def signature(obj: Any):
    """Hashable structure of (device, dtype, shape, stride) for tensors; values only for plain scalars."""
    if isinstance(obj, torch.Tensor):
        return ("T", obj.device.type, obj.device.index, obj.dtype, tuple(obj.shape), tuple(obj.stride()))
    if isinstance(obj, (str, int, float, bool, bytes, type(None))):
        return obj
    if isinstance(obj, (tuple, list)):
        return (type(obj).__name__,) + tuple(signature(x) for x in obj)
    if isinstance(obj, dict):
        return ("dict",) + tuple(sorted((k, signature(v)) for k, v in obj.items()))
    return ("obj", id(obj))


def tree_map(fn: Callable, obj: Any):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (tuple, list)):
        return type(obj)(tree_map(fn, x) for x in obj)
    if isinstance(obj, dict):
        return {k: tree_map(fn, v) for k, v in obj.items()}
    return obj


def tree_copy_(dst: Any, src: Any) -> None:
    if isinstance(dst, torch.Tensor):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, (tuple, list)):
        assert len(dst) == len(src)
        for d, s in zip(dst, src):
            tree_copy_(d, s)
    elif isinstance(dst, dict):
        assert dst.keys() == src.keys()
        for k in dst:
            tree_copy_(dst[k], src[k])
    else:
        assert dst == src, "non-tensor argument changed under a cached graph"


def _first_device(obj: Any):
    found = []
    tree_map(lambda t: found.append(t.device) if t.device.type == "cuda" else None, obj)
    return found[0] if found else None


class GraphedCallable:
    """One captured forward: static inputs -> hipGraph -> static outputs."""

    def __init__(self, fn: Callable, args, kwargs, warmup: int = 2, before_replay: Callable = None):
        self.before_replay = before_replay
        dev = _first_device((args, kwargs))
        if dev is None:
            raise RuntimeError("hipGraph capture needs at least one GPU tensor argument")
        self.device = dev
        self.lock = threading.Lock()

        def to_static(t: torch.Tensor):
            if t.device.type == "cuda":
                return torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device).copy_(t)
            return t.to(dev)              # CPU tensors get a static device home
        self.static_in = tree_map(to_static, (args, kwargs))
        with torch.cuda.device(dev):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):      # lazy init / allocator warm-up outside the capture
                    fn(*self.static_in[0], **self.static_in[1])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, pool=_pool_for(dev.index)):
                self.static_out = fn(*self.static_in[0], **self.static_in[1])

    def __call__(self, *args, **kwargs):
        with self.lock:
            if self.before_replay is not None:
                self.before_replay()      # e.g. re-derive weight buffers whose source parameters were updated in place
            tree_copy_(self.static_in, (args, kwargs))
            self.graph.replay()
            return tree_map(torch.clone, self.static_out)


def make_dynamic_graphed_callable(fn: Callable, warmup: int = 2, before_replay: Callable = None) -> Callable:
    lock = threading.Lock()
    cache = {}

    @functools.wraps(fn)
    def dynamic_graphed_callable(*args, **kwargs):
        key = signature((args, kwargs))
        entry = cache.get(key)
        if entry is None:
            with lock:
                entry = cache.get(key)
                if entry is None:
                    logger.info("capturing hipGraph for %s", getattr(fn, "__name__", type(fn).__name__))
                    entry = GraphedCallable(fn, args, kwargs, warmup, before_replay)
                    cache[key] = entry
        return entry(*args, **kwargs)

    dynamic_graphed_callable._cached = cache
    return dynamic_graphed_callable
