"""Cache of device-side groupings (``het_grouping``) keyed by the identity of the
index tensors they were built from.

The reference ops receive bare index tensors on every call; their atomics-based
kernels need no preprocessing.  Our segmented-reduction kernels want the edge
list grouped by destination / by (relation, node).  That grouping is a pure
function of the index tensors, so it is built once (on the device, by
``het_grouping_create``) and looked up by tensor identity afterwards.  An entry
holds strong references to its source tensors, which pins their storage: a
``data_ptr`` can therefore never be recycled for different contents while the
entry is alive, and ``_version`` catches in-place edits.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import sys
import threading
from collections import OrderedDict
from typing import Optional

import torch

from . import _lib

_MAX_ENTRIES = 32
_cache: "OrderedDict[tuple, Grouping]" = OrderedDict()
_cache_lock = threading.RLock()  # lookups, inserts and evictions of _cache (model threads, the autograd engine's workers)
enabled = True  # process-wide default; tests flip this to exercise the atomics kernels
_tls = threading.local()  # .force: this thread's override of `enabled` (forced(), sampling.one_shot_graphs)


def is_enabled() -> bool:
    """Whether the ops called from this thread use the cached groupings: the thread's override if one is set (``forced``),
    else the process-wide default ``enabled``."""
    f = getattr(_tls, "force", None)
    return enabled if f is None else f


@contextlib.contextmanager
def forced(value: Optional[bool]):
    """Run a block with groupings on / off on THIS thread only (None: no override).  Other threads -- a concurrent model, the
    autograd engine's workers -- are not affected; an autograd node keeps the choice of its forward for its backward
    through ``consistent`` below, whichever thread runs it."""
    old = getattr(_tls, "force", None)
    _tls.force = value
    try:
        yield
    finally:
        _tls.force = old


def consistent(cls):
    """Class decorator for torch.autograd.Function subclasses: backward runs with the groupings on / off as they were
    when forward ran (kernel selection in the two passes has to match: several backward passes reuse what the grouped
    forward built), wherever and on whatever thread autograd calls it."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args, **kwargs):
        ctx._het_plan_on = is_enabled()
        return fwd(ctx, *args, **kwargs)

    def backward(ctx, *grads):
        with forced(getattr(ctx, "_het_plan_on", None)):
            return bwd(ctx, *grads)

    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


class Grouping:
    def __init__(self, handle: C.c_void_p, keep):
        self.handle = handle
        self._keep = keep  # source tensors, kept alive with the handle

    @property
    def num_segments(self) -> int:
        return int(_lib.lib().het_grouping_num_segments(self.handle))

    @property
    def nbytes(self) -> int:
        """Device bytes the grouping holds (inside torch's allocator statistics when _lib.use_torch_allocator() is in effect --
        the default of het_amd.kernels -- else hipMalloc'ed by the library and invisible to them)."""
        return int(_lib.lib().het_grouping_bytes(self.handle)) if self.handle else 0

    def __del__(self):
        # (the library may already be unloaded at interpreter exit: nothing left to free then; and with torch's allocator
        #  installed a destroy calls back into Python -- not from a finalizing interpreter)
        handle, self.handle = self.handle, None
        lib = getattr(_lib, "_lib", None) if _lib is not None else None  # (module globals are cleared at shutdown)
        if handle and lib is not None and sys is not None and not sys.is_finalizing():
            lib.het_grouping_destroy(handle)


def _ident(t: Optional[torch.Tensor]):
    return None if t is None else (t.data_ptr(), t.numel(), t._version)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def get_grouping(rel_ptrs: Optional[torch.Tensor], keys: torch.Tensor, key_bound: int,
                 payload0: Optional[torch.Tensor] = None, payload1: Optional[torch.Tensor] = None) -> Optional[Grouping]:
    """Grouping of the positions of ``keys`` by (relation, key) -- by key alone when
    ``rel_ptrs`` is None.  Returns None when groupings are disabled."""
    if not is_enabled():
        return None
    k = (_ident(rel_ptrs), _ident(keys), int(key_bound), _ident(payload0), _ident(payload1), keys.device.index)
    # One lock around lookup, build and eviction: two threads asking for the same grouping get the same object, and an entry is
    # never evicted between its lookup and its return.  (The caller keeps the returned Grouping alive while it uses it: an
    # eviction by another thread only drops the cache's reference.)
    with _cache_lock:
        g = _cache.get(k)
        if g is not None:
            _cache.move_to_end(k)
            # (the stream this op will read the grouping on: het_grouping_destroy orders the release after it, include/het_amd.h)
            if _lib.has("het_grouping_note_stream"):
                _lib.lib().het_grouping_note_stream(g.handle, C.c_void_p(torch.cuda.current_stream(keys.device).cuda_stream))
            return g
        for t in (rel_ptrs, keys, payload0, payload1):
            if t is not None and not (t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
                raise _lib.HetError("groupings need contiguous int64 tensors on the GPU")
        out = C.c_void_p()
        stream = C.c_void_p(torch.cuda.current_stream(keys.device).cuda_stream)
        args = (_ptr(rel_ptrs), 0 if rel_ptrs is None else rel_ptrs.numel() - 1, _ptr(keys), keys.numel(), int(key_bound),
                _ptr(payload0), _ptr(payload1), stream, C.byref(out))
        with torch.cuda.device(keys.device):
            try:
                _lib.call("het_grouping_create", *args)
            except _lib.HetError:
                # out of device memory while building: the cached groupings of other graphs are the first thing to give back
                # (they live in torch's allocator when it is installed: het_amd/_lib.py use_torch_allocator), then once more
                if not _cache:
                    raise
                _cache.clear()
                torch.cuda.empty_cache()
                _lib.call("het_grouping_create", *args)
        g = Grouping(out, (rel_ptrs, keys, payload0, payload1))
        _cache[k] = g
        while len(_cache) > _MAX_ENTRIES:
            _cache.popitem(last=False)
        return g


def cached_bytes() -> int:
    """Device bytes of all cached groupings (part of torch.cuda.memory_allocated when torch's allocator is installed,
    _lib.allocator_is_external(); otherwise add them to it for a true footprint)."""
    with _cache_lock:
        return sum(g.nbytes for g in _cache.values())


def clear():
    with _cache_lock:
        _cache.clear()
