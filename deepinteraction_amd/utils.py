"""Small host-side helpers."""


def param_key(module_or_tensors):
    """Cache key that changes whenever a parameter/buffer is updated in place, re-assigned,
    moved or cast (`_version` alone does not see `module.to(...)` / `module.half()`)."""
    if hasattr(module_or_tensors, 'parameters'):
        ts = list(module_or_tensors.parameters()) + list(module_or_tensors.buffers())
    else:
        ts = [t for t in module_or_tensors if t is not None]
    return tuple((t._version, t.dtype, t.device, t.data_ptr()) for t in ts)


_SIDE_STREAMS = {}
# which fork/join sites are active (bit 0: encoder layer sides, 1: the two shared convs, 2: depth chain, 3: decoder
# first map, 4: the BEV -> image warp opens the side stream, 5: RoI side of a decoder block beside its self attention - measured
# slower, off); DI_OVERLAP overrides for A/B measurements
import contextlib as _contextlib
import os as _os
import sys as _sys
import threading as _threading
import types as _types

_OVERLAP_DEFAULT = int(_os.environ.get('DI_OVERLAP', '29'))   # measured (tools/overlap_ab.sh), ms/step: 0: 1.848, 1: 1.800, 5: 1.750, 13: 1.726, 15 (three-way fork): 1.86; later build 13: 1.707, 29: 1.693
_OVERLAP_LOCAL = _threading.local()


class _Module(_types.ModuleType):
    """`utils.OVERLAP` is what every fork / join site reads.  Reading returns the calling THREAD's override (`overlap(mask)`
    context: a lane capture, a measurement) or else the process default; assigning sets the process default.  A capture on
    one host thread (GraphedHotPath(overlap=0)) therefore never changes what a forward on another thread sees."""

    @property
    def OVERLAP(self):
        return getattr(_OVERLAP_LOCAL, 'mask', None) if getattr(_OVERLAP_LOCAL, 'mask', None) is not None else _OVERLAP_DEFAULT

    @OVERLAP.setter
    def OVERLAP(self, mask):
        global _OVERLAP_DEFAULT
        _OVERLAP_DEFAULT = int(mask)


_sys.modules[__name__].__class__ = _Module


@_contextlib.contextmanager
def overlap(mask):
    """The fork / join mask of the forwards THIS thread issues inside the context (None: leave it as it is)."""
    saved = getattr(_OVERLAP_LOCAL, 'mask', None)
    if mask is not None:
        _OVERLAP_LOCAL.mask = int(mask)
    try:
        yield
    finally:
        _OVERLAP_LOCAL.mask = saved


def side_stream(device, i=0):
    """The i-th extra HIP stream of `device` for fork/join branches (created once, reused by every forward)."""
    import torch
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), i)
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return s


def fork_join(device, *thunks):
    """Run independent branches concurrently: thunks[0] on the caller's stream, the others on side streams that are
    forked from it (they wait for everything already queued) and joined back before returning.  Under hipGraph capture
    the branches become parallel paths of the graph.  Contract for the caching allocator: tensors that cross the fork
    or the join must stay referenced by the caller until after the join (then no pool can recycle a block under a
    kernel of another stream); every branch's private temporaries live and die on its own stream."""
    import torch
    thunks = [t for t in thunks if t is not None]
    if len(thunks) == 1:
        return [thunks[0]()]
    main = torch.cuda.current_stream(device)
    sides = [side_stream(device, i) for i in range(len(thunks) - 1)]
    outs = [None] * len(thunks)
    for i, s in enumerate(sides):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs[i + 1] = thunks[i + 1]()
    outs[0] = thunks[0]()
    for s in sides:
        main.wait_stream(s)
    return outs
