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
import os as _os
OVERLAP = int(_os.environ.get('DI_OVERLAP', '29'))   # measured (tools/overlap_ab.sh), ms/step: 0: 1.848, 1: 1.800, 5: 1.750, 13: 1.726, 15 (three-way fork): 1.86; later build 13: 1.707, 29: 1.693


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
