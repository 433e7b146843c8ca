"""Small host-side helpers."""


def param_key(module_or_tensors):
    """Cache key that changes whenever a parameter/buffer is updated in place, re-assigned,
    moved or cast (`_version` alone does not see `module.to(...)` / `module.half()`)."""
    if hasattr(module_or_tensors, 'parameters'):
        ts = list(module_or_tensors.parameters()) + list(module_or_tensors.buffers())
    else:
        ts = [t for t in module_or_tensors if t is not None]
    return tuple((t._version, t.dtype, t.device, t.data_ptr()) for t in ts)
