"""One process per GPU, sharded by sample.

Forward inference of the interaction path has no data-path collective: every rank owns its samples
and runs its own replica.  What ranks share is (i) the rule that assigns samples to ranks, (ii) the
timing protocol of bench.py - barrier, timed region, barrier, MAX over ranks - and (iii) for the
training step the gradient all-reduce (RCCL through `torch.distributed` backend "nccl" on GPUs).
These helpers are backend-agnostic so the N > 1 logic is covered by world_size-2 `gloo` tests on CPU
(tests/test_parallel.py).
"""
import os
import time

import torch
import torch.distributed as dist


def env_rank():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


def init(backend, device=None):
    """Join the process group described by MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE.
    backend "nccl" (= RCCL on ROCm) needs `device` (a cuda device) - it is bound at init so the
    first collective does not guess; "gloo" runs on CPU."""
    rank, _, world = env_rank()
    if world == 1:
        return False
    kw = {}
    if backend == 'nccl':
        assert device is not None
        kw['device_id'] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return True


def sample_ids(step, batch_per_rank, rank, world):
    """Global sample indices rank `rank` owns at `step` (weak scaling: the global batch is
    world * batch_per_rank consecutive samples, dealt to ranks in contiguous blocks)."""
    base = (step * world + rank) * batch_per_rank
    return list(range(base, base + batch_per_rank))


def sample_seed(sample_id):
    """Seed of synthetic sample `sample_id` (independent of world size: the same global sample is
    the same tensor no matter which rank generates it)."""
    return 1000 + sample_id


def barrier(device=None):
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    if device is not None and device.type == 'cuda':
        torch.cuda.synchronize(device)


def timed_region(fn, steps, device=None):
    """barrier + sync | `steps` calls of fn | barrier + sync; returns the MAX elapsed seconds over ranks."""
    barrier(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier(device)
    elapsed = time.perf_counter() - t0
    return max_over_ranks(elapsed, device)


def max_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def throughput(units_per_rank_per_step, steps, elapsed_max, world):
    """Whole-job units/s: every rank processed units_per_rank_per_step * steps units within the slowest
    rank's time."""
    return world * units_per_rank_per_step * steps / elapsed_max


def allreduce_gradients(params, world, bucket_bytes=64 << 20):
    """Average gradients across ranks in few large flat buckets (xGMI rings are per-link bound: fewer,
    larger messages).  Parameters without a gradient (e.g. RoI-block branches skipped in this
    iteration, SURVEY 2.2c `find_unused_parameters`) contribute zeros so every rank reduces the same
    layout."""
    if world == 1:
        return
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in bucket])
        dist.all_reduce(flat)
        flat /= world
        o = 0
        for p in bucket:
            n = p.numel()
            g = flat[o:o + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g
            else:
                p.grad.copy_(g)
            o += n
        bucket, size = [], 0
    for p in params:
        if not p.requires_grad:
            continue
        bucket.append(p)
        size += p.numel() * 4
        if size >= bucket_bytes:
            flush()
    flush()
