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


def bind_rank_threads(local_rank, ranks_on_node):
    """One node, N ranks: give every rank its own contiguous slice of the host cores this process may run on
    (`sched_setaffinity`) and size torch's intra-op pool to it.  A rank of the forward bench runs up to five host threads
    that matter (four lane launchers + the main thread) and the training step a scipy Hungarian assignment per sample: with
    8 ranks sharing one socket unpinned, the launchers of one rank migrate onto cores another rank's assignment is using.
    Returns a description for the bench line (`config.cpu_binding`); a no-op description when the platform has no affinity
    call or a rank's slice would be empty."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return dict(bound=False, reason='no sched_getaffinity on this platform')
    n = max(1, int(ranks_on_node))
    per = len(cores) // n
    if n == 1 or per < 1:
        return dict(bound=False, cores_visible=len(cores), ranks_on_node=n,
                    reason='single rank' if n == 1 else 'fewer cores than ranks')
    mine = cores[local_rank * per:(local_rank + 1) * per]
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(per, 16)))
    return dict(bound=True, cores_visible=len(cores), ranks_on_node=n, cores_per_rank=per, first_core=mine[0], last_core=mine[-1],
                torch_threads=torch.get_num_threads())


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


def sum_over_ranks(value, device=None):
    """Sum of a per-rank number over the process group (bench.py: `ranks_seen` = how many ranks answered)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def throughput(units_per_rank_per_step, steps, elapsed_max, world):
    """Whole-job units/s: every rank processed units_per_rank_per_step * steps units within the slowest
    rank's time."""
    return world * units_per_rank_per_step * steps / elapsed_max


class GradientReducer:
    """Gradient averaging across ranks for the sample-sharded training step (the reference wraps the
    detector in `MMDistributedDataParallel(find_unused_parameters=True)`, tools/train.py + mmdet `train_detector`).

    * few large flat float32 buckets (xGMI rings are per-link bound: fewer, larger messages); parameters are
      bucketed in REVERSE registration order - the order backward produces their gradients;
    * OVERLAP with backward: a post-accumulate-grad hook copies each gradient into its bucket slice and, once a
      bucket is complete, its all-reduce is launched asynchronously while backward continues.  Buckets are launched
      strictly in index order so that every rank issues the same sequence of collectives even when a rank did not
      touch some parameter (that bucket then waits for `finish()` on this rank only);
    * `find_unused_parameters` semantics: a parameter no rank produced a gradient for keeps `grad = None` (AdamW
      then leaves it alone - no weight decay on a frozen-by-construction branch such as the detached heat-map
      head); a parameter unused on THIS rank but used elsewhere contributes zeros.  The used-bitmap (and the
      "a rank accumulated two micro-batches" flag) is reduced (MAX) in ONE collective issued after the last bucket,
      never between buckets: ranks launch different bucket prefixes during backward when a parameter is unused on
      one of them, and any collective slipped in there would pair with a bucket all-reduce on the peer.

        red = GradientReducer(model.parameters(), world)
        loss.backward(); red.finish(); optimizer.step()
    """

    def __init__(self, params, world, bucket_bytes=64 << 20, overlap=True):
        self.world = world
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []                      # [(flat, [(param, offset, numel)])]
        self._where = {}
        self._handles = []
        if world == 1:
            return
        cur, size = [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self._hooks = []
        if overlap:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset()

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, dtype=torch.float32, device=ps[0].device)
        items, o = [], 0
        for p in ps:
            items.append((p, o, p.numel()))
            self._where[id(p)] = (len(self.buckets), o)
            o += p.numel()
        self.buckets.append((flat, items))

    def _reset(self):
        self._filled = [0] * len(self.buckets)
        self._seen = set()
        self._next = 0
        self._handles = []
        self._dirty = False

    def _fill(self, p):
        b, o = self._where[id(p)]
        self.buckets[b][0][o:o + p.numel()].copy_(p.grad.reshape(-1))
        self._seen.add(id(p))
        self._filled[b] += 1

    def _launch_ready(self, force=False):
        while self._next < len(self.buckets):
            flat, items = self.buckets[self._next]
            if self._filled[self._next] < len(items):
                if not force:
                    return
                for p, o, n in items:          # not produced on this rank: zeros
                    if id(p) not in self._seen:
                        flat[o:o + n].zero_()
            self._handles.append(dist.all_reduce(flat, async_op=True))
            self._next += 1

    def _on_grad(self, p):
        if id(p) in self._seen:
            # a second backward before finish() (gradient accumulation): the bucket slice - perhaps already on its way
            # through an all-reduce - holds the first micro-batch only.  finish() then redoes the reduction from the
            # accumulated p.grad of every parameter.
            self._dirty = True
            return
        self._fill(p)
        self._launch_ready()

    def _drain(self, extra):
        """Force the remaining buckets out, then reduce `extra` (MAX) and wait for everything.  Ranks may have launched
        DIFFERENT bucket prefixes during backward (a parameter unused on one rank holds its bucket back there), so a
        collective other than a bucket all-reduce may only be issued once a rank has all of its buckets out: after
        `_launch_ready(force=True)` every rank has issued exactly buckets 0..n-1, in index order."""
        self._launch_ready(force=True)
        h = dist.all_reduce(extra, op=dist.ReduceOp.MAX, async_op=True)
        for w in self._handles:
            w.wait()
        h.wait()
        return extra.tolist()

    def finish(self):
        """Call after backward: completes the outstanding buckets, resolves unused parameters, leaves the averaged
        gradients in `p.grad`.  `p.grad` never aliases a bucket buffer (an in-place all-reduce launched from a hook
        would otherwise rewrite a gradient that a second micro-batch is still accumulating into)."""
        if self.world == 1:
            return
        dev = self.params[0].device
        for p in self.params:                  # no-hook mode, or gradients produced outside the hook path
            if p.grad is not None and id(p) not in self._seen:
                self._fill(p)
        # one vector: the used-parameter bitmap + "some rank accumulated" (decides for all ranks together)
        flags = torch.tensor([1.0 if id(p) in self._seen else 0.0 for p in self.params] +
                             [1.0 if self._dirty else 0.0], dtype=torch.float32, device=dev)
        flags = self._drain(flags)
        used, dirty = flags[:-1], flags[-1]
        if dirty > 0:
            # the reductions above carried first-micro-batch data: redo them from the accumulated p.grad.  Every rank
            # takes this branch (MAX) and issues all buckets again, nothing else in between.
            self._reset()
            for p in self.params:
                if p.grad is not None:
                    self._fill(p)
            self._drain(torch.zeros(1, dtype=torch.float32, device=dev))
        inv = 1.0 / self.world
        torch._foreach_mul_([flat for flat, _ in self.buckets], inv)
        dst, src = [], []
        for p, u in zip(self.params, used):
            if u == 0.0:
                p.grad = None                  # unused on every rank: stays frozen, as under DDP
                continue
            b, o = self._where[id(p)]
            g = self.buckets[b][0][o:o + p.numel()].view_as(p)
            if p.grad is None:                 # unused on this rank only: the other ranks' average
                p.grad = g.to(p.dtype, copy=True)
            else:
                dst.append(p.grad)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        self._reset()

    def remove(self):
        for h in getattr(self, '_hooks', []):
            h.remove()
        self._hooks = []


def allreduce_gradients(params, world, bucket_bytes=64 << 20):
    """One-shot form (after backward, no overlap): average the gradients of `params` across ranks with the
    `GradientReducer` rules - parameters unused on every rank keep `grad = None`."""
    if world == 1:
        return
    GradientReducer(params, world, bucket_bytes, overlap=False).finish()
