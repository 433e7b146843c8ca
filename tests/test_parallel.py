"""N > 1 host logic of the sample-sharded path on CPU: two `gloo` processes (world_size 2)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import time
    import torch.distributed as dist
    from deepinteraction_amd import parallel, synth
    assert parallel.init('gloo')
    # (i) sample assignment: disjoint, contiguous, independent of which rank generates a sample
    ids = [parallel.sample_ids(step, 2, rank, world) for step in range(3)]
    gathered = [None] * world
    dist.all_gather_object(gathered, ids)
    flat = sorted(i for r in gathered for step in r for i in step)
    assert flat == list(range(3 * world * 2)), flat
    shape = synth.SHAPE_TINY
    mine = synth.make_inputs(1, shape, seed=parallel.sample_seed(ids[0][0]))['img_feats']
    other = synth.make_inputs(1, shape, seed=parallel.sample_seed(gathered[1 - rank][0][0]))['img_feats']
    sums = [None] * world
    dist.all_gather_object(sums, float(mine.double().sum()))
    assert abs(sums[1 - rank] - float(other.double().sum())) < 1e-9      # same global sample, same tensor
    assert abs(sums[0] - sums[1]) > 1e-6                                   # different samples on the two ranks
    # (ii) timing protocol: the slow rank sets the time on every rank
    el = parallel.timed_region(lambda: time.sleep(0.02 if rank == 0 else 0.06), steps=3)
    assert 0.17 <= el < 1.0, el
    thr = parallel.throughput(2, 3, el, world)
    assert abs(thr - 12 / el) < 1e-9
    # (iii) bucketed gradient averaging, with a parameter that has no gradient on one rank
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)),
          torch.nn.Parameter(torch.zeros(2, 2), requires_grad=False), torch.nn.Parameter(torch.zeros(4))]
    ps[0].grad = torch.full((5, 3), float(rank + 1))
    ps[1].grad = torch.arange(7.0) * (rank + 1)
    if rank == 0:
        ps[3].grad = torch.ones(4)
    parallel.allreduce_gradients(ps, world, bucket_bytes=64)              # tiny buckets: several flushes
    assert torch.allclose(ps[0].grad, torch.full((5, 3), 1.5))
    assert torch.allclose(ps[1].grad, torch.arange(7.0) * 1.5)
    assert ps[2].grad is None
    assert torch.allclose(ps[3].grad, torch.full((4,), 0.5))
    # (iv) the overlapped form: hooks launch the bucket all-reduces DURING backward; a branch no rank touches
    # keeps grad None (DDP find_unused_parameters semantics - AdamW must not decay it), a branch only rank 0 uses
    # is averaged with zeros
    torch.manual_seed(1)
    net = torch.nn.ModuleDict(dict(a=torch.nn.Linear(6, 8), b=torch.nn.Linear(8, 8), only0=torch.nn.Linear(8, 8),
                                   never=torch.nn.Linear(8, 8), c=torch.nn.Linear(8, 1)))
    red = parallel.GradientReducer(net.parameters(), world, bucket_bytes=256)
    assert len(red.buckets) > 2
    ref = {}
    for it in range(2):                                                    # two steps: state resets between them
        for p in net.parameters():
            p.grad = None
        x = torch.full((3, 6), float(rank + 1 + it))
        h = net['b'](torch.relu(net['a'](x)))
        if rank == 0:
            h = h + net['only0'](h)
        net['c'](h).sum().backward()
        local = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
        red.finish()
        allg = [None] * world
        dist.all_gather_object(allg, local)
        for n, p in net.named_parameters():
            gs = [g[n] for g in allg]
            if all(g is None for g in gs):
                assert p.grad is None, n
                assert n.startswith('never')
            else:
                want = sum(g if g is not None else torch.zeros_like(p) for g in gs) / world
                assert torch.allclose(p.grad, want, atol=1e-6), n
        assert net['only0'].weight.grad is not None and net['never'].weight.grad is None
    # (v) gradient accumulation: two backward calls per finish() - the hooks fire twice, the reduction must carry the SUM
    for p in net.parameters():
        p.grad = None
    for micro in range(2):
        x = torch.full((3, 6), float(rank + 1 + micro))
        h = net['b'](torch.relu(net['a'](x)))
        if rank == 0 and micro == 1:
            h = h + net['only0'](h)
        net['c'](h).sum().backward()
    local = {n: (None if p.grad is None else p.grad.clone()) for n, p in net.named_parameters()}
    red.finish()
    allg = [None] * world
    dist.all_gather_object(allg, local)
    for n, p in net.named_parameters():
        gs = [g[n] for g in allg]
        if all(g is None for g in gs):
            assert p.grad is None, n
        else:
            want = sum(g if g is not None else torch.zeros_like(p) for g in gs) / world
            assert torch.allclose(p.grad, want, atol=1e-6), ('accumulation', n)
    # (vi) ranks launch DIFFERENT bucket prefixes during backward: the rank-0-only branch is registered last, so it
    # sits in bucket 0 - rank 0 launches every bucket from its hooks, rank 1 none before finish().  No collective may
    # be issued between them (a flag all-reduce in front of the forced buckets paired with a bucket
    # all-reduce on the peer).  With and without accumulation, and with gradients zeroed IN PLACE between steps
    # (p.grad must not alias the bucket an in-flight all-reduce rewrites).
    torch.manual_seed(2)
    net2 = torch.nn.ModuleDict(dict(a=torch.nn.Linear(6, 8), b=torch.nn.Linear(8, 8), c=torch.nn.Linear(8, 1),
                                    only0=torch.nn.Linear(8, 8)))
    red2 = parallel.GradientReducer(net2.parameters(), world, bucket_bytes=32)
    assert len(red2.buckets) >= 6
    assert red2._where[id(net2['only0'].bias)][0] == 0
    launched = []
    for case, (micros, in_place_zero) in enumerate([(1, False), (2, False), (1, True), (2, True), (1, False)]):
        for p in net2.parameters():
            if in_place_zero and p.grad is not None:
                p.grad.zero_()
            else:
                p.grad = None
        local = {n: None for n, _ in net2.named_parameters()}
        for micro in range(micros):
            x = torch.full((3, 6), float(rank + 1 + micro + case))
            h = net2['b'](torch.relu(net2['a'](x)))
            if rank == 0:
                h = h + net2['only0'](h)
            before = {n: (None if p.grad is None else p.grad.clone()) for n, p in net2.named_parameters()}
            net2['c'](h).sum().backward()
            if micro == 0:
                launched.append(red2._next)
            for n, p in net2.named_parameters():       # this rank's own contribution, independent of the reducer
                if p.grad is None:
                    continue
                d = p.grad.clone() if before[n] is None else p.grad - before[n]
                local[n] = d if local[n] is None else local[n] + d
        red2.finish()
        allg = [None] * world
        dist.all_gather_object(allg, local)
        for n, p in net2.named_parameters():
            gs = [g[n] for g in allg]
            want = sum(g if g is not None else torch.zeros_like(p) for g in gs) / world
            assert torch.allclose(p.grad, want, atol=1e-5), ('prefix', case, n, p.grad, want)
            assert not any(p.grad.data_ptr() == flat.data_ptr() or
                           flat.data_ptr() < p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4
                           for flat, _ in red2.buckets), n
    # the scenario really happened: rank 0 had buckets out during backward, rank 1 none
    assert (launched[0] > 0) == (rank == 0), launched
    out.put((rank, el))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, p.exitcode
    res = dict(out.get(timeout=5) for _ in range(2))
    assert abs(res[0] - res[1]) < 1e-12            # MAX over ranks: identical on both


def test_single_process_defaults():
    sys.path.insert(0, ROOT)
    from deepinteraction_amd import parallel
    assert parallel.sample_ids(2, 3, 0, 1) == [6, 7, 8]
    assert parallel.max_over_ranks(1.25) == 1.25
    assert parallel.throughput(2, 10, 4.0, 1) == 5.0


def test_bench_self_launches_n_ranks():
    """`python bench.py --gpus 2` with no torchrun environment re-launches itself as one process per rank
    (reference launch contract: one command per node, tools/dist_train.sh:7-9).  --dry-run keeps it on CPU / gloo."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run', '--steps', '4'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout                                     # rank 0 prints ONE json line
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['ranks_seen'] == 2 and rec['steps'] == 4
    # the slowest rank (rank 1 sleeps 4 ms per step) sets the time
    assert rec['ms_per_step'] >= 3.9


def test_bench_dry_run_train_wires_the_reducer_on_the_real_parameter_list():
    """`bench.py --gpus 2 --mode train --dry-run`: two gloo ranks, the two hot-path modules built on
    the CPU, the gradient reducer bucketed over their REAL parameter list (414 tensors, 91.6 MB: two 64-MB buckets), hooks
    fired by a synthetic backward of the real shapes; the image RoI blocks are unused on rank 1 (a sample whose views hold
    <= 1 query each: reference decoder_utils.py:726 under find_unused_parameters=True, Fusion_0075_refactor.py:277), the
    detached heat-map head on every rank.  The ranks agree exactly; every rank is pinned to its own slice of the cores."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--mode', 'train', '--dry-run', '--steps', '2'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    cfg = rec['config']
    assert cfg['ranks_seen'] == 2 and cfg['parameters'] > 400 and cfg['gradient_bytes'] > 80e6 and cfg['buckets'] >= 2
    assert cfg['max_abs_error'] < 1e-5
    b = cfg['cpu_binding']
    assert b['ranks_on_node'] == 2 and (not b['bound'] or b['cores_per_rank'] * 2 <= b['cores_visible'])
