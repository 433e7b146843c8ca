"""Is the several-samples-in-flight step bound by the HOST's graph launches?  CPU time of hipGraphLaunch (CUDAGraph.replay)
against the GPU time of a replay, one launching thread against one thread per lane.
Usage: python tools/launch_cost.py [lanes] [captures per lane]"""
import os, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, parallel, synth
from deepinteraction_amd.graphed import GraphedHotPath

L = int(sys.argv[1]) if len(sys.argv) > 1 else 3
PER = int(sys.argv[2]) if len(sys.argv) > 2 else 1
shape = synth.SHAPE_R
enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16) for i in range(L * PER)]
with torch.no_grad():
    cap = max(range(len(pool)), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
    first = GraphedHotPath(enc, dec, pool[cap])
    recs = [first.prepare(d) for d in pool]
    graphs = [GraphedHotPath(enc, dec, first.record_inputs(r)) for r in recs]
own = [graphs[l::L] for l in range(L)]
lanes = [torch.cuda.Stream() for _ in range(L)]
torch.cuda.synchronize()
print('nodes', graphs[0].num_nodes(), 'lanes', L, 'captures per lane', PER, flush=True)

def replay_only(g):
    g.graph.replay()

# 1. one replay at a time: CPU time of the call, GPU time of the replay
g = graphs[0]
for _ in range(20):
    g(); torch.cuda.synchronize()
cpu = []
for _ in range(50):
    t0 = time.perf_counter(); replay_only(g); t1 = time.perf_counter(); torch.cuda.synchronize()
    cpu.append(t1 - t0)
t0 = time.perf_counter()
for _ in range(50):
    replay_only(g)
tq = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f'one stream: CPU per replay() into an idle queue {sum(cpu) / 50 * 1e3:.3f} ms; 50 back to back: CPU {tq / 50 * 1e3:.3f} ms, wall {tw / 50 * 1e3:.3f} ms per replay', flush=True)

# 2. L lanes, one launching thread
def step(k):
    for o, lane in zip(own, lanes):
        with torch.cuda.stream(lane):
            replay_only(o[k % len(o)])
for k in range(30):
    step(k)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(60):
        step(k)
    tq = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print(f'{L} lanes, one thread: CPU {tq / 60 * 1e3:.3f} ms per step, wall {tw / 60 * 1e3:.3f} ms per step = {L * 60 / tw:.1f} samples/s', flush=True)

# 3. one launching thread per lane
def worker(l, n, bar):
    torch.cuda.set_device(0)
    bar.wait()
    with torch.cuda.stream(lanes[l]):
        for k in range(n):
            replay_only(own[l][k % len(own[l])])
for rep in range(3):
    bar = threading.Barrier(L + 1)
    ths = [threading.Thread(target=worker, args=(l, 60, bar)) for l in range(L)]
    for t in ths:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    tq = time.perf_counter() - t0
    torch.cuda.synchronize()
    tw = time.perf_counter() - t0
    print(f'{L} lanes, one thread per lane: CPU {tq / 60 * 1e3:.3f} ms per step, wall {tw / 60 * 1e3:.3f} ms per step = {L * 60 / tw:.1f} samples/s', flush=True)
