"""Only the steady-state step of bench.py's default line (4 single-stream captures in flight, one launching thread per lane,
resident hand-over) - for `rocprofv3 --kernel-trace` + tools/trace_overlap.py.  Usage: python tools/lanes_steps.py [lanes=4] [steps=60]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from deepinteraction_amd import harness, parallel, synth
from deepinteraction_amd.graphed import GraphedHotPath
L = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
shape = synth.SHAPE_R
enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
pool = [harness.to_device(synth.make_inputs(1, shape, seed=parallel.sample_seed(i)), 'cuda', torch.float16) for i in range(2 * L)]
with torch.no_grad():
    cap = max(range(len(pool)), key=lambda i: int(pool[i]['pts_metas']['pillars'].shape[0]))
    step, _, _, graphs, _, _ = bench.graphed_steps(lambda inp, ov: GraphedHotPath(enc, dec, inp, overlap=ov), pool, cap, L, True)
    for _ in range(40):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f'{L} lanes: {dt / N * 1e3:.3f} ms per step = {L * N / dt:.1f} samples/s', flush=True)
