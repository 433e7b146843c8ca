"""Which Python lines launch large device copies in one eager forward (torch.profiler with stacks)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import harness, synth
from torch.profiler import profile, ProfilerActivity
shape = synth.SHAPE_R
enc, dec = harness.build_models(shape, 200, torch.float16, 'cuda')
d = harness.to_device(synth.make_inputs(1, shape, seed=0), 'cuda', torch.float16)
with torch.no_grad():
    harness.forward(enc, dec, d); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        harness.forward(enc, dec, d); torch.cuda.synchronize()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::to', 'aten::add', 'aten::mul', 'aten::cat') and e.input_shapes and e.input_shapes[0] and \
            len(e.input_shapes[0]) >= 2 and torch.tensor(e.input_shapes[0]).prod().item() >= 2_000_000:
        st = [s for s in e.stack if 'deepinteraction_amd' in s][:3]
        print(e.name, e.input_shapes[:2], round(e.device_time_total, 1), 'us', ' <- '.join(st))
