"""torch.profiler view of ONE eager mixed-precision training step: the large element-wise / cast operators with their input shapes and
Python call sites (what the rocprofv3 kernel table cannot attribute).  Usage: python tools/train_torch_profile.py [min_us=20]"""
import os, sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import harness, train_step

tr = train_step.Trainer(harness.SHAPES['R'], 200, torch.device('cuda:0'), 1, amp=os.environ.get('AMP', '1') == '1')
for _ in range(3):
    tr.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step()
    torch.cuda.synchronize()
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rows = []
for e in prof.events():
    if e.device_time_total >= thr and e.name.startswith('aten::') and e.name.split('::')[1] in (
            'add', 'add_', 'mul', 'mul_', 'copy_', '_to_copy', 'to', 'sum', 'cat', 'where', 'clamp', 'clamp_min', 'threshold_backward',
            'fill_', 'zero_', 'div', 'sub', 'relu', 'sigmoid', 'native_dropout', 'masked_fill', 'masked_fill_', 'contiguous', 'clone'):
        st = [f for f in (e.stack or []) if 'deepinteraction_amd' in f or 'train_step' in f]
        rows.append((e.device_time_total, e.name, str(e.input_shapes)[:90], (st[0] if st else '')[-110:]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'{len(rows)} element-wise / copy operators of >= {thr} us: {tot / 1e3:.2f} ms')
for t, n, sh, st in rows[:70]:
    print(f'{t:8.1f} us  {n:22s} {sh:90s} {st}')
print()
print('aggregated by (operator, input shapes), device time of the operator itself:')
ka = prof.key_averages(group_by_input_shape=True)
agg = sorted(ka, key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in agg)
print(f'total self device time {tot / 1e3:.2f} ms')
for e in agg[:int(os.environ.get('ROWS', '60'))]:
    print(f'{e.self_device_time_total:9.1f} us  x{e.count:4d}  {e.key[:40]:40s} {str(e.input_shapes)[:110]}')
