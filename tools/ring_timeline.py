"""Phase time line of the ring window-attention kernel (DI_RING_DBG=16 [+ other bits]): per wavefront of workgroup 0, the
shader-clock stamps of: producers 1 = buffer free, 2 = block issued, 3 = a block announced; consumers 4 = starts waiting for
a block, 5 = has it, 6 = released.  Usage: DI_RING_DBG=16 python tools/ring_timeline.py [variant]"""
import ctypes, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import _lib, ops
var = int(sys.argv[1]) if len(sys.argv) > 1 else 26
g = torch.Generator(device='cuda').manual_seed(0)
mk = lambda: torch.randn(6, 128, 112, 200, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
sets = [(mk(), mk(), mk()) for _ in range(3)]
for r in range(4):
    out = ops.local_attention(*sets[r % 3], 9, 9, 1 / math.sqrt(128), variant=var)
torch.cuda.synchronize()
buf = np.zeros(16 * 128, dtype=np.uint64)
_lib.call('di_local_attn_ring_stamps', buf.ctypes.data_as(ctypes.c_void_p), None)
buf = buf.reshape(16, 128)
NS = 96
t0 = min(int(w[0] & 0x00FFFFFFFFFFFFFF) for w in buf if 0 < w[NS - 1] < NS)
names = {1: 'free', 2: 'issued', 3: 'announced', 4: 'wait', 5: 'got', 6: 'released'}
for w in range(16):
    n = int(buf[w, NS - 1])
    if n == 0 or n >= NS:
        continue
    ev = [(int(x >> np.uint64(56)), int(x & np.uint64(0x00FFFFFFFFFFFFFF)) - t0) for x in buf[w, :n]]
    print(f'wave {w} ({n} stamps):')
    line = []
    for tag, t in ev:
        line.append(f'{names.get(tag, tag)}@{t}')
    print('  ' + ' '.join(line))
    if any(tag == 4 for tag, _ in ev):          # consumer: time waiting vs working
        waits = sum(t2 - t1 for (a, t1), (b, t2) in zip(ev, ev[1:]) if a == 4 and b == 5)
        print(f'  waiting {waits} of {ev[-1][1] - ev[0][1]} clocks')
