"""Per-wave phase timestamps of the local-attention measurement builds: rows = sample points, columns = waves.
    python tools/la_ts8.py 7   (second generation, 20 samples per tile)   |   python tools/la_ts8.py 9   (third, 9 per tile)"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops  # noqa: E402
var = int(sys.argv[1]) if len(sys.argv) > 1 else 9
n, C, H, W = 6, 128, 112, 200
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v = (torch.randn(n, C, H, W, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last) for _ in range(3))
for _ in range(3):
    out = ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(C), variant=var)
torch.cuda.synchronize()
raw = out.permute(0, 2, 3, 1).contiguous().view(-1)[:4 * 400].view(torch.int64).cpu().tolist()
ts = [raw[1 + w * 48: 1 + (w + 1) * 48] for w in range(8)]
t0 = min(t[0] for t in ts)
if var == 7:
    per = [f'{u}:{p}' for u in ('K0', 'K1', 'V0', 'V1') for p in ('iss', 'mma', 'mid', 'com', 'bar')]
else:
    per = ['K0:mma', 'K0:bar', 'K1:mma', 'K1:smx', 'K1:bar', 'V0:mma', 'V0:bar', 'V1:mma', 'V1:bar']
names = ['start', 'prolog'] + per * 6
for c in range(min(47, 2 + len(per) * 4)):
    print(f'{names[c]:8s}', ' '.join(f'{ts[w][c] - t0:7d}' for w in range(8)))
