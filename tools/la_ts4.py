"""Per-wave phase timestamps of the vertical-streaming local-attention kernel (measurement build, variant 26)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops  # noqa: E402
n, C, H, W = 6, 128, 112, 200
g = torch.Generator(device='cuda').manual_seed(0)
q, k, v = (torch.randn(n, C, H, W, device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last) for _ in range(3))
for _ in range(3):
    out = ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(C), variant=ops.LA_MFMA4 + 15)
torch.cuda.synchronize()
raw = out.permute(0, 2, 3, 1).contiguous().view(-1)[:4 * 300].view(torch.int64).cpu().tolist()
ts = [raw[1 + w * 64: 1 + (w + 1) * 64] for w in range(4)]
t0 = min(t[0] for t in ts)
per = ['S:pair0', 'S:pair2', 'S:mma', 'softmax', 'barK', 'commitK', 'O:mma', 'barV', 'commitV']
names = ['start', 'prolog'] + per * 9
prev = [t[0] for t in ts]
for c in range(2 + 9 * 4):
    if ts[0][c] == 0:
        break
    print(f'{names[c]:8s}', ' '.join(f'{ts[w][c] - t0:7d} (+{ts[w][c] - (ts[w][c-1] if c else ts[w][0]):5d})' for w in range(4)))
