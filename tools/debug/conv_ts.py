"""Phase timestamps (shader clock) of workgroup 0 of the image-side 3x3 convolution, chunk 3 (DI_CONV_TS=1 measurement build)."""
import os, sys, torch
os.environ['DI_CONV_TS'] = '1'
PC = os.environ.get('DI_CONV_PC') == '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd import ops
g = torch.Generator(device='cuda').manual_seed(0)
BEV = os.environ.get('SHAPE') == 'bev'
x = torch.randn(*((1, 512, 180, 180) if BEV else (6, 256, 112, 200)), device='cuda', generator=g).relu().half().contiguous(memory_format=torch.channels_last)
conv = torch.nn.Conv2d(512 if BEV else 256, 128, 3, padding=1).cuda().half()
packed = ops.pack_conv3x3(conv.weight, conv.bias)
NW = 12 if PC else 8
names = (['start', 'ky0 issued', 'ky0 barrier', 'ky1 issued', 'ky1 barrier', 'ky2 issued', 'ky2 barrier'] if PC else
         ['start', 'ky0 products + requests issued', 'ky0 barrier', 'ky1 products', 'ky1 barrier', 'ky2 products + halo commit', 'ky2 barrier'])
NST = len(names)
for rep in range(3):
    y = ops.conv3x3(x, *packed)
    torch.cuda.synchronize()
    t = torch.as_strided(y, (y.numel(),), (1,)).view(torch.int64)[:NW * 16].view(NW, 16).cpu()
    if rep < 2:
        continue
    t0 = t[:, 0].min()
    for w in range(NW):
        print('wave', w, ' '.join(f'{int(t[w, k] - t0):6d}' for k in range(NST)))
