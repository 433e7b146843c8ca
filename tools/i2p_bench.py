"""Pillar attention at the benched shape: key-table build and attention pass, HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops, synth  # noqa: E402
from deepinteraction_amd.geometry import SampleGeometry  # noqa: E402
from tools.kernel_bench import timeit  # noqa: E402

shape = synth.SHAPE_R
Hi, Wi = shape['img_hw']
Hb, Wb = shape['bev_hw']
C, dev = 128, 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*sh):
    return torch.randn(*sh, device=dev, generator=g).relu().half().contiguous(memory_format=torch.channels_last)


inp = synth.make_inputs(1, shape, seed=0)
geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), dev)
pm = inp['pts_metas']
pil, coo, num = pm['pillars'].to(dev), pm['pillar_coors'].to(dev), pm['pillars_num_points'].to(dev)
img, bev = rnd(6, C, Hi, Wi), rnd(1, C, Hb, Wb) * 0.2
args = (pil, coo, num, geom.lidar2img, geom.aug_rev, geom.ori_hw)
us = timeit(lambda: ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb)))
print(f'i2p_key_table  (P={pil.shape[0]}): {us:7.1f} us')
keys = ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb))
cnt = keys.table[:Hb * Wb * 4].view(torch.int32)
nk = int(cnt.sum())
so = not os.environ.get('DI_I2P_NO_ORDER')
print('sector order' if so else 'row-major order (XCD stripes)')
byt = 6 * C * Hi * Wi * 2 + nk * 16 + 2 * C * Hb * Wb * 2
for name, k in (('matrix-core (dense stream)', keys), ('wave per cell', ops.i2p_key_table(*args, (Hi, Wi), (Hb, Wb), dense=False))):
    if name.startswith('matrix') and k.dense is None:
        continue
    us = timeit(lambda: ops.i2p_attention(img, bev, *args, keys=k, sector_order=so), iters=50)
    print(f'i2p_attention {name} ({nk} keys, {int((cnt > 0).sum())} cells): {us:7.1f} us   algorithmic {byt / 1e6:.1f} MB -> '
          f'{byt / us / 1e6:.3f} TB/s; gathered rows {nk * 4 * 256 / 1e6:.0f} MB -> {nk * 4 * 256 / us / 1e6:.2f} TB/s')
