"""Debug (GPU): the product encoder at shape R, float32, eval mode - forward with gradients enabled (autograd ops, library
GEMMs) against its own no-grad forward (fused kernels), per view; and the product's completed depth against the oracle's."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionEncoder
from deepinteraction_amd.mmdet3d_plugin.models.utils import encoder_utils as eu
from oracle.refpin import make_golden as mg, make_golden_grad_shapeR as gg
from oracle import encoder as oenc
DEV = 'cuda'
torch.backends.cudnn.deterministic = True
inp = gg.case()
M = DeepInteractionEncoder(num_layers=2, in_channels_img=gg.SHAPE['c_img'], in_channels_pts=gg.SHAPE['c_pts'], hidden_channel=128)
mg.randomize(M, gg.SEED_WEIGHTS)
M = M.to(DEV).eval()
def pmd():
    pm = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(DEV) for p in inp['pts_metas']['pts']]
    return pm
img = inp['img_feats'].to(DEV); pts = inp['pts_feats'].to(DEV)
with torch.no_grad():
    a_im, (a_p0, a_p1) = M(img, pts, inp['img_metas'], pmd())
b_im, (b_p0, b_p1) = M(img.clone().requires_grad_(True), pts.clone().requires_grad_(True), inp['img_metas'], pmd())
for name, a, b in (('img', a_im, b_im), ('pts_conv', a_p0, b_p0), ('pts', a_p1, b_p1)):
    d = (a.float() - b.float()).abs()
    print(f'no-grad vs grad forward {name}: max {d.max().item():.3e} mean {d.mean().item():.3e} scale {a.abs().max().item():.2f}',
          'per view max', [f'{d[v].max().item():.1e}' for v in range(d.shape[0])])
gold = np.load(os.path.join(os.path.dirname(gg.OUT), 'golden', 'grad_shapeR.npz')) if False else np.load(os.path.join(gg.OUT, 'grad_shapeR.npz'))
for tag, outs in (('no-grad', (a_im, a_p0, a_p1)), ('grad', (b_im, b_p0, b_p1))):
    for name, t in zip(('out_img', 'out_pts_conv', 'out_pts'), outs):
        s, _, _ = gg.sample(t)
        d = np.abs(s - gold[name + '.sample']); sc = max(1.0, float(gold[name + '.absmax']))
        print(f'{tag} vs reference golden {name}: median {np.median(d)/sc:.2e} p99 {np.quantile(d,0.99)/sc:.2e} p999 {np.quantile(d,0.999)/sc:.2e} max {d.max()/sc:.2e} frac>2e-4 {(d>2e-4*sc).mean():.4f}')
# depth: product chain vs oracle
from deepinteraction_amd.geometry import SampleGeometry
Hi, Wi = gg.SHAPE['img_hw']
geom = SampleGeometry(inp['img_metas'][0], (Hi, Wi), DEV)
dd = eu.BEVWarp.dense_depth(geom, inp['pts_metas']['pts'][0].to(DEV), Hi, Wi).cpu()
l2i = oenc.lidar2img_tensor(inp['img_metas'], inp['img_feats'])
sp = oenc.BEVWarp().sparse_depth(inp['pts_metas']['pts'][0], inp['img_metas'][0], l2i[0], Hi, Wi)
od = oenc.complete_depth(sp)
df = (dd - od).abs()
print('dense depth product vs oracle: differing pixels', int((df > 1e-4).sum()), 'of', df.numel(), 'max', df.max().item(),
      'per view', [int((df[v] > 1e-4).sum()) for v in range(6)])
sps = geom.sparse_depth.cpu() if hasattr(geom, 'sparse_depth') else None
if sps is not None:
    print('sparse depth differing', int(((sps - sp).abs() > 0).sum()))
