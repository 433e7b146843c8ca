import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_graph_gpu import _to_device, _eager
from deepinteraction_amd import synth
from deepinteraction_amd.configs import decoder_cfg
from deepinteraction_amd.graphed import GraphedHotPath
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder
shape = synth.SHAPE_TINY
torch.manual_seed(3)
enc = DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128).cuda().half().eval()
dec = DeepInteractionDecoder(**decoder_cfg(bev=shape['bev_hw'][0], num_proposals=50)).cuda().half().eval()
a = _to_device(synth.make_inputs(1, shape, seed=1), torch.float16)
small = dict(shape, n_points=shape['n_points'] // 2)
b = _to_device(synth.make_inputs(1, small, seed=2), torch.float16)
def cmp(x, y, tag):
    for key in x:
        d = (x[key].float() - y[key].float()).abs()
        print(tag, key, tuple(x[key].shape), 'max', d.max().item(), 'n>1e-2', int((d > 1e-2).sum()))
for _ in range(3): _eager(enc, dec, a); _eager(enc, dec, b)
r1 = {k: v.clone() for k, v in _eager(enc, dec, a).items()}
r2 = {k: v.clone() for k, v in _eager(enc, dec, a).items()}
cmp(r1, r2, 'eager-eager')
rb = {k: v.clone() for k, v in _eager(enc, dec, b).items()}
g = GraphedHotPath(enc, dec, a)
o = {k: v.clone() for k, v in g()[0][0].items()}
cmp(o, r1, 'graph-eager A')
o = {k: v.clone() for k, v in g()[0][0].items()}
cmp(o, r1, 'graph2-eager A')
g.load(b)
o = {k: v.clone() for k, v in g()[0][0].items()}
cmp(o, rb, 'graph-eager B')
