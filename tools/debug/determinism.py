"""Run-to-run bitwise reproducibility of the hot-path ops and modules (GPU)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from deepinteraction_amd import ops, synth
from test_graph_gpu import _to_device
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionEncoder
torch.manual_seed(0)
n, C, H, W = 6, 128, 112, 200
q, k, v = (torch.randn(n, C, H, W, device='cuda').relu().half().contiguous(memory_format=torch.channels_last) for _ in range(3))
for var in (0, 1, 2, 3, 5, 8, 11):
    outs = [ops.local_attention(q, k, v, 9, 9, 1 / math.sqrt(C), variant=var).clone() for _ in range(6)]
    print('local_attention variant', var, 'bitwise equal:', all(torch.equal(outs[0], o) for o in outs[1:]),
          max((outs[0].float() - o.float()).abs().max().item() for o in outs[1:]))
for shape_name in ('SHAPE_TINY', 'SHAPE_R'):
    shape = getattr(synth, shape_name)
    torch.manual_seed(3)
    enc = DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128).cuda().half().eval()
    a = _to_device(synth.make_inputs(1, shape, seed=1), torch.float16)
    res = []
    with torch.no_grad():
        for _ in range(4):
            img, (p0, p1) = enc(a['img_feats'], a['pts_feats'], a['img_metas'], dict(a['pts_metas']))
            res.append((img.clone(), p0.clone(), p1.clone()))
    for nm, idx in (('img', 0), ('pts_conv', 1), ('pts', 2)):
        print(shape_name, 'encoder', nm, 'equal:', all(torch.equal(res[0][idx], r[idx]) for r in res[1:]),
              max((res[0][idx].float() - r[idx].float()).abs().max().item() for r in res[1:]))
    # stage by stage inside layer 0
    with torch.no_grad():
        x = enc.shared_conv_img(ops.cl(a['img_feats']))
        xs = [enc.shared_conv_img(ops.cl(a['img_feats'])) for _ in range(3)]
        print(shape_name, 'shared_conv_img equal:', all(torch.equal(x, y) for y in xs))
        blk = enc.fusion_blocks[0]
        ys = [blk.I_IML(x, x).clone() for _ in range(3)]
        print(shape_name, 'I_IML equal:', all(torch.equal(ys[0], y) for y in ys[1:]))
        qp = [blk.I_IML.query_project(x).clone() for _ in range(3)]
        print(shape_name, 'query_project equal:', all(torch.equal(qp[0], y) for y in qp[1:]))
from deepinteraction_amd.configs import decoder_cfg
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    for bev in (36, 180):
        torch.manual_seed(3)
        dec = DeepInteractionDecoder(**decoder_cfg(bev=bev, num_proposals=50)).cuda().half().eval()
        x = torch.randn(1, 128, bev, bev, device='cuda').relu().half().contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            for _ in range(3): dec.heatmap_head(x)
            hs = [dec.heatmap_head(x).clone() for _ in range(5)]
            c1 = [dec.heatmap_head[0](x).clone() for _ in range(5)]
        print('deterministic flag', det, 'bev', bev, 'heatmap_head equal:', all(torch.equal(hs[0], h) for h in hs[1:]),
              'first conv+bn+relu equal:', all(torch.equal(c1[0], h) for h in c1[1:]))
