"""Generate the golden vectors committed under tests/golden/ by running the REFERENCE'S OWN
code (TEST INFRASTRUCTURE; run in the build container where /root/reference exists):

    python -m oracle.refpin.make_golden

The reference Python is imported unmodified (oracle/refpin/__init__.py); its locatt_ops binds to
the reference kernels.cuh compiled for the host.  Inputs and module weights are regenerated from
seeds by `golden_cases()` (shared with tests/test_golden.py), so only outputs are stored.
"""
import os

import numpy as np
import torch

from deepinteraction_amd import synth
from oracle import configs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden')


def randomize(module, seed):
    """Overwrite EVERY parameter and BN buffer deterministically, in name order, so that the
    reference class and the oracle class hold identical weights regardless of how their
    constructors consume the RNG."""
    g = torch.Generator().manual_seed(seed)
    sd = module.state_dict()
    for name in sorted(sd):
        t = sd[name]
        if not t.is_floating_point():
            continue
        if name.endswith('running_var'):
            v = torch.rand(t.shape, generator=g) + 0.5
        elif name.endswith('running_mean'):
            v = torch.randn(t.shape, generator=g) * 0.1
        elif t.dim() >= 2:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif name.endswith('bias'):
            v = torch.randn(t.shape, generator=g) * 0.05
        else:                                   # norm scales
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        t.copy_(v)


def locatt_inputs():
    g = torch.Generator().manual_seed(11)
    a, b = torch.randn(2, 16, 9, 11, generator=g), torch.randn(2, 16, 9, 11, generator=g)
    w = torch.randn(2, 9, 11, 81, generator=g)
    return a, b, w


def sparse_depth_input():
    from oracle import encoder as oenc
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=2)
    sparse = oenc.BEVWarp().sparse_depth(inp['pts_metas']['pts'][0], inp['img_metas'][0],
                                         oenc.lidar2img_tensor(inp['img_metas'], torch.zeros(1))[0],
                                         *shape['img_hw'])
    rng = np.random.default_rng(0)
    sparse = sparse.numpy().copy()
    sparse[rng.random(sparse.shape) < 0.6] = 0
    return sparse


def encoder_case(cls_encoder, aug):
    shape = synth.SHAPE_TINY
    inp = synth.make_inputs(1, shape, seed=0, aug=synth.example_aug(0) if aug else None)
    torch.manual_seed(1234)
    m = cls_encoder(num_layers=2, in_channels_img=shape['c_img'], in_channels_pts=shape['c_pts'],
                    hidden_channel=128)
    randomize(m, 5)
    return m.eval(), inp


def decoder_case(cls_decoder):
    shape = synth.SHAPE_TINY
    cfg = configs.decoder_cfg(bev=36, num_proposals=40)
    torch.manual_seed(7)
    m = cls_decoder(**cfg)
    randomize(m, 9)
    g = torch.Generator().manual_seed(0)
    Hi, Wi = shape['img_hw']
    p0, p1 = torch.randn(2, 128, 36, 36, generator=g), torch.randn(2, 128, 36, 36, generator=g)
    img = torch.randn(12, 128, Hi, Wi, generator=g)
    metas = synth.make_inputs(2, shape, seed=0)['img_metas']
    return m.eval(), ([p0, p1], img, metas)


def encoder_pp_case(cls_encoder, aug, fix=lambda m: m):
    shape = synth.SHAPE_PP_TINY
    inp = synth.make_inputs_pp(1, shape, seed=0, aug=synth.example_aug(1) if aug else None)
    torch.manual_seed(1234)
    m = fix(cls_encoder(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'])))
    randomize(m, 15)
    return m.eval(), inp


def decoder_pp_case(cls_decoder):
    m, args = decoder_case(cls_decoder)
    randomize(m, 19)
    return m, args


def summarize(t):
    """A strided sample plus moments: small fixture, still position-sensitive."""
    t = t.detach().float()
    flat = t.reshape(-1)
    idx = torch.arange(0, flat.numel(), max(1, flat.numel() // 4096))
    pos = torch.arange(flat.numel(), dtype=torch.float64)
    return dict(sample=flat[idx].numpy(), sum=np.float64(flat.double().sum()),
                abssum=np.float64(flat.double().abs().sum()),
                possum=np.float64((flat.double() * torch.cos(pos * 0.37)).sum()))


def main():
    from oracle.refpin import load_reference
    ref = load_reference('reference')
    os.makedirs(OUT, exist_ok=True)
    la = ref.stubs.make_locatt_module('reference').localattention
    a, b, w = locatt_inputs()
    np.savez_compressed(os.path.join(OUT, 'locatt.npz'),
                        similar_forward=la.similar_forward(a, b, 9, 9).numpy(),
                        weighting_forward=la.weighting_forward(a, w, 9, 9).numpy(),
                        similar_backward_ori=la.similar_backward(a, w, 9, 9, True).numpy(),
                        similar_backward_loc=la.similar_backward(a, w, 9, 9, False).numpy(),
                        weighting_backward_ori=la.weighting_backward_ori(w, a, 9, 9).numpy(),
                        weighting_backward_weight=la.weighting_backward_weight(a, b, 9, 9).numpy())
    sparse = sparse_depth_input()
    dense = np.stack([ref.depth_map_utils.fill_in_multiscale(d.copy(), extrapolate=False,
                                                             blur_type='bilateral')[0] for d in sparse])
    np.savez_compressed(os.path.join(OUT, 'depth_completion.npz'), dense=dense)
    out = {}
    with torch.no_grad():
        for aug in (False, True):
            m, inp = encoder_case(ref.encoder.DeepInteractionEncoder, aug)
            img, (p0, p1) = m(inp['img_feats'], inp['pts_feats'], inp['img_metas'], inp['pts_metas'])
            for name, t in (('img', img), ('pts_conv', p0), ('pts', p1)):
                for k, v in summarize(t).items():
                    out[f'enc{int(aug)}_{name}_{k}'] = v
        m, (pts, img, metas) = decoder_case(ref.decoder.DeepInteractionDecoder)
        r = m(pts, img, metas)[0][0]
        for k, v in r.items():
            out[f'dec_{k}'] = v.numpy()
        out['dec_query_labels'] = m.query_labels.numpy()
        out['dec_on_the_image_mask'] = torch.stack(m.on_the_image_mask).numpy()
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **out)
    out = {}
    with torch.no_grad():                              # DeepInteraction++
        for aug in (False, True):
            m, inp = encoder_pp_case(ref.encoder_pp.FusionTransformerv4, aug, ref.fix_zero_layer_encoders)
            img, (p0, p1) = m(list(inp['img_feats']), list(inp['pts_feats']), inp['img_metas'], inp['pts_metas'])
            for name, t in (('img', img), ('pts_conv', p0), ('pts', p1)):
                for k, v in summarize(t).items():
                    out[f'enc{int(aug)}_{name}_{k}'] = v
        m, (pts, img, metas) = decoder_pp_case(ref.decoder_pp.DeepInteractionPlusPlusDecoder)
        r = m(pts, img, metas)[0][0]
        for k, v in r.items():
            out[f'dec_{k}'] = v.numpy()
        out['dec_query_labels'] = m.query_labels.numpy()
        out['dec_on_the_image_mask'] = torch.stack(m.on_the_image_mask).numpy()
    np.savez_compressed(os.path.join(OUT, 'modules_pp.npz'), **out)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
