"""More golden GRADIENT vectors at the BENCHED shapes, produced by the REFERENCE'S OWN Python imported unmodified from
/root/reference (TEST INFRASTRUCTURE; run in the build container, one case per invocation):

    python -m oracle.refpin.make_golden_grad_more enc_train     # tests/golden/grad_shapeR_train.npz
    python -m oracle.refpin.make_golden_grad_more head          # tests/golden/grad_head_shapeR.npz
    python -m oracle.refpin.make_golden_grad_more pp_neck       # tests/golden/grad_pp_neck.npz

  * `enc_train`: the MMRI encoder of Fusion_0075_refactor at shape R in TRAIN mode - BatchNorm batch statistics and their
    backward at full size, the attention dropout of the pillar attention switched off on both sides (its mask is drawn from
    each side's own generator).  Same seeded inputs / weights / functional as the eval-mode file (make_golden_grad_shapeR.py).
    What the float32 AND the mixed-precision (`--amp`) training forward / backward of the product are compared with.
  * `head`: the MMPI decoder (`DeepInteractionDecoder`, Q = 200, 1 decoder layer + 4 RoI layers) at shape R in train mode with
    dropout 0, on seeded feature maps, DEFAULT initialisation (drawn once by the ORACLE's constructor under a fixed seed and
    loaded into the reference class - the test loads the same state into the product class; randomised N(0, 1/fan_in) weights
    make the head's gradients ill-conditioned, tests/test_plusplus_training_gpu.py::test_pp_head_gradients_match_oracle) with
    the RoI blocks' residual branches conditioned as in a trained head (`head_prepare`):
    gradients w.r.t. the three feature maps and every parameter, the proposals and the image masks the reference chose.
  * `pp_neck`: `FusionTransformerv4` (DeepInteraction++, BASELINE configs[4]) at the full ++ shape, eval mode (dropout off):
    gradients w.r.t. the input maps and every parameter.

Stored per gradient tensor: a strided sample of <= 4096 values, the absolute sum, the absolute maximum and the L2 norm.
ONE thread (the reference's depth scatter is an index_put_ with duplicate indices: sequential order is the contract, see
make_golden_grad_shapeR.py)."""
import os
import sys
import time

import numpy as np
import torch

from deepinteraction_amd import synth
from oracle import configs
from oracle.refpin.make_golden import OUT, randomize
from oracle.refpin import make_golden_grad_shapeR as gg

SEED_HEAD_INIT, SEED_HEAD_INPUT, SEED_HEAD_FUNCTIONAL = 11, 2, 33
SEED_PP_INPUT, SEED_PP_WEIGHTS, SEED_PP_FUNCTIONAL = 6, 15, 5
Q_HEAD = 200


def sample(t, n=4096):
    flat = t.detach().float().reshape(-1).cpu()
    idx = torch.arange(0, flat.numel(), max(1, flat.numel() // n))
    return dict(sample=flat[idx].numpy(), abssum=np.float64(flat.double().abs().sum()), absmax=np.float32(flat.abs().max()),
                l2=np.float64(flat.double().norm()))


def put(out, name, t, n=4096):
    for k, v in sample(t, n).items():
        out[f'{name}.{k}'] = v


# ------------------------------------------------------------------------------------------------ encoder, train mode
def enc_train_prepare(M):
    """Train mode with the pillar attention's dropout off (reference attribute: the nn.MultiheadAttention's own
    `dropout`; product attribute: `MMRI_I2P.dropout`)."""
    M.train()
    for blk in M.fusion_blocks:
        blk.I2P_block.learnedAlign.dropout = 0.0
        blk.I2P_block.dropout = 0.0
    return M


def main_enc_train():
    from oracle.refpin import load_reference
    ref = load_reference('reference')
    inp = gg.case()
    R = ref.encoder.DeepInteractionEncoder(num_layers=2, in_channels_img=gg.SHAPE['c_img'], in_channels_pts=gg.SHAPE['c_pts'],
                                           hidden_channel=128)
    randomize(R, gg.SEED_WEIGHTS)
    enc_train_prepare(R)
    img = inp['img_feats'].clone().requires_grad_(True)
    pts = inp['pts_feats'].clone().requires_grad_(True)
    t0 = time.time()
    im, (p0, p1) = R(img, pts, inp['img_metas'], inp['pts_metas'])
    t1 = time.time()
    gg.functional((im, p0, p1)).backward()
    print(f'reference encoder (train mode) at shape R: forward {t1 - t0:.1f} s, backward {time.time() - t1:.1f} s', flush=True)
    out = {}
    for name, t in [('d_img_feats', img.grad), ('d_pts_feats', pts.grad)] + \
                   [('p.' + n, p.grad) for n, p in R.named_parameters() if p.grad is not None]:
        put(out, name, t)
    for name, t in (('out_img', im), ('out_pts_conv', p0), ('out_pts', p1)):
        put(out, name, t)
    np.savez_compressed(os.path.join(OUT, 'grad_shapeR_train.npz'), **out)
    print('wrote grad_shapeR_train.npz', len(out), 'arrays')


# ------------------------------------------------------------------------------------------------ head
def head_cfg():
    cfg = configs.decoder_cfg(bev=synth.SHAPE_R['bev_hw'][0], num_proposals=Q_HEAD)
    cfg['dropout'] = 0.0
    return cfg


def head_state():
    """Default initialisation, drawn by the ORACLE's constructor (available wherever the tests run)."""
    from oracle import decoder as odec
    torch.manual_seed(SEED_HEAD_INIT)
    return odec.DeepInteractionDecoder(**head_cfg()).state_dict()


def head_case():
    """(feature maps [img, pts_conv, pts], img_metas): the maps at the magnitude the encoder hands over."""
    shape = synth.SHAPE_R
    g = torch.Generator().manual_seed(SEED_HEAD_INPUT)
    Hb, Wb = shape['bev_hw']
    feats = [torch.randn(6, 128, *shape['img_hw'], generator=g), torch.randn(1, 128, Hb, Wb, generator=g),
             torch.randn(1, 128, Hb, Wb, generator=g)]
    metas = synth.make_inputs(1, shape, seed=12)['img_metas']
    return feats, metas


def head_functional(out, dev='cpu'):
    gen = torch.Generator().manual_seed(SEED_HEAD_FUNCTIONAL)
    return sum((out[k].float() * torch.randn(out[k].shape, generator=gen).to(dev)).sum() for k in sorted(out))


def head_prepare(M):
    """The case's head: default initialisation, CONDITIONED as in a trained network (`harness.condition_head`: the residual
    branches of the four RoI blocks x 0.5, the device of tests/test_shapeR_parity_gpu.py::test_fp16_conditioned_head_B1_Q200 -
    as initialised, every RoI block multiplies a perturbation of its input by 2-3), train mode."""
    from deepinteraction_amd import harness
    M.load_state_dict(head_state())
    harness.condition_head(M)
    return M.train()


def _head_run(cls, feats, metas, **kw):
    R = head_prepare(cls(**head_cfg()))
    f = [t.clone().requires_grad_(True) for t in feats]
    res = R([f[1], f[2]], f[0], metas, **kw)[0][0]
    head_functional(res).backward()
    grads = dict(zip(('d_img', 'd_pts_conv', 'd_pts'), (t.grad for t in f)))
    grads.update({'p.' + n: p.grad for n, p in R.named_parameters() if p.grad is not None})
    return R, res, grads


def main_head():
    from oracle.refpin import load_reference
    ref = load_reference('reference')
    feats, metas = head_case()
    t0 = time.time()
    R, res, grads = _head_run(ref.decoder.DeepInteractionDecoder, feats, metas)
    print(f'reference head (train mode, dropout 0) at shape R, Q={Q_HEAD}: forward + backward {time.time() - t0:.1f} s', flush=True)
    # CONDITIONING, measured on the reference itself: the same run on inputs moved by 1e-6 of their value (what float32
    # convolutions on another device differ by).  RoIs that hang over the map edge pool all-zero bins, and the LayerNorm of a
    # constant row (DynamicConv, decoder_utils.py:614-621) amplifies round-off of that row by 1 / sqrt(eps) in the backward:
    # with the head as initialised the last RoI blocks' gradients move by up to 3 % of their norm and the outputs by 1-2e-3;
    # conditioned (head_prepare) by <= 1.2 % (median 5e-5) and 2.5e-4.  Stored per tensor as `.noise` (relative L2; outputs:
    # max error over max(1, range)); the GPU test bounds the product's deviation by a multiple of it.
    noise, fnoise = {n: 0.0 for n in grads}, {k: 0.0 for k in res}
    for seed in (1, 2, 3):                                     # the largest of three perturbed runs
        gen = torch.Generator().manual_seed(seed)
        moved = [t * (1 + 1e-6 * torch.randn(t.shape, generator=gen)) for t in feats]
        R2, res2, grads2 = _head_run(ref.decoder.DeepInteractionDecoder, moved, metas)
        assert torch.equal(R.query_labels, R2.query_labels), 'the perturbed run chose other proposals'
        for n, t in grads.items():
            noise[n] = max(noise[n], float((grads2[n].double() - t.double()).norm() / t.double().norm().clamp_min(1e-30)))
        for k, v in res.items():
            fnoise[k] = max(fnoise[k], float((res2[k] - v).abs().max() / max(1.0, float(v.abs().max()))))
    out = {}
    for name, t in grads.items():
        put(out, name, t, n=1024 if name.startswith('p.') else 4096)
        out[name + '.noise'] = np.float64(noise[name])
    for k, v in res.items():
        out['out.' + k] = v.detach().numpy()
        out['out.' + k + '.noise'] = np.float64(fnoise[k])
    out['query_labels'] = R.query_labels.numpy()
    out['on_the_image_mask'] = torch.stack(R.on_the_image_mask).numpy()
    np.savez_compressed(os.path.join(OUT, 'grad_head_shapeR.npz'), **out)
    print('wrote grad_head_shapeR.npz', len(out), 'arrays')


# ------------------------------------------------------------------------------------------------ ++ neck
def pp_case():
    return synth.make_inputs_pp(1, synth.SHAPE_PP, seed=SEED_PP_INPUT)


def pp_functional(outs, dev='cpu'):
    gen = torch.Generator().manual_seed(SEED_PP_FUNCTIONAL)
    return sum((o.float() * torch.randn(o.shape, generator=gen).to(dev)).sum() for o in outs)


def main_pp_neck():
    from oracle.refpin import load_reference
    ref = load_reference('reference')
    shape = synth.SHAPE_PP
    inp = pp_case()
    torch.manual_seed(1234)
    R = ref.fix_zero_layer_encoders(ref.encoder_pp.FusionTransformerv4(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'])))
    randomize(R, SEED_PP_WEIGHTS)
    R.eval()
    imgs = [t.clone().requires_grad_(True) for t in inp['img_feats']]
    pts = [t.clone().requires_grad_(True) for t in inp['pts_feats']]
    t0 = time.time()
    # (copies of the lists: the reference pops the first BEV level off the CALLER's list, fusion_transformerv4.py:85)
    oi, (p0, p1) = R(list(imgs), list(pts), inp['img_metas'], inp['pts_metas'])
    t1 = time.time()
    pp_functional((oi, p0, p1)).backward()
    print(f'reference ++ neck (eval) at the full ++ shape: forward {t1 - t0:.1f} s, backward {time.time() - t1:.1f} s', flush=True)
    out = {}
    for k, t in enumerate(imgs):
        put(out, f'd_img{k}', t.grad)
    for k, t in enumerate(pts):
        put(out, f'd_pts{k}', t.grad)
    for n, p in R.named_parameters():
        if p.grad is not None:
            put(out, 'p.' + n, p.grad)
    for name, t in (('out_img', oi), ('out_pts_conv', p0), ('out_pts', p1)):
        put(out, name, t)
    np.savez_compressed(os.path.join(OUT, 'grad_pp_neck.npz'), **out)
    print('wrote grad_pp_neck.npz', len(out), 'arrays')


if __name__ == '__main__':
    if sys.argv[1] != 'head':          # (the head has no duplicate-index scatter)
        torch.set_num_threads(1)
    {'enc_train': main_enc_train, 'head': main_head, 'pp_neck': main_pp_neck}[sys.argv[1]]()
