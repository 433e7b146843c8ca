"""Golden GRADIENT vectors of the MMRI encoder at the benched shape (BASELINE.json configs[1]/[2]: Fusion_0075_refactor
shape R, 2 interaction layers), produced by the REFERENCE'S OWN Python (its autograd Functions around the reference
locatt kernels compiled for the host, F.grid_sample, nn.MultiheadAttention through group_attn, the CPU depth completion),
imported unmodified from /root/reference (TEST INFRASTRUCTURE; run in the build container):

    python -m oracle.refpin.make_golden_grad_shapeR          (~30 min on ONE thread, ~25 GB of host memory)

Inputs, weights and the linear functional are regenerated from seeds by `case()` (shared with
tests/test_training_gpu.py::test_encoder_gradients_at_shape_R_match_reference_golden), so only the gradients are stored:
for every gradient tensor a strided sample of <= 4096 values plus its absolute sum (tests/golden/grad_shapeR.npz, ~2 MB).
Eval mode (running BatchNorm statistics, dropout off): what is pinned is the backward ARITHMETIC of the path at full
size - window-attention, pillar-attention and BEV-gather backward, every 1x1 / 3x3 weight gradient; train-mode BatchNorm
has its own test at the small shape.
"""
import os
import time

import numpy as np
import torch

from deepinteraction_amd import synth
from oracle.refpin.make_golden import OUT, randomize

SHAPE = synth.SHAPE_R
SEED_INPUT, SEED_WEIGHTS, SEED_FUNCTIONAL = 12, 21, 31


def case():
    """(inputs, functional weights generator seed): fp16-representable feature maps so that both sides see identical data."""
    inp = synth.make_inputs(1, SHAPE, seed=SEED_INPUT)
    inp['img_feats'] = inp['img_feats'].half().float()
    inp['pts_feats'] = inp['pts_feats'].half().float()
    return inp


def functional(outs, device='cpu'):
    gen = torch.Generator().manual_seed(SEED_FUNCTIONAL)
    return sum((t.float() * torch.randn(t.shape, generator=gen).to(device)).sum() for t in outs)


def sample(t, n=4096):
    flat = t.detach().float().reshape(-1).cpu()
    idx = torch.arange(0, flat.numel(), max(1, flat.numel() // n))
    return flat[idx].numpy(), np.float64(flat.double().abs().sum()), np.float32(flat.abs().max())


def main():
    from oracle.refpin import load_reference
    ref = load_reference('reference')
    # ONE thread, on purpose: the reference scatters the sparse depth with `depth_map[i, rows, cols] = depth` (encoder_utils.py
    # :172-174), an index_put_ with duplicate indices whose winner is the write ORDER.  torch parallelises index_put_ over
    # threads for 262 144 points: with 8 threads ~25 000 of the 134 400 depth pixels get a different winner than in sequential
    # order, and two 8-thread runs differ from EACH OTHER (the first golden file, made with 8 threads, disagreed with the
    # oracle and the product on 47 % of the image-map samples; per module at shape R the reference's I2P and window attention
    # agree with the oracle to 3e-7, only BEVWarp does not).  Sequential order (highest point index wins) is the contract the
    # oracle (oracle/encoder.py::scatter_depth) and the product's 64-bit atomicMax implement; with one thread the reference
    # follows it exactly.
    torch.set_num_threads(1)
    inp = case()
    R = ref.encoder.DeepInteractionEncoder(num_layers=2, in_channels_img=SHAPE['c_img'], in_channels_pts=SHAPE['c_pts'],
                                           hidden_channel=128)
    randomize(R, SEED_WEIGHTS)
    R.eval()
    img = inp['img_feats'].clone().requires_grad_(True)
    pts = inp['pts_feats'].clone().requires_grad_(True)
    t0 = time.time()
    im, (p0, p1) = R(img, pts, inp['img_metas'], inp['pts_metas'])
    t1 = time.time()
    functional((im, p0, p1)).backward()
    print(f'reference encoder at shape R: forward {t1 - t0:.1f} s, backward {time.time() - t1:.1f} s')
    out = {}
    for name, t in [('d_img_feats', img.grad), ('d_pts_feats', pts.grad)] + \
                   [('p.' + n, p.grad) for n, p in R.named_parameters() if p.grad is not None]:
        s, a, m = sample(t)
        out[name + '.sample'], out[name + '.abssum'], out[name + '.absmax'] = s, a, m
    for name, t in (('out_img', im), ('out_pts_conv', p0), ('out_pts', p1)):
        s, a, m = sample(t)
        out[name + '.sample'], out[name + '.abssum'], out[name + '.absmax'] = s, a, m
    np.savez_compressed(os.path.join(OUT, 'grad_shapeR.npz'), **out)
    print('wrote', os.path.join(OUT, 'grad_shapeR.npz'), len(out), 'arrays')


if __name__ == '__main__':
    main()
