"""TEST INFRASTRUCTURE (never imported by the product): end-to-end parity of the HIP hot path against the oracle at
a whole configuration - the oracle's MMRI encoder (2 layers) + MMPI decoder forward on the CPU, and the statistics
the tests assert and `bench.py`'s `cpu_baseline` leg prints as its `parity` block.

Reference path being compared (all under /root/reference/projects/mmdet3d_plugin/models):
`necks/deepinteraction_encoder.py:79-85` -> `dense_heads/deepinteraction_decoder.py:201-313`.

Conventions
  * both sides see IDENTICAL inputs and parameters: for an fp16 product the inputs and the state_dict are rounded
    through fp16 first (what differs is the arithmetic, not the data);
  * no depth injection: the oracle runs its own scatter + completion, the product its device chain;
  * INT outputs (top-Q proposals, query labels, on-the-image masks) are compared exactly; continuous outputs as
    |got - ref| / max(1, max|ref|) per output key: max, median, 99.9th percentile;
  * an fp16 product may reorder near-tied heat-map scores (the synthetic heat map of random-init weights is noise:
    thousands of scores lie within fp16 round-off of the Q-th one).  The free-running comparison reports how many
    proposals / labels agree; the continuous decoder outputs are then compared against the oracle decoder run on
    the PRODUCT's proposals (`top_override`, an oracle-side switch - the product has no such hook).
"""
import time

import torch

from . import configs, decoder as odec, encoder as oenc


def randomize_bn(mods, seed=5):
    g = torch.Generator().manual_seed(seed)
    for m in mods:
        for sub in m.modules():
            if isinstance(sub, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                sub.running_mean.copy_(torch.randn(sub.running_mean.shape, generator=g) * 0.1)
                sub.running_var.copy_(torch.rand(sub.running_var.shape, generator=g) + 0.5)


def build_oracle(shape, num_proposals, seed=1234, state=None, round_fp16=False, num_layers=2):
    """Oracle encoder + decoder (fp32, CPU, eval).  `state` = (encoder state_dict, decoder state_dict) to load
    (e.g. the product's); round_fp16: True rounds every floating parameter / buffer through fp16, 'maps' only those the
    product's fp16 mode holds in fp16 (`deepinteraction_amd.precision.half_maps_`: the encoder and the two heat-map
    heads) - the "identical parameters" comparison that isolates the arithmetic."""
    torch.manual_seed(seed)
    E = oenc.DeepInteractionEncoder(num_layers, shape['c_img'], shape['c_pts'], 128)
    D = odec.DeepInteractionDecoder(**configs.decoder_cfg(bev=shape['bev_hw'][0], num_proposals=num_proposals))
    if state is None:
        randomize_bn([E, D])
    else:
        E.load_state_dict({k: v.detach().float().cpu() for k, v in state[0].items()})
        D.load_state_dict({k: v.detach().float().cpu() for k, v in state[1].items()})
    if round_fp16:
        mods = (E, D) if round_fp16 is True else (E, D.heatmap_head, D.heatmap_head_img)
        for m in mods:
            for t in list(m.parameters()) + list(m.buffers()):
                if t.is_floating_point():
                    t.data = t.data.half().float()
    return E.eval(), D.eval()


def oracle_encoder(E, inputs, fp16_inputs=False):
    q = (lambda t: t.half().float()) if fp16_inputs else (lambda t: t.float())
    t0 = time.time()
    with torch.no_grad():
        img, (p0, p1) = E(q(inputs['img_feats'].cpu()), q(inputs['pts_feats'].cpu()), inputs['img_metas'],
                          inputs['pts_metas'])
    return dict(img=img, pts_conv=p0, pts=p1, seconds=time.time() - t0)


def build_oracle_pp(shape, num_proposals, state, round_fp16=False, num_layers=2):
    """Oracle DeepInteraction++ neck + head (fp32, CPU, eval) holding `state` = (neck state_dict, head state_dict) of the
    product (reference `necks/fusion_transformerv4.py:26-127`, `dense_heads/deepinteractionplusplus_decoder.py:20-320`,
    kwargs of `Fusion_0075_plusplus.py:210-303`).  round_fp16='maps': the parameters the product's fp16 mode holds in fp16
    (the whole neck and the two heat-map heads) are rounded through fp16 - the comparison that isolates the arithmetic."""
    from . import plusplus as opp
    E = opp.FusionTransformerv4(**configs.encoder_pp_cfg(shape['c_img'], shape['c_pts'], num_layers))
    D = opp.DeepInteractionPlusPlusDecoder(**configs.decoder_cfg(bev=shape['bev_hw'][0], num_proposals=num_proposals))
    E.load_state_dict({k: v.detach().float().cpu() for k, v in state[0].items()})
    D.load_state_dict({k: v.detach().float().cpu() for k, v in state[1].items()})
    if round_fp16:
        mods = (E, D) if round_fp16 is True else (E, D.heatmap_head, D.heatmap_head_img)
        for m in mods:
            for t in list(m.parameters()) + list(m.buffers()):
                if t.is_floating_point():
                    t.data = t.data.half().float()
    return E.eval(), D.eval()


def oracle_encoder_pp(E, inputs):
    """The ++ neck on a `synth.make_inputs_pp` batch (lists of levels)."""
    t0 = time.time()
    with torch.no_grad():
        img, (p0, p1) = E([f.float().cpu() for f in inputs['img_feats']], [f.float().cpu() for f in inputs['pts_feats']],
                          inputs['img_metas'], inputs['pts_metas'])
    return dict(img=img, pts_conv=p0, pts=p1, seconds=time.time() - t0)


def oracle_decoder(D, enc, img_metas, top_override=None):
    t0 = time.time()
    with torch.no_grad():
        out = D([enc['pts_conv'], enc['pts']], enc['img'], img_metas, top_override=top_override)[0][0]
    return dict(out=out, labels=D.query_labels.clone(), masks=[m.clone() for m in D.on_the_image_mask],
                top=D.top_proposals.clone(), seconds=time.time() - t0,
                nms_margin=D.nms_margin.clone(), nms_heat=D.nms_heat.clone())


def heatmap_score_ties(prod_score, ref, tol=2e-3):
    """TIE-AWARE comparison of `query_heatmap_score` (B, classes, Q) against an oracle run `ref` on the SAME proposals
    (reference deepinteraction_decoder.py:225-253: an entry is the heat value of (class, cell) when that cell is the maximum of
    its 3x3 neighbourhood - `heat * (heat == local_max)` - and 0 otherwise).  Two fp16 heat-map heads may decide a near-tie the
    other way: the entry then differs by its WHOLE value (0 <-> heat).  Every entry must either agree within `tol`, or be such a
    flip - one side 0, the other the cell's heat value within `tol` - of a cell whose heat value is within `tol` of its 3x3
    maximum in the oracle.  Returns (entries, flips, worst error of the agreeing entries, worst margin of a flip)."""
    got, want = prod_score.detach().float().cpu(), ref['out']['query_heatmap_score'].float()
    B, C, Q = want.shape
    cell = (ref['top'] % ref['nms_margin'].shape[-1])[:, None, :].expand(-1, C, -1)
    margin = ref['nms_margin'].gather(-1, cell)              # 0 where the oracle's cell survives the NMS
    heat = ref['nms_heat'].gather(-1, cell)
    err = (got - want).abs()
    agree = err <= tol
    flip = ~agree & (((got == 0) & ((want - heat).abs() <= tol)) | ((want == 0) & ((got - heat).abs() <= tol)))
    bad = ~agree & ~flip
    assert not bool(bad.any()), ('query_heatmap_score differs without being an NMS flip', int(bad.sum()),
                                 float(err[bad].max()))
    worst_margin = float(margin[flip].max()) if bool(flip.any()) else 0.0
    assert worst_margin <= tol, ('an NMS flip of a cell that is NOT a near-tie', worst_margin)
    return dict(entries=int(want.numel()), flips=int(flip.sum()), max_err_agreeing=float(err[agree].max()),
                worst_flip_margin=worst_margin)


def rel_stats(got, ref):
    """max / median / 99.9-percentile of |got - ref| / max(1, max|ref|), and the scale."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    scale = max(1.0, ref.abs().max().item())
    d = ((got - ref).abs() / scale).flatten()
    ds = d[:: d.numel() // 4_000_000 + 1] if d.numel() > 4_000_000 else d      # quantiles on a strided subsample
    return dict(max=float(d.max()), median=float(ds.median()), p999=float(torch.quantile(ds.double(), 0.999)),
                scale=scale, frac_gt_1e3=float((d > 1e-3).float().mean()), frac_gt_1e2=float((d > 1e-2).float().mean()),
                abs_max=float(d.max()) * scale)          # in the output's own unit (`center`: BEV cells)


def compare_encoder(got, ref):
    """got: (img, [pts_conv, pts]) of the product; ref: oracle_encoder() -> {key: stats}."""
    gi, (g0, g1) = got
    return {k: rel_stats(g, ref[k]) for k, g in (('img', gi), ('pts_conv', g0), ('pts', g1))}


def compare_decoder(prod_out, prod_labels, prod_masks, prod_top, free, forced=None):
    """prod_*: the product decoder's output dict / query_labels / on_the_image_mask / flattened (class, cell)
    proposals; `free` = oracle_decoder() free-running; `forced` = oracle_decoder(top_override=product proposals) or
    None (the continuous comparison then uses `free`, which is only meaningful for identical proposals)."""
    labels, top = prod_labels.cpu(), prod_top.cpu()
    B, Q = top.shape
    common = sum(len(set(top[b].tolist()) & set(free['top'][b].tolist())) for b in range(B)) / float(B * Q)
    res = dict(proposals_identical=bool(torch.equal(top, free['top'])), proposal_set_overlap=common,
               label_agreement=float((labels == free['labels']).float().mean()))
    ref = forced if forced is not None else free
    res['labels_equal_on_same_proposals'] = bool(torch.equal(labels, ref['labels']))
    res['mask_agreement'] = [float((a.cpu() == b).float().mean()) for a, b in zip(prod_masks, ref['masks'])]
    res['keys'] = {k: rel_stats(prod_out[k], ref['out'][k]) for k in ref['out']}
    res['first_proposals'] = dict(product=top[0, :6].tolist(), oracle=free['top'][0, :6].tolist(),
                                  product_labels=labels[0, :6].tolist(), oracle_labels=free['labels'][0, :6].tolist())
    return res


def summarize(enc_stats, dec_stats):
    """Compact form for the bench JSON line."""
    r3 = lambda x: float(f'{x:.3g}')
    out = {k: dict(max=r3(v['max']), median=r3(v['median']), p999=r3(v['p999'])) for k, v in enc_stats.items()}
    out.update({f'dec.{k}': dict(max=r3(v['max']), median=r3(v['median']), p999=r3(v['p999']),
                                 frac_gt_1e3=r3(v['frac_gt_1e3']), abs_max=r3(v['abs_max']))
                for k, v in dec_stats['keys'].items()})
    out['proposal_set_overlap'] = r3(dec_stats['proposal_set_overlap'])
    out['label_agreement'] = r3(dec_stats['label_agreement'])
    out['mask_agreement'] = [r3(m) for m in dec_stats['mask_agreement']]
    out['first_proposals'] = dec_stats['first_proposals']
    return out
