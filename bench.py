#!/usr/bin/env python
"""bench.py - forward throughput of the interaction hot path on MI355X.

Metric (BASELINE.json): samples/sec forward, Fusion_0075 synthetic.  A "step" is one forward of
the full MMRI encoder (2 layers) + MMPI decoder (1 decoder layer + 4 RoI layers, Q=200) over one
batch of synthetic Fusion_0075_refactor-shaped inputs (BASELINE.json configs[1]: image features
6x256x112x200, BEV 512x180x180, 262 144 points, fp16), inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Forward inference shards by sample with no data-path collective: every rank runs
its own replica on its own samples ("weak" scaling); the only collectives are the timing barrier
and the max-over-ranks reduction.  Rank 0 prints ONE JSON line.

Also reported on the same line:
  roofline      the dominant kernel (fused local-window attention on the 6x112x200 image maps):
                algorithmic bytes 4*n*C*H*W*2 = 137.6 MB per launch / average launch duration
                measured live with HIP events on the launch stream during the timed steps
  cpu_baseline  the CPU oracle (PyTorch fp32 restatement of the reference path, kind "port")
                timed on this box's host cores on one full-size sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_models(shape, num_proposals, dtype, device):
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder
    from deepinteraction_amd.configs import decoder_cfg
    torch.manual_seed(1234)
    enc = DeepInteractionEncoder(num_layers=2, in_channels_img=shape['c_img'], in_channels_pts=shape['c_pts'],
                                 hidden_channel=128)
    dec = DeepInteractionDecoder(**decoder_cfg(bev=shape['bev_hw'][0], num_proposals=num_proposals))
    g = torch.Generator().manual_seed(5)
    for m in list(enc.modules()) + list(dec.modules()):      # non-trivial BN statistics (random-init weights)
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    return enc.to(device, dtype).eval(), dec.to(device, dtype).eval()


def to_device(inp, device, dtype):
    pm = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
    pm['pts'] = [p.to(device) for p in inp['pts_metas']['pts']]
    return dict(img_feats=inp['img_feats'].to(device, dtype).contiguous(memory_format=torch.channels_last),
                pts_feats=inp['pts_feats'].to(device, dtype).contiguous(memory_format=torch.channels_last),
                img_metas=inp['img_metas'], pts_metas=pm)


def forward(enc, dec, d):
    img, pts = enc(d['img_feats'], d['pts_feats'], d['img_metas'], d['pts_metas'])
    return dec(pts, img, d['img_metas'])


def cpu_baseline(shape, num_proposals, budget_s=25.0):
    """The CPU oracle (kind "port": PyTorch fp32 restatement of the reference path, its per-sample
    and per-view Python loops and scipy depth completion included) on this box's host cores.
    BOUNDED: a 1/16-area probe of the workload is timed first; then the largest of
    {1/16, 1/4, full} area samples whose predicted time fits `budget_s` is timed and scaled by its
    area fraction to full-size-sample units.  Reported baseline only."""
    from deepinteraction_amd import synth
    from oracle import configs, decoder as odec, encoder as oenc
    cores = min(os.cpu_count() or 1, 16)          # more threads only add overhead to these small ops
    torch.set_num_threads(cores)

    def run(div):
        Hi, Wi = shape['img_hw'][0] // div, shape['img_hw'][1] // div
        Hb = shape['bev_hw'][0] // div
        sh = dict(shape, img_hw=(Hi, Wi), input_shape=(Hi * 4, Wi * 4), bev_hw=(Hb, Hb),
                  n_points=shape['n_points'] // (div * div))
        inp = synth.make_inputs(1, sh, seed=0)
        torch.manual_seed(1234)
        E = oenc.DeepInteractionEncoder(2, sh['c_img'], sh['c_pts'], 128).eval()
        D = odec.DeepInteractionDecoder(**configs.decoder_cfg(bev=Hb, num_proposals=num_proposals)).eval()
        with torch.no_grad():
            t0 = time.time()
            img, pts = E(inp['img_feats'], inp['pts_feats'], inp['img_metas'], inp['pts_metas'])
            D(pts, img, inp['img_metas'])
            return time.time() - t0, sh

    t16, sh = run(4)
    div, dt = 4, t16
    for d, growth in ((2, 4.0), (1, 16.0)):       # predicted from the probe, ~linear in area
        if t16 * growth * 1.3 <= budget_s:
            div = d
    if div != 4:
        dt, sh = run(div)
    frac = 1.0 / (div * div)
    return dict(value=round(frac / dt, 5), unit='samples/s', cores=cores, kind='port',
                sample=(f'oracle MMRI+MMPI forward, fp32, {cores} threads, on a 1/{div * div}-area sample '
                        f'(image feats 6x{sh["c_img"]}x{sh["img_hw"][0]}x{sh["img_hw"][1]}, BEV {sh["bev_hw"][0]}^2, '
                        f'{sh["n_points"]} points) in {dt:.1f} s; value = area fraction / time'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=1, help='samples per GPU per step')
    ap.add_argument('--proposals', type=int, default=200)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
    ap.add_argument('--shape', default='R', choices=['R', 'A', 'TINY'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from deepinteraction_amd import ops, synth
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    shape = dict(R=synth.SHAPE_R, A=synth.SHAPE_A, TINY=synth.SHAPE_TINY)[args.shape]
    dtype = dict(f16=torch.float16, f32=torch.float32)[args.dtype]
    enc, dec = build_models(shape, args.proposals, dtype, device)
    data = to_device(synth.make_inputs(args.batch, shape, seed=1000 * rank), device, dtype)
    n_pillars = int(data['pts_metas']['pillars'].shape[0])

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            forward(enc, dec, data)
        sync()
        ops.PROFILE = []                               # HIP-event pairs around the local-attention launches
        t0 = time.perf_counter()
        for _ in range(args.steps):
            forward(enc, dec, data)
        sync()
        elapsed = time.perf_counter() - t0
        prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # roofline of the dominant kernel: fused local-window attention on the image maps
    Hi, Wi = shape['img_hw']
    n_img = 6 * args.batch
    es = 2 if dtype == torch.float16 else 4
    alg_bytes = 4 * n_img * 128 * Hi * Wi * es
    durs = [s.elapsed_time(e) * 1e-3 for (name, n, s, e) in prof if name == 'local_attn_fwd' and n == n_img]
    avg = sum(durs) / max(len(durs), 1)
    achieved = alg_bytes / avg / 1e9 if durs else None
    roofline = dict(bound='hbm', kernel='di_local_attn_fwd, image side 6x112x200, 9x9, C=128 (local_attn_m2_kernel, 16x4 tiles)',
                    achieved=None if achieved is None else round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                    traffic=None, avg_launch_us=round(avg * 1e6, 2), launches=len(durs),
                    algorithmic_bytes=alg_bytes)
    pmc = os.path.join(ROOT, 'profiles', 'pmc_local_attn.json')
    if os.path.exists(pmc):                            # HBM bytes per launch from a separate rocprofv3 --pmc pass
        roofline['traffic'] = json.load(open(pmc)).get('hbm_bytes_per_launch')

    if rank == 0:
        total = args.gpus * args.batch * args.steps
        out = dict(metric='samples/sec forward (Fusion_0075 synthetic)', value=round(total / elapsed, 3),
                   unit='samples/s', n_gpus=args.gpus, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype='f16' if dtype == torch.float16 else 'f32', data='synthetic',
                   config=dict(workload='Full MMRI encoder (2 layers) + MMPI decoder forward, '
                                        f'Fusion_0075_refactor shapes (shape {args.shape}), random-init weights',
                               batch_per_gpu=args.batch, global_batch=args.batch * args.gpus,
                               num_proposals=args.proposals, pillars=n_pillars,
                               parallelism=f'{args.gpus} independent replicas, sharded by sample'),
                   roofline=roofline)
        if args.gpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(shape, args.proposals)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
