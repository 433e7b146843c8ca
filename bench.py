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
    ap.add_argument('--eager', action='store_true',
                    help='launch every kernel from the host each step instead of replaying the captured hipGraph')
    ap.add_argument('--roofline-steps', type=int, default=5,
                    help='eager forwards run after the timed region to time the dominant kernel with HIP events')
    args = ap.parse_args()

    from deepinteraction_amd import ops, parallel, synth
    rank, local, world = parallel.env_rank()
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    parallel.init('nccl', device)                 # RCCL over xGMI; only the timing protocol uses it
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    shape = dict(R=synth.SHAPE_R, A=synth.SHAPE_A, TINY=synth.SHAPE_TINY)[args.shape]
    dtype = dict(f16=torch.float16, f32=torch.float32)[args.dtype]
    enc, dec = build_models(shape, args.proposals, dtype, device)
    # weak scaling: rank r owns samples [r*batch, (r+1)*batch) of the global batch (deepinteraction_amd/parallel.py)
    ids = parallel.sample_ids(0, args.batch, rank, world)
    data = to_device(synth.make_inputs(args.batch, shape, seed=parallel.sample_seed(ids[0])), device, dtype)
    n_pillars = int(data['pts_metas']['pillars'].shape[0])

    # One step = one forward of encoder + decoder on the resident batch.  Default: the forward is captured
    # once into a hipGraph (deepinteraction_amd/graphed.py) and every step replays it - all ~600 kernels
    # run each step, only the host-side launch work is gone.  --eager launches them from Python instead.
    with torch.no_grad():
        if args.eager:
            step = lambda: forward(enc, dec, data)
        else:
            from deepinteraction_amd.graphed import GraphedHotPath
            step = GraphedHotPath(enc, dec, data)
        for _ in range(args.warmup):
            step()
        # barrier + synchronize | K steps | barrier + synchronize, MAX over ranks
        elapsed = parallel.timed_region(step, args.steps, device)
        # dominant-kernel timing: HIP events right around the launch, on the launch stream, in eager
        # forwards of the same model and data (events cannot bracket one kernel inside a graph replay)
        forward(enc, dec, data)
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(args.roofline_steps):
            forward(enc, dec, data)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
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
                    algorithmic_bytes=alg_bytes,
                    timed_in=f'{args.roofline_steps} eager forwards right after the timed region, HIP events on the launch stream')
    pmc = os.path.join(ROOT, 'profiles', 'pmc_local_attn.json')
    if os.path.exists(pmc):                            # HBM bytes per launch from a separate rocprofv3 --pmc pass
        roofline['traffic'] = json.load(open(pmc)).get('hbm_bytes_per_launch')

    if rank == 0:
        out = dict(metric='samples/sec forward (Fusion_0075 synthetic)',
                   value=round(parallel.throughput(args.batch, args.steps, elapsed, world), 3),
                   unit='samples/s', n_gpus=args.gpus, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling='weak',
                   vs_baseline=None, dtype='f16' if dtype == torch.float16 else 'f32', data='synthetic',
                   config=dict(workload='Full MMRI encoder (2 layers) + MMPI decoder forward, '
                                        f'Fusion_0075_refactor shapes (shape {args.shape}), random-init weights',
                               batch_per_gpu=args.batch, global_batch=args.batch * args.gpus,
                               num_proposals=args.proposals, pillars=n_pillars,
                               launch='eager' if args.eager else 'hipGraph replay of the captured forward',
                               parallelism=f'{args.gpus} independent replicas, sharded by sample'),
                   roofline=roofline)
        if args.gpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(shape, args.proposals)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
