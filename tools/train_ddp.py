#!/usr/bin/env python
"""Data-parallel training step of the interaction hot path on synthetic nuScenes-shaped batches
(BASELINE.json configs[2] at 1 GPU, configs[3] at N GPUs).

    python tools/train_ddp.py --steps 10                                              (1 GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tools/train_ddp.py --steps 10                                 (N GPUs, RCCL)

One process per GPU; every rank draws its own samples (deepinteraction_amd/parallel.py), runs encoder +
decoder forward in train() mode, the head loss against synthetic ground truth (Hungarian assignment on the
host, as the reference), backward through the HIP kernels, ONE bucketed gradient all-reduce over RCCL/xGMI
(`parallel.allreduce_gradients`: parameters a rank did not touch - `heatmap_head`, skipped RoI branches -
contribute zeros, the `find_unused_parameters=True` of the reference config) and AdamW with the
reference's lr / weight decay / grad clip (Fusion_0075_refactor.py:252-253).  Rank 0 prints one JSON line.
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import det3d_compat as dc, parallel, synth
from deepinteraction_amd.configs import decoder_cfg
from deepinteraction_amd.mmdet3d_plugin import DeepInteractionDecoder, DeepInteractionEncoder

TRAIN_CFG = dict(
    dataset='nuScenes',
    assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                  cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                  reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
    pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[1440, 1440, 40], voxel_size=[0.075, 0.075, 0.2],
    out_size_factor=8, code_weights=[1.0] * 8 + [0.2, 0.2], point_cloud_range=[-54.0, -54.0, -5.0, 54.0, 54.0, 3.0])


def synth_gt(seed, n=30):
    g = torch.Generator().manual_seed(seed)
    xy = (torch.rand(n, 2, generator=g) - 0.5) * 100
    z = torch.rand(n, 1, generator=g) * 2 - 2.5
    dims = torch.stack([torch.rand(n, generator=g) * 2 + 0.5, torch.rand(n, generator=g) * 5 + 0.5,
                        torch.rand(n, generator=g) * 2 + 0.8], 1)
    yaw = (torch.rand(n, 1, generator=g) - 0.5) * 6.28
    vel = torch.randn(n, 2, generator=g)
    return dc.LiDARBoxes(torch.cat([xy, z, dims, yaw, vel], 1)), torch.randint(0, 10, (n,), generator=g)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1, help='samples per GPU (the reference trains with 2)')
    ap.add_argument('--shape', default='R', choices=['R', 'TINY'])
    ap.add_argument('--pool', type=int, default=2, help='distinct pre-generated batches per rank')
    a = ap.parse_args()
    rank, local, world = parallel.env_rank()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    parallel.init('nccl', dev)
    shape = dict(R=synth.SHAPE_R, TINY=synth.SHAPE_TINY)[a.shape]
    bev = shape['bev_hw'][0]
    tc = dict(TRAIN_CFG, grid_size=[bev * 8, bev * 8, 40], voxel_size=[108.0 / (bev * 8)] * 2 + [0.2])
    torch.manual_seed(0)                                           # identical initial weights on every rank
    enc = DeepInteractionEncoder(2, shape['c_img'], shape['c_pts'], 128).to(dev).train()
    dec = DeepInteractionDecoder(**dict(decoder_cfg(bev=bev, num_proposals=200), train_cfg=tc)).to(dev).train()
    params = [p for m in (enc, dec) for p in m.parameters()]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)

    # a small pool of device-resident batches per rank, built before the timed region (the data loader is
    # out of scope; generating 262 144 points + pillars on the host takes longer than the step)
    pool = []
    for i in range(a.pool):
        ids = parallel.sample_ids(i, a.batch, rank, world)
        inp = synth.make_inputs(a.batch, shape, seed=parallel.sample_seed(ids[0]))
        pm = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp['pts_metas'].items()}
        pm['pts'] = [p.to(dev) for p in inp['pts_metas']['pts']]
        pool.append((inp['img_feats'].to(dev), inp['pts_feats'].to(dev), inp['img_metas'], pm,
                     [synth_gt(parallel.sample_seed(s)) for s in ids]))

    def step(i):
        img_in, pts_in, metas, pm, gts = pool[i % len(pool)]
        inp = dict(img_metas=metas)
        img, pts = enc(img_in, pts_in, metas, dict(pm))
        losses = dec.loss([g[0] for g in gts], [g[1] for g in gts], dec(pts, img, inp['img_metas']))
        loss = sum(v for k, v in losses.items() if k != 'matched_ious')
        opt.zero_grad(set_to_none=True)
        loss.backward()
        parallel.allreduce_gradients(params, world)                # one RCCL all-reduce per bucket
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], max_norm=0.1, norm_type=2)
        opt.step()
        return float(loss)
    for i in range(a.warmup):
        step(i)
    losses = []
    it = iter(range(a.warmup, a.warmup + a.steps))
    elapsed = parallel.timed_region(lambda: losses.append(step(next(it))), a.steps, dev)
    if rank == 0:
        print(json.dumps(dict(metric='samples/sec training step (forward + loss + backward + all-reduce + AdamW)',
                              value=round(parallel.throughput(a.batch, a.steps, elapsed, world), 3), unit='samples/s',
                              n_gpus=world, steps=a.steps, ms_per_step=round(elapsed / a.steps * 1e3, 2),
                              batch_per_gpu=a.batch, shape=a.shape, data=f'synthetic, pool of {a.pool} device-resident batches per rank',
                              first_loss=round(losses[0], 4), last_loss=round(losses[-1], 4))))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
