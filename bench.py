#!/usr/bin/env python
"""bench.py - throughput of the interaction hot path on MI355X.

Metric (BASELINE.json): samples/sec forward, Fusion_0075 synthetic.  A "step" is one pass of the hot path over one
batch: the full MMRI encoder (2 layers) + MMPI decoder (1 decoder layer + 4 RoI layers, Q=200) forward on the next
device-resident synthetic sample of a small pool.  Default workload = BASELINE.json configs[1]: Fusion_0075_refactor
shapes (image features 6x256x112x200, BEV 512x180x180, 262 144 points), fp16.
Hand-over of a sample to the captured forward (`--handover`): `resident` (default since round 5) = zero-copy, the
sample lies in the static input buffers of the captured forward that is replayed (where its producer wrote it; one
capture per pool sample stands for that), every replay reads it from HBM; `copy` (rounds 2-4) = the next pool sample is
first copied into the static buffers (`GraphedHotPath.load`: features, points, pillars, geometry constants) - the line
carries that figure too, as `copy_handover`.

    python bench.py --gpus N --steps K --warmup W           N > 1 without a torchrun environment re-launches itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W              as `torch.distributed.run --nproc-per-node N` (one rank per GPU, RCCL)

Other workloads of BASELINE.json `configs` (parity-test configurations; not the headline line):
    --model pp      configs[4]: DeepInteraction++ (Fusion_0075_plusplus neck + head) forward
    --mode train    configs[2] (N = 1) / configs[3] (N > 1): forward + head loss + backward + gradient all-reduce + AdamW

The path shards by sample with no data-path collective in the forward: every rank runs its own replica
on its own samples ("weak" scaling); the collectives are the timing barrier, the max-over-ranks reduction and - in
training - the gradient all-reduce over RCCL.  Rank 0 prints ONE JSON line, with
  roofline      the dominant kernel (fused local-window attention on the 6x112x200 image maps; round 4: the ring generation,
                csrc/local_attn_ring.hip): algorithmic bytes
                4*n*C*H*W*2 = 137.6 MB per launch / average launch duration measured live with HIP events on the
                launch stream; `traffic`, `mfma_busy`, `lds_busy` from the committed rocprofv3 --pmc passes (profiles/)
  cpu_baseline  the CPU oracle (PyTorch fp32 restatement of the reference path, kind "port") timed on this box's
                host cores on one full-size sample (rank 0, N=1 only)
  parity        the product's outputs on that same sample against the oracle's (same state_dict, no depth injection)
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=1, help='samples per GPU per step')
    ap.add_argument('--proposals', type=int, default=200)
    ap.add_argument('--dtype', default='f16', choices=['f16', 'f32'])
    ap.add_argument('--shape', default='R', choices=['R', 'A', 'TINY'])
    ap.add_argument('--model', default='v1', choices=['v1', 'pp'], help='pp = DeepInteraction++ (configs[4])')
    ap.add_argument('--mode', default='forward', choices=['forward', 'train'], help='train = configs[2]/[3]')
    ap.add_argument('--inflight', type=int, default=4,
                    help='samples in flight per GPU: N lanes (HIP streams), each replaying captured forwards of its own samples one '
                         'behind the other (a step = N x batch samples).  The decoder half of a forward is a chain of launches that '
                         'fill a fraction of the chip; more samples in flight fill more of it.  4 since the lanes are launched from '
                         'one host thread each (--launch-threads: 1 030 / 1 090 / 1 092 / 1 108 samples/s at 3 / 4 / 6 / 8 lanes, '
                         'tools/launch_cost.py; behind ONE launching thread 957 / 939 at 3 / 4, the default 3 of the first half '
                         'of round 5); 2 = the reference\'s samples_per_gpu (Fusion_0075_refactor.py:94), the default of rounds '
                         '2-4; 1 = one sample at a time (the latency figure, also reported as `single_sample`)')
    ap.add_argument('--launch-threads', type=int, default=1, choices=[0, 1],
                    help='1: every lane is launched from its own host thread (graphed.LaneLaunchers) - a replay costs the launching '
                         'thread 0.3-1.0 ms of hipGraphLaunch, of the order of the GPU time per sample; 0: one launching thread')
    ap.add_argument('--amp', action='store_true',
                    help='train mode: mixed precision - the hot path under torch.autocast(fp16) (fp16 activations, the fused '
                         'matrix-core window attention forward / backward of csrc/local_attn_train.hip), float32 master weights, '
                         'BatchNorm statistics, soft-max, scatter accumulation and loss; the line then says dtype f16.  Default: '
                         'float32, the arithmetic of the reference configuration')
    ap.add_argument('--train-eager', action='store_true',
                    help='train mode: host launches instead of the two replayed hipGraphs (forward, backward) around the eager loss')
    ap.add_argument('--from-images', action='store_true',
                    help='forward mode: the captured forward starts from the six camera images - the frozen ResNet-50 + FPN '
                         'stand-in (FrozenResNetFPN, torch / MIOpen, random init) runs inside every replay; with --model pp '
                         'the Swin-T + FPN stand-in (FrozenSwinFPN; that combination was not yet run on the device)')
    ap.add_argument('--from-points', action='store_true',
                    help='start every step from the raw points: the pillars of pts_metas are rebuilt by the voxeliser inside '
                         'the captured forward (detector glue, detectors/deepinteraction.py:120-171) instead of being loaded')
    ap.add_argument('--from-lidar', action='store_true',
                    help='forward mode: the frozen LiDAR branch (hard voxelisation at 0.075 m, HardSimpleVFE, the sparse 3-D encoder '
                         'without spconv - csrc/sparse_conv.hip -, SECOND, SECONDFPN: FrozenLidarBackbone, random init) runs EAGERLY on the '
                         'lane\'s stream in front of every replay and writes the BEV map into the captured forward\'s static input (its '
                         'shapes depend on the number of active voxels: not capturable).  Combine with --from-images --from-points for '
                         'the forward from raw sensor tensors')
    ap.add_argument('--from-raw', action='store_true',
                    help='the per-sample host work INSIDE the step: every step starts from NCHW device feature maps (the '
                         'reference boundary: frozen backbones emit NCHW), raw points / pillars and the metas - the channels-last '
                         'transposing copy, GraphedHotPath.prepare() (padding, host 4x4 inverses, H2D of the geometry constants) '
                         'and load() all run inside the timed region.  A secondary line; the headline hands prepared records')
    ap.add_argument('--pool', type=int, default=8, help='distinct device-resident samples cycled through the steps')
    ap.add_argument('--handover', default='resident', choices=['resident', 'copy'],
                    help='graph-replayed forward modes: how a step\'s sample becomes the captured forward\'s input.  resident '
                         '(default since round 5): zero-copy - the producer has written the sample into the static input buffers '
                         'of its in-flight slot (one captured forward per pool sample stands for that here: a lane replays the '
                         'forwards of ITS samples in turn, every replay reads its maps / points / pillars from HBM where they '
                         'lie).  copy (rounds 2-4): one captured forward per lane, every step first copies the next pool sample '
                         '(110 MB, ~40 us + 5 small geometry copies) into its static buffers - timed again after the headline and '
                         'reported as `copy_handover`')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true',
                    help='launch every kernel from the host each step instead of replaying the captured hipGraph')
    ap.add_argument('--roofline-steps', type=int, default=5,
                    help='eager forwards run after the timed region to time the dominant kernel with HIP events')
    ap.add_argument('--settle-ms', type=float, default=300.0,
                    help='untimed replays before the W warmup steps: the clocks of a just-leased, idle GPU ramp for a few '
                         'hundred ms (the driver times 20 steps = 0.07 s); reported in config.settle_ms')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / timing protocol only (CPU, gloo): no GPU work; used by tests/test_parallel.py')
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: one process per GPU through torch.distributed.run
    on this node (the reference's tools/dist_train.sh:7-9 launch contract: one command per node)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ CPU side
def cpu_baseline(shape, num_proposals, state, sample, product_out, budget_s=60.0, state_same=None, conditioned=None):
    """The CPU oracle (kind "port": PyTorch fp32 restatement of the reference path, its per-sample and per-view
    Python loops and scipy depth completion included) on this box's host cores, BOUNDED: a 1/16-area probe of the
    workload is timed first; when the predicted full-size time fits `budget_s` the oracle runs the full-size
    `sample` with the product's own state_dict - that run is both the baseline and the reference of the `parity`
    block - otherwise the largest area fraction that fits is timed and scaled, and parity is left to the tests."""
    import torch
    from deepinteraction_amd import synth
    from oracle import parity
    cores = min(os.cpu_count() or 1, 16)          # more threads only add overhead to these small ops
    torch.set_num_threads(cores)

    def probe(div):
        Hi, Wi = shape['img_hw'][0] // div, shape['img_hw'][1] // div
        Hb = shape['bev_hw'][0] // div
        sh = dict(shape, img_hw=(Hi, Wi), input_shape=(Hi * 4, Wi * 4), bev_hw=(Hb, Hb),
                  n_points=shape['n_points'] // (div * div))
        inp = synth.make_inputs(1, sh, seed=0)
        E, D = parity.build_oracle(sh, num_proposals)
        t0 = time.time()
        parity.oracle_decoder(D, parity.oracle_encoder(E, inp), inp['img_metas'])
        return time.time() - t0, sh

    t16, sh = probe(4)
    par = None
    if t16 * 16.0 * 1.3 <= budget_s:
        E, D = parity.build_oracle(shape, num_proposals, state=state)
        t0 = time.time()
        ref_enc = parity.oracle_encoder(E, sample)
        free = parity.oracle_decoder(D, ref_enc, sample['img_metas'])
        dt, div, sh = time.time() - t0, 1, shape
        if product_out is not None:
            got_enc, out, labels, masks, top = product_out
            forced = parity.oracle_decoder(D, ref_enc, sample['img_metas'], top_override=top.cpu())
            par = parity.summarize(parity.compare_encoder(got_enc, ref_enc),
                                   parity.compare_decoder(out, labels, masks, top, free, forced))
            par['vs'] = ('CPU oracle full forward on the same sample with the FLOAT32 parameters of the model (before the '
                         'fp16 conversion of the map side), no depth injection; continuous outputs as '
                         '|got-ref|/max(1,max|ref|) (abs_max in the output\'s unit: BEV cells for center); decoder outputs '
                         'against the oracle decoder run on the product\'s proposals')
            if state_same is not None:        # the arithmetic alone: the oracle holds the product's own parameter values
                E2, D2 = parity.build_oracle(shape, num_proposals, state=state_same)
                ref2 = parity.oracle_encoder(E2, sample)
                free2 = parity.oracle_decoder(D2, ref2, sample['img_metas'])
                forced2 = parity.oracle_decoder(D2, ref2, sample['img_metas'], top_override=top.cpu())
                same = parity.summarize(parity.compare_encoder(got_enc, ref2),
                                        parity.compare_decoder(out, labels, masks, top, free2, forced2))
                same.pop('first_proposals', None)
                same['vs'] = ('the same with the oracle holding the product\'s parameter VALUES (encoder and heat-map '
                              'heads rounded through fp16, token path float32 on both sides): arithmetic only')
                par['identical_parameters'] = same
            if conditioned is not None:       # (head state_dict, product outputs): the same encoder under a conditioned head
                dec_state_c, (_, out_c, labels_c, masks_c, top_c) = conditioned
                _, Dc = parity.build_oracle(shape, num_proposals, state=(state[0], dec_state_c))
                free_c = parity.oracle_decoder(Dc, ref_enc, sample['img_metas'])
                forced_c = parity.oracle_decoder(Dc, ref_enc, sample['img_metas'], top_override=top_c.cpu())
                dc = parity.compare_decoder(out_c, labels_c, masks_c, top_c, free_c, forced_c)
                r3 = lambda x: float(f'{x:.3g}')
                par['conditioned_head'] = dict(
                    {f'dec.{k}': dict(max=r3(v['max']), median=r3(v['median']), p999=r3(v['p999']), frac_gt_1e3=r3(v['frac_gt_1e3']))
                     for k, v in dc['keys'].items()},
                    vs='the same encoder, kernels and float32-parameter oracle, but the head\'s four RoI blocks conditioned '
                       '(harness.condition_head: residual branches x 0.5 - as initialised every block multiplies its input '
                       'error by 2-3, which is what sets the tail of dec.* above)')
    else:
        div = 2 if t16 * 4.0 * 1.3 <= budget_s else 4
        dt, sh = probe(div) if div != 4 else (t16, sh)
    frac = 1.0 / (div * div)
    base = dict(value=round(frac / dt, 5), unit='samples/s', cores=cores, kind='port',
                protocol='ONE timed oracle forward after a 1/16-area probe (bounded: the full-size forward takes ~17 s; BASELINE.md 3 '
                         'names 2 warm-up + 5 timed iterations, which would be two minutes of CPU work inside a bench run that must '
                         'finish within minutes) - a baseline figure, not a benchmark of the oracle',
                sample=(f'oracle MMRI+MMPI forward, fp32, {cores} threads, on a 1/{div * div}-area sample '
                        f'(image feats 6x{sh["c_img"]}x{sh["img_hw"][0]}x{sh["img_hw"][1]}, BEV {sh["bev_hw"][0]}^2, '
                        f'{sh["n_points"]} points) in {dt:.1f} s; value = area fraction / time'))
    return base, par


def cpu_baseline_pp(shape, num_proposals, state, sample, product_out):
    """DeepInteraction++ (configs[4]): the CPU oracle (`oracle/plusplus.py`, kind "port") runs ONE full-size sample with
    the model's float32 parameters - ~40 s on 8-16 host threads (the per-camera Python loops of the polar attention, the
    deformable gathers as grid_sample) - which is both the baseline and the reference of the `parity` block."""
    import torch
    from oracle import parity
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    E, D = parity.build_oracle_pp(shape, num_proposals, state)
    t0 = time.time()
    ref_enc = parity.oracle_encoder_pp(E, sample)
    free = parity.oracle_decoder(D, ref_enc, sample['img_metas'])
    dt = time.time() - t0
    par = None
    if product_out is not None:
        got_enc, out, labels, masks, top = product_out
        forced = parity.oracle_decoder(D, ref_enc, sample['img_metas'], top_override=top.cpu())
        par = parity.summarize(parity.compare_encoder(got_enc, ref_enc),
                               parity.compare_decoder(out, labels, masks, top, free, forced))
        par['vs'] = ('CPU oracle full ++ forward (neck + head) on the same sample with the FLOAT32 parameters of the model, '
                     'no depth injection; continuous outputs as |got-ref|/max(1,max|ref|); decoder outputs against the '
                     'oracle head run on the product\'s proposals')
    Hi, Wi = shape['img_hw']
    base = dict(value=round(1.0 / dt, 5), unit='samples/s', cores=cores, kind='port',
                sample=(f'oracle FusionTransformerv4 + DeepInteractionPlusPlusDecoder forward, fp32, {cores} threads, ONE full-size '
                        f'sample (image levels 6x{shape["c_img"]}x{Hi}x{Wi} / {Hi // 2}x{Wi // 2}, BEV {shape["bev_hw"][0]}^2, '
                        f'{shape["n_points"]} points) in {dt:.1f} s'))
    return base, par


def cpu_baseline_train(args):
    """The training line's CPU figure: the oracle (kind "port") forward + BACKWARD on this box's host cores, bounded - the
    1/4-area sample of the workload (the full-size forward alone is 17 s; with the backward it would be a minute), scaled by
    the area.  The scalar that is differentiated is the sum of the head's outputs (the loss itself is the product's own code -
    the reference's loss functions live in mmdet / mmdet3d, absent here); the backward of the whole MMRI + MMPI graph is what
    costs the time either way.  --model pp: the ++ oracle at the full shape is ~40 s forward; a 1/4-area ++ shape is used."""
    import torch
    from deepinteraction_amd import harness, synth
    from oracle import parity
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    pp = args.model == 'pp'
    shape = (harness.SHAPES_PP if pp else harness.SHAPES)[args.shape]
    div = 2 if shape['bev_hw'][0] >= 100 else 1
    Hi, Wi = shape['img_hw'][0] // div, shape['img_hw'][1] // div
    if pp:
        Hi, Wi = Hi // 2 * 2, Wi // 2 * 2
    Hb = shape['bev_hw'][0] // div
    sh = dict(shape, img_hw=(Hi, Wi), input_shape=(Hi * 4, Wi * 4), bev_hw=(Hb, Hb), n_points=shape['n_points'] // (div * div))
    nprop = args.proposals if Hb >= 50 else 24
    if pp:
        inp = synth.make_inputs_pp(1, sh, seed=0)
        e32, d32 = harness.build_models_pp(sh, nprop, torch.float32, 'cpu')
        E, D = parity.build_oracle_pp(sh, nprop, (e32.state_dict(), d32.state_dict()))
        feats = ([f.float() for f in inp['img_feats']], [f.float() for f in inp['pts_feats']])
    else:
        inp = synth.make_inputs(1, sh, seed=0)
        E, D = parity.build_oracle(sh, nprop)
        feats = (inp['img_feats'].float(), inp['pts_feats'].float())
    for p_ in [p for m in (E, D) for p in m.parameters()]:
        p_.requires_grad_(True)
    t0 = time.time()
    with torch.enable_grad():
        img, (p0, p1) = E(feats[0], feats[1], inp['img_metas'], inp['pts_metas'])
        out = D([p0, p1], img, inp['img_metas'])[0][0]
        scalar = sum(v.float().sum() for v in out.values() if torch.is_tensor(v) and v.is_floating_point() and v.requires_grad)
        scalar.backward()
    dt = time.time() - t0
    frac = 1.0 / (div * div)
    return dict(value=round(frac / dt, 5), unit='samples/s', cores=cores, kind='port',
                sample=(f'oracle {"FusionTransformerv4 + ++ head" if pp else "MMRI + MMPI"} forward + backward (sum of the head outputs), '
                        f'fp32, {cores} threads, ONE step on a 1/{div * div}-area sample (image feats {Hi}x{Wi}, BEV {Hb}^2, '
                        f'{sh["n_points"]} points) in {dt:.1f} s; value = area fraction / time'))


def pmc_file(name):
    p = os.path.join(ROOT, 'profiles', name)
    return json.load(open(p)) if os.path.exists(p) else {}


# ------------------------------------------------------------------------------------------------ workloads
def run_dry(args, parallel, rank, world):
    """No GPU: exercises launcher, rank environment, core binding, barrier / max-over-ranks timing on gloo.  --mode train: also
    the training step's gradient-reducer wiring on the REAL parameter list (the two modules built on the CPU, a synthetic
    backward of the real shapes - `train_step.synthetic_backward` - with the image RoI blocks unused on odd ranks, as when
    every view of a rank's sample holds <= 1 query: reference decoder_utils.py:726, find_unused_parameters=True)."""
    parallel.init('gloo')
    binding = parallel.bind_rank_threads(int(os.environ.get('LOCAL_RANK', rank)), world)
    extra = {}
    if args.mode == 'train':
        import torch
        from deepinteraction_amd import harness, synth, train_step
        enc, dec = harness.build_models(synth.SHAPE_TINY, args.proposals, torch.float32, 'cpu')
        enc.train(), dec.train()
        named = [(('enc.' if m is enc else 'dec.') + n, p) for m in (enc, dec) for n, p in m.named_parameters()]
        params = train_step.trainable_parameters(enc, dec)
        red = parallel.GradientReducer(params, world)
        unused = ('dec.decode_head.0.', 'dec.decode_head.2.') if rank % 2 == 1 else ()
        worst = [0.0]

        def step():
            for p in params:
                p.grad = None
            ws = train_step.synthetic_backward(named, rank, step.i, unused=unused + ('dec.heatmap_head.',))
            red.finish()
            # every rank knows every rank's weights: the averaged gradient must be the mean over ALL ranks (zeros where unused)
            for i, (n, p) in enumerate(named):
                if n.startswith('dec.heatmap_head.'):
                    assert p.grad is None, n                      # unused everywhere: stays frozen
                    continue
                want = 0.0
                for r in range(world):
                    image_block = n.startswith('dec.decode_head.0.') or n.startswith('dec.decode_head.2.')
                    if not (r % 2 == 1 and image_block):
                        want += (r + 1) * 0.5 + (step.i + 1) * 0.125 + (i % 7) * 0.03125
                worst[0] = max(worst[0], float((p.grad - want / world).abs().max()))
            step.i += 1
        step.i = 0
        el = parallel.timed_region(step, args.steps)
        worst_all = parallel.max_over_ranks(worst[0])
        assert worst_all < 1e-5, worst_all
        extra = dict(parameters=len(params), gradient_bytes=sum(p.numel() for p in params) * 4, buckets=len(red.buckets),
                     unused_on_odd_ranks='decode_head.0 / decode_head.2 (image RoI blocks)', max_abs_error=worst_all)
        workload = 'dry-run train: gradient reducer on the real parameter list, synthetic backward'
    else:
        el = parallel.timed_region(lambda: time.sleep(0.002 * (rank + 1)), args.steps)
        workload = 'dry-run'
    seen = parallel.sum_over_ranks(1)
    if rank == 0:
        print(json.dumps(dict(metric='dry-run (launcher + timing protocol only)', value=round(
            parallel.throughput(args.batch, args.steps, el, world), 3), unit='samples/s', n_gpus=args.gpus,
            steps=args.steps, warmup=args.warmup, ms_per_step=round(el / args.steps * 1e3, 3), higher_is_better=True,
            scaling='weak', vs_baseline=None, dtype='none', data='none',
            config=dict(workload=workload, ranks_seen=int(seen), backend='gloo', cpu_binding=binding, **extra))))


def main():
    args = parse()
    if args.eager or args.dry_run or args.mode != 'forward' or (args.model == 'pp' and args.from_images):
        args.inflight = 1          # several samples in flight exist for the graph-replayed forwards only
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))
    import torch
    # Lines with LIBRARY convolutions (the training step's 3x3 convolutions, the frozen backbones of --from-images / --from-lidar):
    # MIOpen times its solvers per convolution shape during the warm-up, like the reference's tools do under the config flag
    # `cudnn_benchmark` (tools/test.py:149-151).  DI_MIOPEN_FIND=0 keeps the immediate-mode pick (A/B); the headline has no
    # library convolution.
    lib_convs = args.mode == 'train' or args.from_images or args.from_lidar
    if os.environ.get('DI_MIOPEN_FIND', '1' if lib_convs else '0') == '1':
        torch.backends.cudnn.benchmark = True
    from deepinteraction_amd import parallel
    rank, local, world = parallel.env_rank()
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.dry_run:
        run_dry(args, parallel, rank, world)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    from deepinteraction_amd import harness, ops, synth
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    parallel.init('nccl', device)                 # RCCL over xGMI
    ranks_seen = int(parallel.sum_over_ranks(1, device))
    # one node, N ranks: every rank's host threads (lane launchers, the training step's Hungarian assignment) on its own cores
    binding = parallel.bind_rank_threads(local, world)
    if args.mode == 'train':
        from deepinteraction_amd import train_step
        out = train_step.bench(args, rank, world, device)       # --model pp: the DeepInteraction++ step (configs[4]), eager launches
        if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_train(args)
    elif args.model == 'pp':
        out = bench_forward_pp(args, rank, world, device)
    else:
        out = bench_forward(args, rank, world, device)
    if rank == 0:
        out['config']['ranks_seen'] = ranks_seen
        out['config']['cpu_binding'] = binding
        out['config']['miopen_find'] = bool(torch.backends.cudnn.benchmark)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


def settle(step, ms):
    """Untimed: run `step` for about `ms` milliseconds so that the timed region starts on a GPU at its working clocks."""
    import torch
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(8):
            step()
        torch.cuda.synchronize()


def launch_text(resident):
    if resident:
        return ('per step and sample in flight: hipGraph replay of the captured forward whose static input buffers hold the '
                'sample (zero-copy hand-over: one capture per pool sample, a lane replays the captures of its samples in turn)')
    return ('per step and sample in flight: load() of the next pool sample into the captured buffers + hipGraph replay of the '
            'captured forward')


def graphed_steps(capture, pool, cap, n_lanes, resident, raw_pool=None, launch_threads=True, before=None):
    """The graph-replayed step of both forward lines.  `capture(inputs, overlap)` -> GraphedHotPath.  Returns (step, step_copy, step1,
    graphs, records, g): `step` = one step of `n_lanes` samples in flight under the chosen hand-over, `step_copy` = the same
    with the copying hand-over of rounds 2-4 (on the first capture of every lane), `step1` = one sample at a time, `g` = the
    LAST capture (the modules' output attributes point at its static outputs: the parity leg load()s pool[0] into it).

    resident: one capture per pool sample, all with the layout and pillar capacity of the largest sample (captured from
    `prepare()`d records, padded as load() pads); lane l owns samples l, l + n_lanes, ... and replays their captures in turn
    on its stream - no capture is ever replayed on two streams.  `before`: a one-element list holding a callable(capture) that the
    lane runs on its stream right before the replay (the eager LiDAR branch of --from-lidar), or None."""
    import torch
    from deepinteraction_amd.graphed import LaneLaunchers
    # captures that run side by side are single-stream ones (GraphedHotPath `overlap`); the one-at-a-time figure uses
    # captures with the forward's own fork / join branches.  DI_OVERLAP set: that value everywhere (A/B runs)
    lane_overlap = 0 if n_lanes > 1 and 'DI_OVERLAP' not in os.environ else None
    first = capture(pool[cap], None)                               # the largest sample sets the capacity
    records = [first.prepare(d) for d in pool]
    solo = [first] if not resident or lane_overlap is not None else []
    if resident:
        while len(records) < n_lanes:
            records = records + records
        if lane_overlap is not None and len(records) > 1:
            solo.append(capture(first.record_inputs(records[1]), None))
        graphs = [capture(first.record_inputs(r), lane_overlap) for r in records]
        own = [graphs[l::n_lanes] for l in range(n_lanes)]
    else:
        own = [[first if lane_overlap is None else capture(pool[cap], lane_overlap)]] + \
              [[capture(pool[cap], lane_overlap)] for _ in range(n_lanes - 1)]
        graphs = [o[0] for o in own]
    g = graphs[-1]
    lanes = [torch.cuda.Stream() for _ in own] if n_lanes > 1 else [None]
    for lane in lanes:                                             # the lanes start after everything queued so far
        if lane is not None:
            lane.wait_stream(torch.cuda.current_stream())
    it, turn, one = [0], [0], [0]
    # one launching host thread per lane (graphed.LaneLaunchers: four lanes are host-bound behind one launching thread)
    launchers = LaneLaunchers(lanes) if n_lanes > 1 and launch_threads else None

    def issue(fns):
        if launchers is not None:
            launchers.run(fns)
            return
        for fn, lane in zip(fns, lanes):
            with torch.cuda.stream(lane) if lane is not None else contextlib.nullcontext():
                fn()

    def copy_then_replay(o, i):
        def fn():
            if raw_pool is not None:
                o[0].load_raw(raw_pool[i % len(raw_pool)])           # NCHW -> channels-last inside the one copy
            else:
                o[0].load(records[i % len(records)])                 # per-sample: copies into the captured buffers ...
            o[0]()                                                   # ... and one replay of the captured forward
        return fn

    def step_copy():
        issue([copy_then_replay(o, it[0] + l) for l, o in enumerate(own)])
        it[0] += len(own)

    def fed(gg):
        def fn():
            before[0](gg)
            gg()
        return fn if before is not None and before[0] is not None else gg

    def step_resident():
        issue([fed(o[turn[0] % len(o)]) for o in own])               # the next of every lane's samples, where it lies
        turn[0] += 1

    def step1():
        if resident:
            seq = solo if solo else graphs
            fed(seq[one[0] % len(seq)])()
        else:
            solo[0].load(records[one[0] % len(records)])
            solo[0]()
        one[0] += 1
    return (step_resident if resident else step_copy), step_copy, step1, graphs, records, g


def secondary_lines(args, parallel, device, world, n_lanes, step1, step_copy):
    """After the headline's timed region: the one-sample-at-a-time figure (when several are in flight) and, under the
    resident hand-over, the same step with the copying hand-over of rounds 2-4."""
    import torch
    single = copy_handover = None
    if n_lanes > 1:
        torch.cuda.synchronize()
        for _ in range(max(2, args.warmup // 2)):
            step1()
        e1 = parallel.timed_region(step1, args.steps, device)
        single = dict(value=round(parallel.throughput(args.batch, args.steps, e1, world), 3), unit='samples/s',
                      ms_per_step=round(e1 / args.steps * 1e3, 3), inflight=1)
    if step_copy is not None:
        torch.cuda.synchronize()
        for _ in range(max(2, args.warmup)):
            step_copy()
        e2 = parallel.timed_region(step_copy, args.steps, device)
        copy_handover = dict(value=round(parallel.throughput(args.batch * n_lanes, args.steps, e2, world), 3), unit='samples/s',
                             ms_per_step=round(e2 / args.steps * 1e3, 3), inflight=n_lanes,
                             note='the protocol of rounds 2-4: every step first copies the next pool sample into the static '
                                  'buffers of one captured forward per lane (GraphedHotPath.load: one device-to-device copy of '
                                  'all maps / points / pillars + the geometry constants)')
    return single, copy_handover


def _line(args, metric, value, elapsed, dtype, workload, extra_cfg):
    return dict(metric=metric, value=round(value, 3), unit='samples/s', n_gpus=args.gpus, steps=args.steps,
                warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True,
                scaling='weak', vs_baseline=None, dtype=dtype, data='synthetic',
                config=dict(workload=workload, batch_per_gpu=args.batch * max(1, args.inflight),
                            global_batch=args.batch * max(1, args.inflight) * args.gpus,
                            settle_ms=args.settle_ms,
                            parallelism=f'{args.gpus} replicas, sharded by sample (one process per GPU)', **extra_cfg))


def bench_forward(args, rank, world, device):
    import torch
    from deepinteraction_amd import harness, ops, parallel, synth
    shape = harness.SHAPES[args.shape]
    dtype = dict(f16=torch.float16, f32=torch.float32)[args.dtype]
    enc, dec = harness.build_models(shape, args.proposals, dtype, device)
    # weak scaling: rank r owns samples [r*batch, (r+1)*batch) of each global batch (deepinteraction_amd/parallel.py);
    # a pool of `--pool` device-resident batches per rank is cycled through the steps
    host_pool = []
    for i in range(max(1, args.pool)):
        ids = parallel.sample_ids(i, args.batch, rank, world)
        inp = synth.make_inputs(args.batch, shape, seed=parallel.sample_seed(ids[0]))
        inp['img_feats'], inp['pts_feats'] = inp['img_feats'].to(dtype).float(), inp['pts_feats'].to(dtype).float()
        host_pool.append(inp)
    dev_pool = [harness.to_device(inp, device, dtype) for inp in host_pool]
    n_pillars = [int(d['pts_metas']['pillars'].shape[0]) for d in dev_pool]

    image_net = None
    if args.from_images:
        assert not args.eager and not args.from_raw, '--from-images is a graph mode of its own'
        from deepinteraction_amd.mmdet3d_plugin import FrozenResNetFPN
        image_net = FrozenResNetFPN(out_channels=shape['c_img'], levels=(0,), dtype=dtype)
        image_net.load_mmdet_state(*image_net.synthetic_state(0)).to(device)
        H, W = shape['input_shape']
        for i, (h, d) in enumerate(zip(host_pool, dev_pool)):   # the maps every other leg of this run sees are the net's own
            cams = torch.randn(6 * args.batch, 3, H, W, generator=torch.Generator().manual_seed(1000 + i))
            d['images'] = cams.to(device, dtype).contiguous(memory_format=torch.channels_last)
            d['img_feats'] = image_net(d['images'])[0]
            h['img_feats'] = d['img_feats'].float().cpu()

    with torch.no_grad():
        if args.eager:
            it = [0]

            def step():
                d = dev_pool[it[0] % len(dev_pool)]
                it[0] += 1
                harness.forward(enc, dec, d)
            g = None
        else:
            from deepinteraction_amd.graphed import GraphedHotPath
            cap = max(range(len(dev_pool)), key=lambda i: n_pillars[i])     # the largest sample sets the capacity
            glue = None
            if args.from_points:
                from deepinteraction_amd.mmdet3d_plugin import PointGlue
                Hb, Wb = shape['bev_hw']
                rng = list(synth.PC_RANGE)
                glue = PointGlue(dict(max_num_points=20, max_voxels=(30000, 60000), point_cloud_range=rng,
                                      voxel_size=[(rng[3] - rng[0]) / Wb, (rng[4] - rng[1]) / Hb, rng[5] - rng[2]])).eval()
            n_lanes = max(1, args.inflight)
            resident = args.handover == 'resident' and not args.from_raw
            raw_pool = None
            if args.from_raw:          # NCHW-contiguous device maps, as a frozen backbone hands them over
                raw_pool = [dict(d, img_feats=d['img_feats'].contiguous(), pts_feats=d['pts_feats'].contiguous()) for d in dev_pool]
            before = [None]
            step, step_copy, step1, graphs, records, g = graphed_steps(
                lambda inp, ov: GraphedHotPath(enc, dec, inp, glue=glue, image_net=image_net, overlap=ov), dev_pool, cap, n_lanes, resident, raw_pool, bool(args.launch_threads), before)
        lidar_ms = None
        if args.from_lidar:
            assert not args.eager and shape['c_pts'] == 512, '--from-lidar feeds the 512-channel BEV input of the reference configuration'
            from deepinteraction_amd.mmdet3d_plugin import FrozenLidarBackbone
            rng = list(synth.PC_RANGE)
            grid = shape['bev_hw'][0] * 8                               # 1440 at shape R: voxels of 0.075 m
            lidar = FrozenLidarBackbone.synthetic(
                dict(max_num_points=10, max_voxels=(120000, 160000), point_cloud_range=rng,
                     voxel_size=[(rng[3] - rng[0]) / grid, (rng[4] - rng[1]) / grid, (rng[5] - rng[2]) / 41.0]),
                (41, grid, grid), device, dtype=dtype).eval()
            assert resident, '--from-lidar needs the resident hand-over (every capture keeps its own points)'

            def feed(gg):     # on the lane's stream, from the lane's thread: LiDAR branch (eager) -> the capture's static BEV input
                gg.pts_feats.copy_(lidar(gg.pts)[0])
            before[0] = feed
            step_copy = None
            lidar(graphs[0].pts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i_ in range(5):
                lidar(graphs[i_ % len(graphs)].pts)
            torch.cuda.synchronize()
            lidar_ms = (time.perf_counter() - t0) / 5 * 1e3
        settle(step, args.settle_ms)
        for _ in range(args.warmup):
            step()
        # barrier + synchronize | K steps | barrier + synchronize, MAX over ranks
        elapsed = parallel.timed_region(step, args.steps, device)

        single = copy_handover = None
        if not args.eager:
            single, copy_handover = secondary_lines(args, parallel, device, world, n_lanes, step1, step_copy if resident else None)

        # parity sample: the product's outputs on pool[0], in the benched launch mode
        product_out = None
        if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
            if g is not None:
                g.load(records[0])
                res, (img, pts) = g()[0][0], g.enc_out
            else:
                (img, pts), res = harness.forward(enc, dec, dev_pool[0])
                res = res[0][0]
            torch.cuda.synchronize()
            product_out = ((img.float().cpu(), [t.float().cpu() for t in pts]),
                           {k: v.float().cpu() for k, v in res.items()}, dec.query_labels.cpu(),
                           [m.cpu() for m in dec.on_the_image_mask], dec.top_proposals.cpu())
            if g is not None:      # the replayed graph against host launches of the same sample: must be identical
                (_, _), eg = harness.forward(enc, dec, dev_pool[0])
                torch.cuda.synchronize()
                graph_vs_eager = dict(
                    max_abs=max(float((eg[0][0][k].float().cpu() - product_out[1][k]).abs().max()) for k in product_out[1]),
                    proposals_identical=bool(torch.equal(dec.top_proposals.cpu(), product_out[4])))
        # dominant-kernel timing: HIP events right around the launch, on the launch stream, in eager forwards of
        # the same model and data (events cannot bracket one kernel inside a graph replay).  Single stream (the
        # fork/join sites off): a kernel's roofline fraction is a property of the kernel running alone; under the
        # default two-stream, two-samples-in-flight schedule its wall duration includes time-sharing with whatever
        # runs beside it (`in_step_avg_us`).
        from deepinteraction_amd import utils as di_utils

        def kernel_times(overlap):
            with di_utils.overlap(overlap):          # this thread's forwards only (utils.OVERLAP is thread-aware since round 6)
                harness.forward(enc, dec, dev_pool[0])
                torch.cuda.synchronize()
                ops.PROFILE = []
                for _ in range(args.roofline_steps):
                    harness.forward(enc, dec, dev_pool[0])
                torch.cuda.synchronize()
                got, ops.PROFILE = ops.PROFILE, None
                return got
        in_step = kernel_times(di_utils.OVERLAP)
        ops.PROFILE = kernel_times(0)
        prof, ops.PROFILE = ops.PROFILE, None
    Hi, Wi = shape['img_hw']
    n_img = 6 * args.batch
    es = 2 if dtype == torch.float16 else 4
    # round 6: both image-side attentions of a layer are ONE launch over 2 x n_img images (pair buffers, DI_PAIR_ATTN=0: two)
    pair = any(name == 'local_attn_fwd' and n == 2 * n_img for (name, n, s, e) in prof)
    n_la = 2 * n_img if pair else n_img
    alg_bytes = 4 * n_la * 128 * Hi * Wi * es
    def la_times(events, stream=False):
        # dispatch-bound events (the kernel's own begin / end time stamps, as rocprofv3 reports them) when the launch site
        # supports them; `stream`: the interval between two events recorded on the launch stream around the launch
        return [(s.stream_ms(e) if stream else s.elapsed_time(e)) * 1e-3 for (name, n, s, e) in events
                if name == 'local_attn_fwd' and n == n_la]
    durs, shared, durs_stream = la_times(prof), la_times(in_step), la_times(prof, stream=True)
    bound = all(s.dispatch_bound() for (name, n, s, e) in prof if name == 'local_attn_fwd' and n == n_la)
    avg = sum(durs) / max(len(durs), 1)
    achieved = alg_bytes / avg / 1e9 if durs else None
    # The streaming floor of THIS box for the launch's bytes: an element-wise kernel that reads three maps of the launch's size
    # and writes one (torch.addcmul), cold inputs (three rotating sets > the 256 MB Infinity Cache), graph replay, HIP events.
    # The fit of rounds 3-4 (launch time = 26 us + 0.047 us per MB through the vector L1) says the window attention cannot go
    # below this figure: `frac_of_stream_floor` = floor / launch time is the fraction of the attainable it reaches.
    stream_floor_us = None
    if durs and device.type == 'cuda' and dtype == torch.float16:
        gq = torch.Generator(device=device).manual_seed(0)
        mk = lambda: torch.randn(n_la, 128, Hi, Wi, device=device, generator=gq).half().contiguous(memory_format=torch.channels_last)
        sets = [(mk(), mk(), mk()) for _ in range(3)]
        outs = [torch.empty_like(sets[0][0]) for _ in range(3)]
        f3 = lambda i: torch.addcmul(sets[i][0], sets[i][1], sets[i][2], out=outs[i])
        f3(0)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for r_ in range(9):
                f3(r_ % 3)
        gr.replay()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5):
            gr.replay()
        ev1.record()
        torch.cuda.synchronize()
        stream_floor_us = ev0.elapsed_time(ev1) / 45 * 1e3
        del gr, sets, outs
    pmc = pmc_file('pmc_local_attn.json')
    roofline = dict(bound='hbm', kernel=f'di_local_attn_fwd, image side {n_la}x{Hi}x{Wi}' + (' (I_IML and P2I of a layer in ONE launch)' if pair else '') + ', 9x9, C=128 '
                                        f'({ops.local_attention_kernel_name()})',
                    achieved=None if achieved is None else round(achieved, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                    traffic=pmc.get('hbm_bytes_per_launch'), mfma_busy=pmc.get('mfma_busy'),
                    lds_busy=pmc.get('lds_busy'), pmc_source=pmc.get('source'),
                    pmc_note='traffic / mfma_busy / lds_busy are read from the committed rocprofv3 --pmc session named in pmc_source '
                             '(profiles/pmc_local_attn.json), not measured by this run; achieved / frac / avg_launch_us are live',
                    avg_launch_us=round(avg * 1e6, 2), launches=len(durs), algorithmic_bytes=alg_bytes,
                    stream_event_avg_us=round(sum(durs_stream) / max(len(durs_stream), 1) * 1e6, 2),
                    stream_floor_us=None if stream_floor_us is None else round(stream_floor_us, 2),
                    frac_of_stream_floor=None if not stream_floor_us or not durs else round(stream_floor_us / (avg * 1e6), 4),
                    in_step_avg_us=round(sum(shared) / max(len(shared), 1) * 1e6, 2),
                    timed_in=f'{args.roofline_steps} eager single-stream forwards right after the timed region; avg_launch_us: HIP events '
                             + ('BOUND TO THE DISPATCH on the launch stream (hipExtLaunchKernelGGL start / stop events: the '
                                'kernel\'s own begin / end time stamps, the figure rocprofv3 --kernel-trace reports; rounds 1-4 '
                                'reported stream_event_avg_us)' if bound else 'recorded on the launch stream around the launch')
                             + '; stream_event_avg_us: two events recorded on the stream around the launch (adds the 2-3 us of '
                               'their own packets); in_step_avg_us: the same with the two-stream schedule')
    out = _line(args, 'samples/sec forward (Fusion_0075 synthetic)',
                parallel.throughput(args.batch * max(1, args.inflight), args.steps, elapsed, world), elapsed,
                'f16' if dtype == torch.float16 else 'f32',
                ('frozen LiDAR branch run eagerly on the lane\'s stream before every replay (HIP voxeliser, sparse 3-D encoder on csrc/sparse_conv.hip, SECOND + SECONDFPN through torch / MIOpen) + ' if args.from_lidar else '') +
                ('frozen ResNet-50 + FPN image network (torch / MIOpen) + ' if args.from_images else '') +
                'Full MMRI encoder (2 layers) + MMPI decoder forward, '
                f'Fusion_0075_refactor shapes (shape {args.shape}), random-init weights',
                dict(num_proposals=args.proposals, pillars=n_pillars, pool=len(dev_pool), inflight=max(1, args.inflight),
                     from_points=bool(args.from_points), from_raw=bool(args.from_raw), from_images=bool(args.from_images),
                     from_lidar=bool(args.from_lidar), lidar_branch_ms=None if lidar_ms is None else round(lidar_ms, 3),
                     launch='eager' if args.eager else launch_text(resident),
                     handover=None if args.eager else ('resident' if resident else 'copy'),
                     launch_threads=None if args.eager else (n_lanes if args.launch_threads and n_lanes > 1 else 1),
                     lane_captures=None if args.eager else ('single-stream' if n_lanes > 1 and 'DI_OVERLAP' not in os.environ
                                                            else 'with the forward\'s fork / join branches'),
                     graph_nodes=None if g is None else g.num_nodes()))
    # the cross-attention kernels BASELINE names, each against its own byte roofline: live timing of
    # the same profiled forwards (both timers), the counters from the committed rocprofv3 --pmc passes of the same forward
    from deepinteraction_amd import measure
    if durs and dtype == torch.float16 and args.shape != 'TINY':
        pm0 = dev_pool[0]['pts_metas']
        b0, b1 = (0, pm0['pillars'].shape[0]) if args.batch == 1 else (0, int((pm0['pillar_coors'][:, 0] == 0).sum()))
        from deepinteraction_amd.geometry import SampleGeometry
        Hb, Wb = shape['bev_hw']
        g0 = SampleGeometry(dev_pool[0]['img_metas'][0], (Hi, Wi), device)
        kt = ops.i2p_key_table(pm0['pillars'][b0:b1], pm0['pillar_coors'][b0:b1], pm0['pillars_num_points'][b0:b1], g0.lidar2img,
                               g0.aug_rev, g0.ori_hw, (Hi, Wi), (Hb, Wb), dense=False)
        n_keys = int(kt.table[:Hb * Wb * 4].view(torch.int32).sum())
        alg = measure.algorithmic_bytes(shape, 1, es, n_keys)      # per launch: the encoder launches these per SAMPLE, except ...
        alg_b = measure.algorithmic_bytes(shape, args.batch, es, n_keys)   # ... the window attentions and 1x1 chains (whole batch)
        summ = measure.committed()
        ks = [
            measure.kernel_row('window attention, image side (I_IML + P2I' + (' in one launch: 2' if pair else ': 4') + ' launches / forward)',
                               roofline['kernel'], prof, 'local_attn_fwd', n_la, (2 if pair else 1) * alg_b['local_attn_img'],
                               measure.pmc_row(summ, 'local_attn_ring_kernel')),
            measure.kernel_row('window attention, BEV side (P_IML: 2 launches / forward)', 'di_local_attn_fwd, 180x180 (m2 generation)',
                               prof, 'local_attn_fwd', args.batch, alg_b['local_attn_bev'], measure.pmc_row(summ, 'local_attn_m2_kernel')),
            measure.kernel_row('pillar attention image -> BEV (MMRI_I2P: 2 launches / forward and sample)',
                               'di_i2p_attn_dense_fwd (matrix-core stream kernel, csrc/i2p_dense.hip)' if ops.I2P_DENSE else
                               'di_i2p_attn_fwd (wave per cell, csrc/cross_modal.hip)', prof, 'i2p_attn_fwd', 6, alg['i2p_attn'],
                               measure.pmc_row(summ, 'attn_dense_kernel') or measure.pmc_row(summ, 'i2p_attn_kernel')),
            measure.kernel_row('BEV -> image warp + K / V projection (BEVWarp + P2I keys / values: 2 launches / forward and sample)',
                               'di_pointwise_multi_warp_fwd', prof, 'pointwise_multi_warp', 6, alg['warp_project_kv'],
                               measure.pmc_row(summ, 'pointwise_multi_kernel<5, true>')),
            measure.kernel_row('q / k / v / P2I-query projections of the image map (2 launches / forward)', 'di_pointwise_multi_fwd, 4 chains',
                               prof, 'pointwise_multi', (n_img, 4), alg_b['pointwise_multi_img'],
                               measure.pmc_row(summ, 'pointwise_multi_kernel<5, false>')),
            measure.kernel_row('out_proj + integration chain, image side (2 launches / forward)', 'di_pointwise_chain_masked_fwd <256, 256>',
                               prof, 'pointwise_chain', n_img * 512, alg_b['pointwise_chain_img'],
                               measure.pmc_row(summ, 'pointwise_chain_kernel<256, 256>'), size='large'),
        ]
        for k in ks:
            k['key_count'] = n_keys if 'pillar' in k['name'] else None
        roofline['kernels'] = ks
        ring = measure.pmc_row(summ, 'local_attn_ring_kernel')
        if ring is not None and ring.get('hbm_bytes_per_launch'):
            # the committed counters of THIS round's launch shape (one launch for both image-side attentions of a layer)
            roofline.update(traffic=ring['hbm_bytes_per_launch'], mfma_busy=ring.get('mfma_busy'), lds_busy=ring.get('lds_busy'),
                            valu_busy=ring.get('valu_busy'), pmc_source=summ.get('source'),
                            pmc_note='traffic / mfma_busy / valu_busy / lds_busy are read from the committed rocprofv3 --pmc passes named '
                                     'in pmc_source (profiles/forward_roofline.json), not measured by this run; achieved / frac / '
                                     'avg_launch_us are live')
        per_fwd = (4 * alg_b['local_attn_img'] + 2 * alg_b['local_attn_bev'] + 2 * args.batch * alg['i2p_attn']
                   + 2 * args.batch * alg['warp_project_kv'] + 2 * alg_b['pointwise_multi_img'] + 2 * alg_b['pointwise_chain_img'])
        roofline['forward'] = measure.forward_block(summ, per_fwd)
    from deepinteraction_amd import _lib
    roofline['ring_spin_timeouts'] = int(_lib.lib().di_local_attn_ring_timeouts(None))     # bounded flag spins that gave up: must be 0
    assert roofline['ring_spin_timeouts'] == 0, 'the ring window-attention kernel gave up on a flag spin: results are invalid'
    out['roofline'] = roofline
    if single is not None:
        out['single_sample'] = single
    if copy_handover is not None:
        out['copy_handover'] = copy_handover
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        same = ({k: v.float().cpu() for k, v in enc.state_dict().items()},
                {k: v.float().cpu() for k, v in dec.state_dict().items()})
        # the model as built, before precision.half_maps_ rounded its map side: same seed, same init, float32 on the host
        e32, d32 = harness.build_models(shape, args.proposals, torch.float32, 'cpu')
        state = (e32.state_dict(), d32.state_dict())
        conditioned = None
        if dtype == torch.float16 and product_out is not None:
            import copy
            from deepinteraction_amd import precision
            dec_c = harness.condition_head(copy.deepcopy(d32))
            _, dec_cd = precision.to_inference(enc, copy.deepcopy(dec_c).to(device), dtype)
            with torch.no_grad():
                (ic, pc), rc = harness.forward(enc, dec_cd.eval(), dev_pool[0])
            torch.cuda.synchronize()
            conditioned = (dec_c.state_dict(), (None, {k: v.float().cpu() for k, v in rc[0][0].items()}, dec_cd.query_labels.cpu(),
                                                [m.cpu() for m in dec_cd.on_the_image_mask], dec_cd.top_proposals.cpu()))
        base, par = cpu_baseline(shape, args.proposals, state, host_pool[0], product_out,
                                 state_same=same if dtype == torch.float16 else None, conditioned=conditioned)
        out['cpu_baseline'] = base
        if par is not None and g is not None:
            par['graph_vs_eager'] = graph_vs_eager
        out['parity'] = par
    return out


def bench_forward_pp(args, rank, world, device):
    """BASELINE.json configs[4]: DeepInteraction++ forward (Fusion_0075_plusplus.py:210-303 neck + head)."""
    import torch
    from deepinteraction_amd import configs, ops, parallel, synth
    from deepinteraction_amd.graphed import GraphedHotPath
    from deepinteraction_amd.mmdet3d_plugin import DeepInteractionPlusPlusDecoder, FusionTransformerv4
    from deepinteraction_amd import harness
    shape = synth.SHAPE_PP if args.shape != 'TINY' else synth.SHAPE_PP_TINY
    bev = shape['bev_hw'][0]
    dtype = dict(f16=torch.float16, f32=torch.float32)[args.dtype]
    nprop = args.proposals if bev >= 100 else 24
    # fp16 = the mixed mode of precision.half_maps_ (fp16 neck + heat-map heads, float32 token path in the head), as v1
    enc, dec = harness.build_models_pp(shape, nprop, dtype, device)
    host_pool = []
    for i in range(max(1, args.pool)):
        inp = synth.make_inputs_pp(args.batch, shape, seed=parallel.sample_seed(parallel.sample_ids(i, args.batch, rank, world)[0]))
        inp['img_feats'] = [f.to(dtype).float() for f in inp['img_feats']]
        inp['pts_feats'] = [f.to(dtype).float() for f in inp['pts_feats']]
        host_pool.append(inp)
    pool = [harness.to_device_pp(inp, device, dtype) for inp in host_pool]
    n_pillars = [int(d['pts_metas']['pillars'].shape[0]) for d in pool]
    image_net = None
    if args.from_images:        # the ++ image side (the plugin's Swin-T + FPN, levels 0-1) inside the captured forward
        assert not args.eager, '--from-images is a graph mode'
        from deepinteraction_amd.mmdet3d_plugin import FrozenSwinFPN
        image_net = FrozenSwinFPN(out_channels=shape['c_img'], levels=(0, 1), dtype=dtype)
        image_net.load_mmdet_state(*image_net.synthetic_state(0)).to(device)
        H, W = shape['input_shape']
        for i, d in enumerate(pool):
            cams = torch.randn(6 * args.batch, 3, H, W, generator=torch.Generator().manual_seed(1000 + i))
            d['images'] = cams.to(device, dtype).contiguous(memory_format=torch.channels_last)
            d['img_feats'] = list(image_net(d['images']))

    def eager(d):
        im, p = enc(d['img_feats'], d['pts_feats'], d['img_metas'], dict(d['pts_metas']))
        return dec(p, im, d['img_metas'])
    with torch.no_grad():
        it = [0]
        if args.eager:
            g = None

            def step():
                eager(pool[it[0] % len(pool)])
                it[0] += 1
        else:
            cap = max(range(len(pool)), key=lambda i: n_pillars[i])
            n_lanes = max(1, args.inflight)
            resident = args.handover == 'resident'
            # as the v1 line: N independent captured forwards in flight, own sample each
            step, step_copy, step1, graphs, records, g = graphed_steps(
                lambda inp, ov: GraphedHotPath(enc, dec, inp, image_net=image_net, overlap=ov), pool, cap, n_lanes, resident, None, bool(args.launch_threads))
        settle(step, args.settle_ms)
        for _ in range(args.warmup):
            step()
        elapsed = parallel.timed_region(step, args.steps, device)
        single = copy_handover = None
        if not args.eager:
            single, copy_handover = secondary_lines(args, parallel, device, world, n_lanes, step1, step_copy if resident else None)
        product_out = graph_vs_eager = None
        want_cpu = rank == 0 and args.gpus == 1 and not args.no_cpu_baseline and not args.from_images
        if want_cpu:               # the product's outputs on pool[0], in the benched launch mode
            def outputs(res):
                return ({k: v.float().cpu() for k, v in res[0][0].items()}, dec.query_labels.cpu(),
                        [m.cpu() for m in dec.on_the_image_mask], dec.top_proposals.cpu())
            if g is not None:
                g.load(records[0])
                res = g()
                img, pts = g.enc_out
            else:
                img, pts = enc(pool[0]['img_feats'], pool[0]['pts_feats'], pool[0]['img_metas'], dict(pool[0]['pts_metas']))
                res = dec(pts, img, pool[0]['img_metas'])
            torch.cuda.synchronize()
            product_out = ((img.float().cpu(), [t.float().cpu() for t in pts]),) + outputs(res)
            if g is not None:
                eg = outputs(eager(pool[0]))
                graph_vs_eager = dict(max_abs=max(float((eg[0][k] - product_out[1][k]).abs().max()) for k in eg[0]),
                                      proposals_identical=bool(torch.equal(eg[3], product_out[4])))
        eager(pool[0])
        torch.cuda.synchronize()
        ops.PROFILE = []
        for _ in range(args.roofline_steps):
            eager(pool[0])
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    Hi, Wi = shape['img_hw']
    es = 2 if dtype == torch.float16 else 4
    nq = 6 * args.batch * Hi * Wi
    S = 6 * args.batch * (Hi * Wi + (Hi // 2) * (Wi // 2))
    alg = (S * 128 + nq * (8 * 2 * 4 * 3 + 128)) * es        # value + packed offsets/logits + output
    durs = [s.elapsed_time(e) * 1e-3 for (name, n, s, e) in prof if name == 'ms_deform_attn_fwd' and n == nq]
    durs_stream = [s.stream_ms(e) * 1e-3 for (name, n, s, e) in prof if name == 'ms_deform_attn_fwd' and n == nq]
    # the image self-attention (2 levels) and the P2I cross attention (1 level) share nq: the 2-level one is slower
    durs, durs_stream = sorted(durs)[len(durs) // 2:], sorted(durs_stream)[len(durs_stream) // 2:]
    avg = sum(durs) / max(len(durs), 1)
    out = _line(args, 'samples/sec forward (Fusion_0075_plusplus synthetic)',
                parallel.throughput(args.batch * max(1, args.inflight), args.steps, elapsed, world), elapsed,
                'f16' if dtype == torch.float16 else 'f32',
                'DeepInteraction++ forward: FusionTransformerv4 neck (2 layers) + DeepInteractionPlusPlusDecoder, '
                'Fusion_0075_plusplus shapes (2 image levels 112x200 / 56x100, BEV 180x180), random-init weights',
                dict(num_proposals=args.proposals, pillars=n_pillars, pool=len(pool), from_images=bool(args.from_images),
                     inflight=max(1, args.inflight),
                     launch='eager' if args.eager else launch_text(resident),
                     handover=None if args.eager else ('resident' if resident else 'copy'),
                     launch_threads=None if args.eager else (n_lanes if args.launch_threads and n_lanes > 1 else 1),
                     lane_captures=None if args.eager else ('single-stream' if n_lanes > 1 and 'DI_OVERLAP' not in os.environ
                                                            else 'with the forward\'s fork / join branches'),
                     graph_nodes=None if g is None else g.num_nodes()))
    out['roofline'] = dict(bound='hbm', kernel='pp::ms_deform_attn_hm_kernel<2, 4>, image self-attention (2 levels, 134 400 queries): '
                                               'head-major value map (bs, 8, S, 16), lanes = (query, head, corner column, channel half), '
                                               'quad-shared geometry (round 5; rounds 1-4: pp::ms_deform_attn_kernel on channels-last values)',
                           achieved=round(alg / avg / 1e9, 1) if durs else None, peak=HBM_PEAK_GBS, unit='GB/s',
                           frac=round(alg / avg / 1e9 / HBM_PEAK_GBS, 4) if durs else None,
                           traffic=pmc_file('pmc_ms_deform_attn.json').get('hbm_bytes_per_launch'),
                           avg_launch_us=round(avg * 1e6, 2), launches=len(durs), algorithmic_bytes=alg,
                           stream_event_avg_us=round(sum(durs_stream) / max(len(durs_stream), 1) * 1e6, 2),
                           timed_in='eager forwards after the timed region; avg_launch_us from HIP events bound to the dispatch '
                                    '(the kernel\'s own time stamps, as rocprofv3 reports them), stream_event_avg_us from events '
                                    'recorded on the stream around the launch')
    if single is not None:
        out['single_sample'] = single
    if copy_handover is not None:
        out['copy_handover'] = copy_handover
    if want_cpu:
        e32, d32 = harness.build_models_pp(shape, nprop, torch.float32, 'cpu')      # same seed / init, before half_maps_
        base, par = cpu_baseline_pp(shape, nprop, (e32.state_dict(), d32.state_dict()), host_pool[0], product_out)
        if par is not None and graph_vs_eager is not None:
            par['graph_vs_eager'] = graph_vs_eager
        out['cpu_baseline'], out['parity'] = base, par
    return out


if __name__ == '__main__':
    main()
