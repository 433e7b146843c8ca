"""Measurement plumbing shared by `bench.py` and `train_step.bench`: algorithmic bytes of the hot kernels (SURVEY 8(d):
every input read once + every output written once), the live per-kernel timing rows of a profiled eager forward
(`ops.PROFILE` events: bound to the dispatch where the launch site supports it, and recorded on the stream around the
launch - BOTH are reported, rounds stay comparable), and the committed counter figures of `profiles/forward_roofline.json`
(tools/forward_roofline.py: rocprofv3 --pmc passes + kernel trace of the same forward).  Not part of the product path."""
import json
import os

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def committed(name='forward_roofline.json'):
    p = os.path.join(ROOT, 'profiles', name)
    return json.load(open(p)) if os.path.exists(p) else {}


def pmc_row(summary, *needles):
    """The committed per-kernel row whose kernel name contains every needle (None when the summary has none)."""
    for k in summary.get('kernels', []):
        if all(n in k['kernel'] for n in needles):
            return k
    return None


def algorithmic_bytes(shape, batch, es, n_keys):
    """Bytes per LAUNCH of the cross-attention kernels BASELINE names, at `shape` (maps of 128 channels, element size `es`).
    n_keys: valid (point, camera) keys of the sample (the pillar attention reads 32 B per key)."""
    Hi, Wi = shape['img_hw']
    Hb, Wb = shape['bev_hw']
    n_img, C = 6 * batch, 128
    img_map, bev_map = n_img * C * Hi * Wi * es, batch * C * Hb * Wb * es
    return dict(
        local_attn_img=4 * img_map,                                   # q, k, v in, out
        local_attn_bev=4 * bev_map,
        # image maps once + the key stream + folded query in + context out + the valid mask
        i2p_attn=img_map + n_keys * 32 + 2 * bev_map + batch * Hb * Wb * es,
        # BEV map + completed depth in, K and V maps out (the warped map itself is never written)
        warp_project_kv=bev_map + n_img * Hi * Wi * 4 + 2 * img_map,
        pointwise_multi_img=5 * img_map,                               # the image map in, four projections out
        pointwise_chain_img=4 * img_map,                               # out_proj + integration: three maps in, one out
    )


def _avg(xs):
    return sum(xs) / len(xs) if xs else None


def kernel_row(label, what, prof, name, n, alg_bytes, pmc=None, size=None):
    """One entry of `roofline.kernels`: live timing of the launches (name, n) of the profiled forwards + the committed counters
    (`size`: 'large' / 'small' when the committed row mixes two map sizes of one kernel - its byte count is then per size, its
    duration and busy figures stay means over both and are labelled so)."""
    ev = [(s, e) for (nm, nn, s, e) in prof if nm == name and nn == n]
    disp = [s.elapsed_time(e) * 1e3 for s, e in ev]
    strm = [s.stream_ms(e) * 1e3 for s, e in ev]
    bound = bool(ev) and all(s.dispatch_bound() for s, _ in ev)
    us = _avg(disp)
    row = dict(name=label, kernel=what, launches=len(ev), algorithmic_bytes=int(alg_bytes),
               avg_launch_us=None if us is None else round(us, 2),
               stream_event_avg_us=None if not strm else round(_avg(strm), 2),
               timer='dispatch time stamps (hipExtLaunchKernelGGL events, = rocprofv3)' if bound else 'stream events around the launch',
               achieved=None if not us else round(alg_bytes / us / 1e3, 1), peak=HBM_PEAK_GBS, unit='GB/s',
               frac=None if not us else round(alg_bytes / us / 1e3 / HBM_PEAK_GBS, 4))
    if pmc is not None:
        row['pmc'] = {k: pmc.get(k) for k in ('hbm_bytes_per_launch', 'mfma_busy', 'valu_busy', 'lds_busy', 'avg_us') if k in pmc}
        if size is not None and pmc.get('hbm_bytes_per_launch_' + size) is not None:
            row['pmc']['hbm_bytes_per_launch'] = pmc['hbm_bytes_per_launch_' + size]
            row['pmc']['note'] = f'bytes: the {size} launches of this kernel; avg_us / busy figures: means over both of its map sizes'
        if row['pmc'].get('hbm_bytes_per_launch'):
            row['pmc']['traffic_over_algorithmic'] = round(row['pmc']['hbm_bytes_per_launch'] / alg_bytes, 3)
    return row


def forward_block(summary, alg_listed):
    """`roofline.forward`: the committed whole-forward figures (kernel time of the serial forward, HBM bytes by the counters)."""
    f = dict(summary.get('forward', {}))
    if not f:
        return None
    f['algorithmic_bytes_listed_kernels'] = int(alg_listed)
    f['source'] = summary.get('source')
    f['note'] = ('committed: rocprofv3 kernel trace of the serial forward + separate --pmc passes over the eager forward '
                 '(tools/forward_roofline.py); hbm_bytes_pmc = sum over kernels of (2*FETCH_SIZE + WRITE_SIZE)*1024 x launches per '
                 'forward; frac_of_peak = hbm_bytes_pmc / kernel_us / 8 TB/s; algorithmic_bytes_listed_kernels = launches x '
                 'algorithmic bytes of the kernels in roofline.kernels only')
    return f
