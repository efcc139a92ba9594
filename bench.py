#!/usr/bin/env python
"""Headline benchmark: SR train images/sec (4x, 128->512) of the SRN generator step on MI355X.

Workload at N=1 (BASELINE.json configs[1]): RRDBNet nf=64 nb=23 (ESRGAN), batch 16 of 128x128 LR crops,
generator-only L1 step (forward + backward + Adam + weight repack), synthetic torch.rand data
(Generator seed 1234 + rank), kaiming x0.1 weights under torch.manual_seed(0).

`python bench.py --gpus N` launches N ranks itself (re-exec under torch.distributed.run, one process per GPU, RCCL) when
it was not already started by a launcher; every rank runs the same per-GPU batch (weak scaling) and the gradients are
all-reduced over RCCL, overlapped with the backward kernels.  `n_gpus` is the size of the process group, not the flag.

Prints ONE JSON line (rank 0):
  * `roofline`: per-kernel rows measured live in one extra PRODUCTION step after the timed region: every launch of that step
    carries its own start/stop events on its launch stream (hipExtLaunchKernel: the dispatch's begin/end timestamps, the very
    numbers `rocprofv3 --kernel-trace --stats` prints, see profiles/), `achieved` = algorithmic FLOPs per launch / average
    launch duration; the headline row is the kernel with the largest total time.
  * `cpu_baseline`: the oracle (fp32 PyTorch restatement of the reference step) on this box's host cores, 1 warm-up + 3 timed
    steps of the bench workload at batch 1 (and configs[0] exactly), rank 0 at N=1 only.
  * `secondary`: driver-visible numbers for configs[2] (full GAN step) and configs[4] (DSN iteration) on the same box.
"""
import argparse
import json
import os
import re
import sys
import time

import torch


def log(msg):
    sys.stderr.write('[bench %.1fs] %s\n' % (time.time() - T0, msg))
    sys.stderr.flush()


T0 = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TFLOP_PER_IMAGE_TRAIN = 1.762  # 3 x 587.43 GFLOP per 128x128 LR image (nf64/nb23)
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak at 2.4 GHz (MI355X_MICROARCH.md)


def make_dasr_opt(nf, nb, fs, fea='l1'):
    o = make_opt(nf, nb)
    o.update(model='DASR', multiweights=True, allow_random_perceptual=True)   # synthetic bench: seeded VGG19 / AlexNet (says so in `data`)
    o['path'].update(pretrain_model_D_target=None, pretrain_model_D_source=None)
    o['network_D'] = {'which_model_D': 'discriminator_patch', 'norm_type': 'Batch', 'act_type': 'leakyrelu', 'mode': 'CNA', 'nf': 64,
                      'in_nc': 9 if fs == 'wavelet' else 3, 'n_layers': 2}
    o['train'].update({'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_D': 1e-4, 'weight_decay_D': 0, 'beta1_D': 0.9, 'fs': fs,
                       'fs_kernel_size': 9, 'norm': True, 'sup_LL': True, 'pixel_LL_weight': 1, 'feature_criterion': fea,
                       'feature_weight': 1, 'gan_type': 'vanilla', 'ragan': False, 'gan_H_target': 0.01, 'gan_H_source': 0,
                       'G_update_inter': 1, 'D_update_inter': 1})
    return o


def make_opt(nf, nb):
    return {'is_train': True, 'gpu_ids': [0], 'scale': 4, 'chop': False, 'val_lpips': False, 'model': 'sr', 'name': 'bench',
            'path': {'pretrain_model_G': None, 'models': '/tmp', 'training_state': '/tmp'},
            'network_G': {'which_model_G': 'RRDB_net', 'norm_type': None, 'mode': 'CNA', 'nf': nf, 'nb': nb, 'in_nc': 3,
                          'out_nc': 3, 'gc': 32, 'scale': 4},
            'train': {'lr_G': 2e-4, 'lr_scheme': 'MultiStepLR', 'lr_steps': [70000, 150000], 'lr_gamma': 0.5,
                      'pixel_criterion': 'l1', 'pixel_weight': 1.0, 'manual_seed': 0}}


# ---------------------------------------------------------------------------------------------------------------------------
# roofline: per-kernel durations of one production step
# ---------------------------------------------------------------------------------------------------------------------------
_KNAME = {'launch_glds': 'conv_glds_kernel', 'launch_ring3': 'conv_ring3_kernel', 'launch': 'conv_kernel', 'launch_wgrad': 'wgrad_kernel', 'launch_wgrad3': 'wgrad3_kernel', 'launch_wgrad3_ld': 'wgrad3_ld_kernel',
          'launch_wgrad4': 'wgrad4_kernel'}


def kernel_name(tag):
    """launcher tag (__PRETTY_FUNCTION__ of the launch template, or the kernel's own name) -> the kernel name rocprofv3 prints"""
    m = re.search(r'::(\w+)\(.*\)\s*\[(.*)\]\s*$', tag)
    if not m:
        return tag
    vals = [kv.split('=')[1].strip() for kv in m.group(2).split(',')]
    return '%s<%s>' % (_KNAME.get(m.group(1), m.group(1)), ', '.join(vals))


def profiled_steps(run_step, nsteps=1, capacity=16384):
    """run `nsteps` production steps inside a profiling session of the library; returns per-launch records + wall time"""
    import ctypes as C
    from dasr_amd import _lib
    L = _lib.lib()
    torch.cuda.synchronize()
    _lib.check(L.dasr_prof_begin(capacity), 'prof_begin')
    t0 = time.perf_counter()
    for _ in range(nsteps):
        run_step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    us, fl, by = (C.c_float * capacity)(), (C.c_double * capacity)(), (C.c_double * capacity)()
    op, tg = (C.c_int32 * capacity)(), (C.c_char_p * capacity)()
    n = L.dasr_prof_end(capacity, us, fl, by, op, tg)
    if n < 0:
        raise RuntimeError('dasr_prof_end: %d' % n)
    recs = [(tg[i].decode(), us[i], fl[i], (op[i] >> 8) & 0xff) for i in range(n)]   # (kernel tag, duration us, algorithmic flops, time bucket)
    return recs, wall, n >= capacity


def dominant_kernel_timed(run_step, raw_tag, nsteps):
    """average launch duration of ONE kernel family over `nsteps` ordinary steps: a profiling session filtered to its launches (dasr_prof_filter: two events per
    matching launch, nothing else), and the wall time of those steps.  VERDICT r05 item 3: a step in which every launch carries events ran ~6 % slower than a timed step
    and the per-launch figure of the chained launches inherited it (9.5-9.9 ms in the bench line against 8.9 ms in rocprofv3's kernel stats)."""
    from dasr_amd import _lib
    L = _lib.lib()
    m = re.search(r'(\w+)\(', raw_tag)   # launcher tags are __PRETTY_FUNCTION__ strings ("int ns::launch_x(args) [T = ...]"): filter on the function's name; kernels tag themselves
    key = (m.group(1) if m else raw_tag)[:100]
    torch.cuda.synchronize()
    _lib.check(L.dasr_prof_filter(key.encode()), 'prof_filter')
    try:
        recs, wall, _ = profiled_steps(run_step, nsteps, capacity=4096)
    finally:
        _lib.check(L.dasr_prof_filter(None), 'prof_filter')
    us = [r[1] for r in recs if r[0] == raw_tag]
    return (sum(us) / len(us) if us else None), len(us), wall / nsteps * 1e3


def roofline_from_step(run_step, peak_measured, streams, nsteps=1, timed_steps=0):
    recs, wall, truncated = profiled_steps(run_step, nsteps)
    rows, buckets = {}, {}
    for tag, us, fl, bk in recs:
        b = buckets.setdefault(bk, [0, 0.0, 0.0])
        b[0] += 1
        b[1] += us
        b[2] += fl
        r = rows.setdefault(tag, {'launches': 0, 'total_us': 0.0, 'flops': 0.0})
        r['launches'] += 1
        r['total_us'] += us
        r['flops'] += fl
    ksum = sum(r['total_us'] for r in rows.values())
    out = []
    for tag, r in sorted(rows.items(), key=lambda kv: -kv[1]['total_us']):
        if r['flops'] <= 0:
            continue
        avg = r['total_us'] / r['launches']
        fpl = r['flops'] / r['launches']
        ach = fpl / avg / 1e6  # TFLOP/s
        out.append({'kernel': kernel_name(tag), 'launches_per_step': r['launches'] // nsteps, 'avg_launch_us': round(avg, 2),
                    'flops_per_launch': fpl, 'achieved': round(ach, 1), 'frac': round(ach / PEAK_BF16_TFLOPS, 4),
                    'share_of_kernel_time': round(r['total_us'] / ksum, 4)})
    other = sum(r['total_us'] for r in rows.values() if r['flops'] <= 0)
    head = dict(out[0]) if out else {}
    head_raw = max((kv for kv in rows.items() if kv[1]['flops'] > 0), key=lambda kv: kv[1]['total_us'])[0] if out else None
    timed = None
    if head and timed_steps > 0:
        # the dominant kernel again, over ordinary steps: only ITS launches carry events (two per launch); this is the figure `achieved` / `frac` are computed from
        t_us, t_n, t_ms = dominant_kernel_timed(run_step, head_raw, timed_steps)
        if t_us:
            timed = {'avg_launch_us': round(t_us, 2), 'launches': t_n, 'steps': timed_steps, 'ms_per_step_of_those_steps': round(t_ms, 2),
                     'avg_launch_us_in_fully_instrumented_step': head['avg_launch_us']}
            head['avg_launch_us'] = round(t_us, 2)
            head['achieved'] = round(head['flops_per_launch'] / t_us / 1e6, 1)
            head['frac'] = round(head['achieved'] / PEAK_BF16_TFLOPS, 4)
    overlap = ksum / (wall * 1e6) if wall > 0 else None
    traffic, tsrc = None, None
    try:  # HBM bytes per launch of the headline kernel: from the committed PMC passes (rocprofv3 --pmc cannot run inside bench.py)
        pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        ent = pm['kernels'].get(head.get('kernel'))
        if ent:
            traffic = int((ent['fetch_size_kb_raw'] * pm['fetch_correction_gfx950'] + ent['write_size_kb']) * 1024.0)
            tsrc = pm['source']
    except Exception:
        pass
    busy = None
    try:  # MFMA-pipe busy fraction from the committed SQ counter passes (the rocprof definition of "MFMA utilisation"), single-stream launches
        pb = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_mfma_busy.json')))
        busy = {'kernel': pb['kernels'].get(head.get('kernel')), 'rrdb_trunk_time_weighted': pb['rrdb_trunk_time_weighted'], 'source': pb['source']}
    except Exception:
        pass
    roof = {'bound': 'mfma', 'achieved': head.get('achieved'), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': head.get('frac'),
            'traffic': traffic, 'traffic_source': tsrc, 'kernel': head.get('kernel'), 'avg_launch_us': head.get('avg_launch_us'),
            'flops_per_launch': head.get('flops_per_launch'),
            'peak_at_observed_clock': round(peak_measured, 1) if peak_measured else None,
            'frac_of_peak_at_observed_clock': round(head['achieved'] / peak_measured, 4) if (peak_measured and head) else None,
            'dominant_kernel_over_ordinary_steps': timed,
            'streams': streams, 'kernel_time_over_wall': round(overlap, 3) if overlap else None,
            'non_mfma_kernel_time_share': round(other / ksum, 4) if ksum else None,
            'method': 'one extra production step after the timed region in which every launch has its own start/stop events on its launch stream '
                      '(hipExtLaunchKernel = the dispatch timestamps rocprofv3 --kernel-trace reports) gives the per-kernel table and names the headline '
                      'kernel (largest total time); the headline kernel is then timed AGAIN over ordinary steps in which only its launches carry events '
                      '(dominant_kernel_over_ordinary_steps: the fully instrumented step runs ~6 % slower and inflated it); achieved = algorithmic FLOPs '
                      'per launch / that average launch duration; with streams > 1 launches of the sub-batch '
                      'streams overlap (kernel_time_over_wall = sum of launch durations / step wall time), so the chip-level rate of a kernel '
                      'is up to that factor above its per-launch rate',
            'mfma_busy_pmc': busy, 'per_kernel': out[:8], 'truncated': truncated,
            # algorithmic FLOPs of the whole profiled step (sum over all MFMA launches of what the plan builder attributes to each op)
            'tflop_per_step': round(sum(r['flops'] for r in rows.values()) / nsteps / 1e12, 3),
            'profiled_step_ms': round(wall * 1e3 / nsteps, 2)}
    # where the step's kernel time goes, by the plan builders' bucket tags (dasr_amd/engine.py BUCKETS): launches, summed launch durations (streams
    # overlap: the sum exceeds the wall time by kernel_time_over_wall), algorithmic TFLOP and the resulting rate
    from dasr_amd.engine import BUCKETS
    roof['buckets'] = [{'bucket': BUCKETS.get(k, str(k)), 'launches': v[0] // nsteps, 'kernel_ms': round(v[1] / nsteps / 1e3, 3),
                        'share': round(v[1] / ksum, 4) if ksum else None, 'tflop': round(v[2] / nsteps / 1e12, 3),
                        'tflops_rate': round(v[2] / v[1] / 1e6, 1) if v[1] > 0 and v[2] > 0 else None}
                       for k, v in sorted(buckets.items(), key=lambda kv: -kv[1][1])]
    return roof


def secondary_roofline(run_step, streams, workload=None):
    """roofline block of a secondary workload (configs[2] / configs[4]): the same live per-launch measurement as the headline, trimmed: the dominant
    kernel's per-launch rate, the six largest kernels, the algorithmic FLOPs of the step and the step-level fraction of the dense bf16 peak"""
    r = roofline_from_step(run_step, None, streams)
    keep = ('bound', 'achieved', 'peak', 'unit', 'frac', 'kernel', 'avg_launch_us', 'flops_per_launch', 'kernel_time_over_wall', 'non_mfma_kernel_time_share',
            'tflop_per_step', 'profiled_step_ms', 'buckets')
    out = {k: r.get(k) for k in keep}
    out['traffic'], out['traffic_source'] = None, None
    try:   # HBM bytes per launch of the dominant kernel from the committed PMC passes of this workload (profiles/pmc_traffic.json, `workloads`)
        pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
        ent = pm.get('workloads', {}).get(workload, {}).get(out.get('kernel'))
        if ent:
            out['traffic'] = int((ent['fetch_size_kb_raw'] * pm['fetch_correction_gfx950'] + ent['write_size_kb']) * 1024.0)
            tag = pm.get('workloads_tag')   # the secondary workloads' PMC passes may come from an earlier session than the headline's (profiles/pmc_traffic.json)
            out['traffic_source'] = ('profiles/%s_%s_pmc_traffic.txt' % (tag, workload)) if tag else \
                pm['source'].split(':')[0].replace('_pmc_traffic.txt', '_%s_pmc_traffic.txt' % workload)
    except Exception:
        pass
    out['per_kernel'] = r['per_kernel'][:6]
    if r.get('tflop_per_step') and r.get('profiled_step_ms'):
        out['mfma_util_step'] = round(r['tflop_per_step'] / (r['profiled_step_ms'] * 1e-3) / PEAK_BF16_TFLOPS, 4)
    return out


# ---------------------------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = port of the reference step) in a bounded subprocess
# ---------------------------------------------------------------------------------------------------------------------------
def _usable_cores():
    """host cores this process may really use: affinity mask, capped by the cgroup CPU quota if there is one"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


def _cpu_baseline_worker(nf, nb, lr_size, threads):
    """SURVEY 8(d) protocol: 1 warm-up + 3 timed steps; the bench workload at batch 1, then configs[0] exactly (nf32 nb4, batch 2, 64x64)"""
    from oracle import trainers
    torch.set_num_threads(threads)
    res = {}
    for name, (nf_, nb_, b, s) in (('bench', (nf, nb, 1, lr_size)), ('cfg0', (32, 4, 2, 64))):
        torch.manual_seed(0)
        t = trainers.SRTrainer(make_opt(nf_, nb_))
        g = torch.Generator().manual_seed(1234)
        data = {'LR': torch.rand(b, 3, s, s, generator=g), 'HR': torch.rand(b, 3, 4 * s, 4 * s, generator=g)}
        times = []
        for i in range(4):
            t.feed_data(data)
            t0 = time.time()
            t.optimize_parameters(i + 1)
            times.append(time.time() - t0)
        res[name] = {'batch': b, 'lr': s, 'warmup_s': times[0], 'timed_s': times[1:]}
        print(json.dumps(res))
        sys.stdout.flush()


def cpu_baseline(nf, nb, lr_size, timeout=240):
    import subprocess
    threads = min(int(os.environ.get('DASR_CPU_THREADS', '64')), _usable_cores())
    code = 'import sys; sys.path.insert(0, %r); import bench; bench._cpu_baseline_worker(%d, %d, %d, %d)' % (ROOT, nf, nb, lr_size, threads)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, timeout=timeout, env=env, text=True)
    except subprocess.TimeoutExpired as e:
        out = e
    try:
        res = json.loads((out.stdout or '').strip().splitlines()[-1])
        r = res['bench']
    except Exception as e:  # timeout before the first result or failure: report, do not fail the GPU measurement
        return {'value': None, 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'sample': 'cpu baseline did not finish: %r' % (e,)}
    mean = sum(r['timed_s']) / len(r['timed_s'])
    d = {'value': round(r['batch'] / mean, 4), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
         'sample': 'oracle SRTrainer nf%d nb%d, batch 1 x %dx%d LR (linear in the batch: the bench batch is this step repeated), 1 warm-up + %d timed '
                   'steps (%s s), fp32 torch CPU, %d threads' % (nf, nb, lr_size, lr_size, len(r['timed_s']),
                                                                 ' '.join('%.2f' % x for x in r['timed_s']), threads),
         'step_s': [round(x, 3) for x in r['timed_s']]}
    if 'cfg0' in res:
        c = res['cfg0']
        m0 = sum(c['timed_s']) / len(c['timed_s'])
        d['configs0'] = {'value': round(c['batch'] / m0, 3), 'unit': 'images/s',
                         'sample': 'configs[0] exactly: nf32 nb4, batch 2 x 64x64 LR, 1 warm-up + 3 timed steps (%s s)' % ' '.join('%.2f' % x for x in c['timed_s'])}
    return d


# ---------------------------------------------------------------------------------------------------------------------------
def sweep(model, data, a):
    """interleaved A/B of kernel variants on the whole step (dasr_set_tuning keys, see include/dasr_hip.h)"""
    from dasr_amd import _lib
    L = _lib.lib()
    step = [0]

    def run_steps(n):
        for _ in range(n):
            step[0] += 1
            model.update_learning_rate()
            model.feed_data(data)
            model.optimize_parameters(step[0])
        torch.cuda.synchronize()

    run_steps(2)
    combos = [[tuple(int(x) for x in kv.split('=')) for kv in c.split(';') if kv] for c in (a.sweep_combos or '6=0,6=1').split(',')]
    for rnd in range(a.sweep_rounds):
        for combo in combos:
            for k, v in combo:
                _lib.check(L.dasr_set_tuning(k, v), 'set_tuning %d' % k)
            run_steps(1)
            t0 = time.perf_counter()
            run_steps(3)
            log('sweep step time, tuning %s: %.2f ms/step' % (combo, (time.perf_counter() - t0) / 3 * 1e3))


_STREAMS0 = os.environ.get('DASR_STREAMS', '2')


def streams_default():
    return max(1, int(_STREAMS0))


def setup_dist(a):
    """one process per GPU.  Started by a launcher (WORLD_SIZE set): join its group.  Started directly with --gpus N > 1: re-exec
    under torch.distributed.run so that `python bench.py --gpus N` really measures N GPUs."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        import socket
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log('launching %d ranks: %s' % (a.gpus, ' '.join(cmd)))
        os.execv(sys.executable, cmd)
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (a.gpus, world))
    from dasr_amd.dist import DataParallelGroup
    dp = DataParallelGroup() if world > 1 else None
    if dp:
        torch.cuda.set_device(dp.device_index)
        import torch.distributed as dist
        assert dist.get_world_size() == a.gpus
    return dp


def timed_loop(step_fn, a, dp):
    for _ in range(a.warmup):
        step_fn()
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step_fn()
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    dt = time.perf_counter() - t0
    return dp.max_over_ranks(dt) if dp else dt


def bench_dsn(a, dp, as_secondary=False):
    """second hot path: one DSN iteration = G fwd, D fwd on [fake; real], losses, D wgrad, G bwd, 2x Adam"""
    from dasr_amd.dsn_model import DSNModel
    world = dp.world if dp else 1
    rank = dp.rank if dp else 0
    torch.manual_seed(0)
    m = DSNModel(dict(filter=a.fs, w_per=0.01, per_type=a.per_type, allow_random_perceptual=True))   # synthetic bench: seeded perceptual net (says so in `data`)
    if dp:
        m.dp = dp
        for net in m.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    b = 8 if (as_secondary or a.batch == 16) else a.batch
    c = 256 if (as_secondary or a.lr_size == 128) else 4 * a.lr_size
    g = torch.Generator().manual_seed(1234 + rank)
    hr, bic, real = (torch.rand(b, 3, c, c, generator=g).cuda(), torch.rand(b, 3, c // 4, c // 4, generator=g).cuda(),
                     torch.rand(b, 3, c // 4, c // 4, generator=g).cuda())
    dt = timed_loop(lambda: m.iteration(hr, bic, real), a, dp)
    # De_resnet: 39.5 GMAC fwd per 256 crop, x3 for fwd+dgrad+wgrad
    tf = 3 * 2 * 39.5e-3 * (c / 256.0) ** 2
    ips = b * world * a.steps / dt
    out = {'metric': 'DSN train crops/sec (De_resnet + FSD, %dx%d HR crops)' % (c, c), 'value': round(ips, 2), 'unit': 'images/s',
           'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': m.dtype_note if hasattr(m, 'dtype_note') else 'split-bf16 MFMA operands (~fp32), fp32 accumulate',
           'data': 'synthetic (torch.rand, seed 1234+rank; default nn init, seed 0; perceptual net seeded random)',
           'config': {'workload': 'configs[4]: DSN iteration, De_resnet(8 blocks) + FSD discriminator (%s filter) + colour/texture/%s losses, '
                                  'batch %d of %dx%d crops per GPU' % (a.fs, 'LPIPS(alex)' if a.per_type == 'LPIPS' else 'VGG16', b, c, c), 'global_batch': b * world, 'parallelism': 'dp%d' % world},
           'generator_tflops': round(ips * tf, 1)}
    if not as_secondary:
        out['log'] = m.get_current_log()
    if rank == 0:
        out['roofline'] = secondary_roofline(lambda: m.iteration(hr, bic, real), 1, 'dsn_lpips' if a.per_type == 'LPIPS' else 'dsn_vgg')
    else:
        m.iteration(hr, bic, real)
    return out


def bench_srn(a, dp, dasr, as_secondary=False):
    from dasr_amd import options, _lib
    from dasr_amd.models import create_model
    world = dp.world if dp else 1
    rank = dp.rank if dp else 0
    torch.manual_seed(0)
    batch = (getattr(a, 'sec_batch', None) or (32 if dasr else 16)) if as_secondary else a.batch
    model = create_model(options.dict_to_nonedict(make_dasr_opt(a.nf, a.nb, a.fs, a.fea) if dasr else make_opt(a.nf, a.nb)))
    if dp:
        model.dp = dp
        for net in model.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    g = torch.Generator().manual_seed(1234 + rank)
    s = a.lr_size
    if dasr:
        n = batch // 2
        data = {'LR_fake': torch.rand(n, 3, s, s, generator=g).cuda(), 'LR_real': torch.rand(n, 3, s, s, generator=g).cuda(),
                'HR': torch.rand(n, 3, 4 * s, 4 * s, generator=g).cuda(), 'HR_unpair': torch.rand(n, 3, 4 * s, 4 * s, generator=g).cuda(),
                'fake_w': torch.rand(n, 1, s, s, generator=g).cuda()}
    else:
        data = {'LR': torch.rand(batch, 3, s, s, generator=g).cuda(), 'HR': torch.rand(batch, 3, 4 * s, 4 * s, generator=g).cuda()}
    log('model built (%s)' % ('dasr' if dasr else 'sr'))
    if a.sweep:
        sweep(model, data, a)
        return None
    step = [0]

    def one_step():
        step[0] += 1
        model.update_learning_rate()
        model.feed_data(data)
        model.optimize_parameters(step[0])

    dt = timed_loop(one_step, a, dp)
    logd = model.get_current_log()
    loss = logd.get('l_pix', logd.get('loss/l_g_pix'))
    log('timed steps done: %.2f ms/step' % (dt / a.steps * 1e3))
    ips = batch * world * a.steps / dt
    full = (a.nf == 64 and a.nb == 23 and s == 128)
    out = {'metric': 'SR train images/sec (4x, 128->512)', 'value': round(ips, 3), 'unit': 'images/s', 'n_gpus': world,
           'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None,
           'dtype': ('f16 MFMA operands (DASR_RDB_PREC=2), fp32 accumulate / fp32 residual stream' if getattr(model.netG, 'rdb_f16', False)
                     else 'bf16 MFMA operands, fp32 accumulate / fp32 residual stream'),
           'data': 'synthetic (torch.rand, seed 1234+rank; kaiming x0.1 weights, seed 0)',
           'config': {'workload': ((getattr(a, 'sec_label', None) or 'configs[2]') + ': full SRN GAN step (RRDBNet nf=%d nb=%d + NLayer patch-D + %s perceptual, fs=%s), %d G crops of '
                                   '%dx%d LR per GPU (n=%d source + %d target)' % (a.nf, a.nb, 'LPIPS(alex)' if a.fea == 'LPIPS' else 'VGG19-54', a.fs, batch,
                                                                                  s, s, batch // 2, batch // 2))
                      if dasr else
                      'configs[1]: RRDBNet nf=%d nb=%d 4x SR, batch %d of %dx%d LR per GPU, generator-only L1 step '
                      '(fwd+bwd+Adam)' % (a.nf, a.nb, batch, s, s),
                      'global_batch': batch * world, 'parallelism': 'dp%d' % world},
           'final_loss': loss}
    if dasr:
        out['metric'] = 'SRN GAN train G-images/sec (4x, 128->512)'
    if full and not dasr:
        out['mfma_util_step'] = round(ips * TFLOP_PER_IMAGE_TRAIN / world / PEAK_BF16_TFLOPS, 4)
    if rank == 0 and not as_secondary:
        import ctypes as C
        pk = C.c_float(0.0)
        BL = _lib.bench_lib()   # probes live in libdasr_bench.so, not in the product library
        peak_measured = pk.value if BL.dasr_probe_mfma_peak(20000, C.byref(pk), None) == 0 else None
        streams = len(getattr(model, '_out_plans', None) or [0])
        out['roofline'] = roofline_from_step(one_step, peak_measured, streams, timed_steps=max(2, min(a.steps, 6)) if world == 1 else 0)   # (the other ranks run ONE matching step)
        # the same MFMA-only stream on operands that toggle (random bf16 in (-1, 1)) and on all-zero operands: the spread is the clock the power
        # management allows under that switching activity -- the ceiling an MFMA-bound kernel on real data can approach on this box
        pr = {}
        for mode, key in ((2, 'zeros'), (0, 'random_bf16')):
            if BL.dasr_probe_mfma_data(19968, mode, C.byref(pk), None) == 0:
                pr[key] = round(pk.value, 1)
        out['roofline']['mfma_only_tflops_by_operand_data'] = pr
        if pr.get('random_bf16') and out['roofline'].get('achieved'):
            # the dense-MFMA rate THIS box sustains on operands that toggle (the clock its power management allows under that switching activity): the ceiling an
            # MFMA-bound kernel on real data can approach here; `peak` / `frac` stay the guide's 2.5 PFLOP/s
            out['roofline']['peak_on_data'] = pr['random_bf16']
            out['roofline']['frac_of_peak_on_data'] = round(out['roofline']['achieved'] / pr['random_bf16'], 4)
        log('roofline done')
    elif not as_secondary:
        one_step()  # the other ranks take part in the profiled step's collectives
    if as_secondary:
        if rank == 0:
            out['roofline'] = secondary_roofline(one_step, len(getattr(model, '_out_plans', None) or [0]), 'dasr_lpips' if a.fea == 'LPIPS' else 'dasr_vgg')
        else:
            one_step()
    if not as_secondary and not dasr and world == 1 and full and not a.no_secondary:
        # The same production step (same plans, launches and bytes) on ALL-ZERO operands: what the data-dependent shader clock costs on this box
        # (profiles/r04_zero_data.txt).  Reported beside the measurement, never as `value`.
        P = model.netG.params
        saved = (P.flat.clone(), P.m.clone(), P.v.clone())
        kept = dict(data)
        try:
            for k in data:
                data[k] = torch.zeros_like(kept[k])
            P.flat.zero_(); P.m.zero_(); P.v.zero_()
            model.netG.repack()
            one_step()
            one_step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                one_step()
            torch.cuda.synchronize()
            ms0 = (time.perf_counter() - t0) / 4 * 1e3
            out['roofline']['zero_operand_step'] = {
                'ms_per_step': round(ms0, 2), 'mfma_util_step': round(batch * TFLOP_PER_IMAGE_TRAIN / (ms0 / 1e3) / PEAK_BF16_TFLOPS, 4),
                'note': 'same launches and bytes, every weight and image zero: no operand toggling, the shader clock holds; the difference to '
                        'ms_per_step is the data-dependent clock, not code'}
        finally:
            data.update(kept)
            P.flat.copy_(saved[0]); P.m.copy_(saved[1]); P.v.copy_(saved[2])
            model.netG.repack()
            del saved
    if not as_secondary and not dasr and streams_default() > 1 and len(getattr(model, '_out_plans', None) or [0]) > 1 and not a.no_secondary:
        # (only when the production schedule runs sub-batch streams: the chained-trunk schedule is single-stream already; skipped with --no-secondary = the rocprofv3 / PMC runs, so profiles/*.csv hold launches of the production schedule only)
        # the same step with ONE stream (every launch covers the whole per-GPU batch and has the chip to itself): the per-launch rates of
        # the kernels without the overlap of the sub-batch streams.  Not the production schedule: reported beside it, never as `value`.
        os.environ['DASR_STREAMS'] = '1'
        try:
            one_step()
            one_step()
            if rank == 0:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                one_step()
                torch.cuda.synchronize()
                ms1 = (time.perf_counter() - t0) * 1e3
                r1 = roofline_from_step(one_step, None, 1)
                out['roofline']['single_stream'] = {'ms_per_step': round(ms1, 2), 'per_kernel': r1['per_kernel'],
                                                    'note': 'DASR_STREAMS=1: batch-%d launches, no overlap between launches' % batch}
                # `frac` above is strictly flops / launch duration in the production schedule, where two launches share the chip; the same
                # kernel with the chip to itself:
                same = [k for k in r1['per_kernel'] if k['kernel'] == out['roofline'].get('kernel')]
                if same:
                    out['roofline']['frac_single_stream'] = same[0]['frac']
            else:
                one_step()
                one_step()
        finally:
            os.environ['DASR_STREAMS'] = str(streams_default())
    del model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=16, help='LR crops per GPU per step')
    ap.add_argument('--lr-size', type=int, default=128)
    ap.add_argument('--nf', type=int, default=64)
    ap.add_argument('--nb', type=int, default=23)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs[2] / configs[4] secondary measurements')
    ap.add_argument('--tune', type=str, default='', help='kernel variant knobs, e.g. 1=1,2=0 (dasr_set_tuning key=value)')
    ap.add_argument('--sweep', action='store_true', help='A/B kernel variants on the whole step (stderr table), then exit')
    ap.add_argument('--sweep-combos', type=str, default='', help="tuning combos, e.g. '6=0,6=1;1=13' (dasr_set_tuning key=value; ';' joins, ',' separates combos)")
    ap.add_argument('--sweep-rounds', type=int, default=3)
    ap.add_argument('--model', type=str, default='sr', choices=['sr', 'dasr', 'dsn'],
                    help="sr: configs[1] generator-only step (the headline line); dasr: configs[2] full GAN step (batch = G crops per GPU); "
                         "dsn: configs[4] DSN iteration (De_resnet + FSD discriminator, --batch HR crops of 4*lr-size per GPU)")
    ap.add_argument('--fs', type=str, default='wavelet', choices=['wavelet', 'gau', 'avg_pool'])
    ap.add_argument('--fea', type=str, default='l1', choices=['l1', 'LPIPS'], help="--model dasr: feature_criterion (l1 = VGG19-54 features, BASELINE configs[2]; LPIPS = the shipped train_DASR.json criterion)")
    ap.add_argument('--per-type', dest='per_type', type=str, default='VGG', choices=['VGG', 'LPIPS'], help='--model dsn: perceptual term')
    a = ap.parse_args()
    dp = setup_dist(a)
    rank = dp.rank if dp else 0

    from dasr_amd import _lib
    for kv in [x for x in a.tune.split(',') if x]:
        k, v = kv.split('=')
        _lib.check(_lib.lib().dasr_set_tuning(int(k), int(v)), 'set_tuning')

    if a.model == 'dsn':
        out = bench_dsn(a, dp)
    else:
        out = bench_srn(a, dp, a.model == 'dasr')
    if out is None:
        return
    default_workload = a.model == 'sr' and a.nf == 64 and a.nb == 23 and a.lr_size == 128 and a.batch == 16
    if default_workload and not a.no_secondary:
        # driver-visible numbers for the other GPU configs of BASELINE.json, measured in the same process (fewer steps)
        sec = argparse.Namespace(**vars(a))
        sec.steps, sec.warmup = max(2, a.steps // 2), 2
        out['secondary'] = []
        sec_l = argparse.Namespace(**vars(sec))
        sec_l.fea, sec_l.per_type = 'LPIPS', 'LPIPS'   # the criteria the reference's shipped configs / CLI defaults select
        # the shape the reference's SHIPPED training config runs (codes/SRN/options/train/train_DASR.json:21-22: batch_size 8 -> 2 n = 16 G crops, HR_size 128 -> 32 x 32 LR;
        # LPIPS criterion): 32 tiles of 16 x 32 pixels per conv launch -- what a user who drops that JSON in gets
        sec_ship = argparse.Namespace(**vars(sec_l))
        sec_ship.lr_size, sec_ship.sec_batch, sec_ship.sec_label = 32, 16, 'shipped shape (train_DASR.json: batch_size 8, HR_size 128)'
        sec_ship.steps, sec_ship.warmup = max(16, 2 * a.steps), 4   # (a 7-ms step: four timed steps are one host hiccup away from twice the number -- seen once, 15.1 ms, round 6)
        for fn in (lambda: bench_srn(sec, dp, True, as_secondary=True), lambda: bench_dsn(sec, dp, as_secondary=True),
                   lambda: bench_srn(sec_l, dp, True, as_secondary=True), lambda: bench_dsn(sec_l, dp, as_secondary=True),
                   lambda: bench_srn(sec_ship, dp, True, as_secondary=True)):
            try:
                r = fn()
                out['secondary'].append({k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'dtype', 'config', 'roofline') if k in r})
            except Exception as e:  # a secondary measurement must not take the headline line down
                out['secondary'].append({'error': repr(e)})
        log('secondary done')
    if rank != 0:
        return
    if out['n_gpus'] == 1 and not a.no_cpu_baseline and a.model == 'sr':
        out['cpu_baseline'] = cpu_baseline(a.nf, a.nb, a.lr_size)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
