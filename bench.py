#!/usr/bin/env python
"""Headline benchmark: SR train images/sec (4x, 128->512) of the SRN generator step on MI355X.

Workload at N=1 (BASELINE.json configs[1]): RRDBNet nf=64 nb=23 (ESRGAN), batch 16 of 128x128 LR crops,
generator-only L1 step (forward + backward + Adam + weight repack), synthetic torch.rand data
(Generator seed 1234 + rank), kaiming x0.1 weights under torch.manual_seed(0).  With --gpus N the same
per-GPU batch runs on every rank (weak scaling) with an RCCL all-reduce of the gradients.

Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel (3x3 dense-block conv, Cout=32, bf16 MFMA)
timed in isolation with HIP events on the launch stream; `cpu_baseline` = the oracle (fp32 PyTorch restatement
of the reference step) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
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
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def make_dasr_opt(nf, nb, fs):
    o = make_opt(nf, nb)
    o.update(model='DASR', multiweights=True)
    o['path'].update(pretrain_model_D_target=None, pretrain_model_D_source=None)
    o['network_D'] = {'which_model_D': 'discriminator_patch', 'norm_type': 'Batch', 'act_type': 'leakyrelu', 'mode': 'CNA', 'nf': 64,
                      'in_nc': 9 if fs == 'wavelet' else 3, 'n_layers': 2}
    o['train'].update({'lr_G': 1e-4, 'weight_decay_G': 0, 'beta1_G': 0.9, 'lr_D': 1e-4, 'weight_decay_D': 0, 'beta1_D': 0.9, 'fs': fs,
                       'fs_kernel_size': 9, 'norm': True, 'sup_LL': True, 'pixel_LL_weight': 1, 'feature_criterion': 'l1',
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


def roofline_dominant_kernel(model, plans, reps=25):
    """Time the Cout=32 dense-block conv kernel (conv1..conv4 of one RDB: Cin 64/96/128/160) the way the step runs it:
    one launch sequence per concurrently processed sub-batch, each on its own stream, bracketed by HIP events on the
    launching stream.  `achieved` is the rate the chip sustains on this kernel = (concurrent launches x algorithmic
    FLOPs per launch) / average launch duration; with 2 sub-batch streams two launches are in flight at any time, so
    `avg_launch_us` is what rocprofv3 reports per launch (profiles/) and achieved = launches_concurrent * flops / it."""
    from dasr_amd.engine import OpList
    from dasr_amd import _lib
    ols, flops = [], 0.0
    for plan in plans:
        convs = [o for o in plan.fwd.ops if o.op == _lib.OP_CONV and o.conv.prec == 1 and o.conv.cout == 32][:4]
        ol = OpList()
        for _ in range(reps):       # one executor call per stream: the measurement is GPU-bound, not host-launch-bound
            for o in convs:
                ol.add(o)
        flops += sum(2.0 * o.conv.N * o.conv.Hout * o.conv.Wout * 9 * o.conv.cin * o.conv.cout for o in convs)
        ols.append(ol)
    streams = [torch.cuda.current_stream()] if len(plans) == 1 else [torch.cuda.Stream() for _ in plans]
    cur = torch.cuda.current_stream()

    def run_all():
        for st, ol in zip(streams, ols):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                ol.run()
        for st in streams:
            cur.wait_stream(st)

    run_all()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run_all()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps           # one pass of the 4 convs on every stream
    achieved = flops / (ms * 1e-3) / 1e12
    k = len(plans)
    traffic = None
    try:  # HBM bytes per launch from the committed PMC pass (bench.py cannot collect counters itself), scaled to this launch shape
        pm = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_dominant_kernel.json')))
        per_img = (pm['fetch_size_kb_raw'] * pm['fetch_correction_gfx950'] + pm['write_size_kb']) * 1024.0 / pm['images_per_launch']
        traffic = round(per_img * plans[0].N)
    except Exception:
        pass
    return {'bound': 'mfma', 'achieved': round(achieved, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
            'kernel': 'conv_glds_kernel<1,67> (dense-block conv1-4 forward, Cout=32, bf16 MFMA, LDS-DMA staging; <1,68> is its data-gradient twin)',
            'launches_concurrent': k, 'avg_launch_us': round(ms * 1e3 / 4, 1), 'flops_per_launch': flops / (4 * k),
            'note': 'achieved = launches_concurrent * flops_per_launch / avg_launch_us (sub-batch streams overlap launches)'}


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
    from oracle import trainers
    torch.set_num_threads(threads)
    opt = make_opt(nf, nb)
    torch.manual_seed(0)
    t = trainers.SRTrainer(opt)
    g = torch.Generator().manual_seed(1234)
    small = {'LR': torch.rand(1, 3, 32, 32, generator=g), 'HR': torch.rand(1, 3, 128, 128, generator=g)}
    data = {'LR': torch.rand(1, 3, lr_size, lr_size, generator=g), 'HR': torch.rand(1, 3, 4 * lr_size, 4 * lr_size, generator=g)}
    t.feed_data(small)
    t.optimize_parameters(1)  # warm-up (thread pools, allocator) on a small crop
    t.feed_data(data)
    t0 = time.time()
    t.optimize_parameters(2)
    print(json.dumps({'dt': time.time() - t0}))


def cpu_baseline(nf, nb, lr_size, timeout=150):
    """oracle (port of the reference step) on the host cores, bounded sample: batch 1, small warm-up + 1 timed step.
    Runs in a subprocess with a timeout so a slow/oversubscribed host cannot stall the GPU benchmark."""
    import subprocess
    threads = min(int(os.environ.get('DASR_CPU_THREADS', '64')), _usable_cores())
    code = 'import sys; sys.path.insert(0, %r); import bench; bench._cpu_baseline_worker(%d, %d, %d, %d)' % (ROOT, nf, nb, lr_size, threads)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, timeout=timeout, env=env, text=True)
        dt = json.loads(out.stdout.strip().splitlines()[-1])['dt']
    except Exception as e:  # timeout or failure: report, do not fail the GPU measurement
        return {'value': None, 'unit': 'images/s', 'cores': threads, 'kind': 'port', 'sample': 'cpu baseline did not finish: %r' % (e,)}
    return {'value': round(1.0 / dt, 4), 'unit': 'images/s', 'cores': threads, 'kind': 'port',
            'sample': 'oracle SRTrainer nf%d nb%d, batch 1 x %dx%d LR, small warm-up + 1 timed step (%.1f s), fp32 torch CPU, %d threads'
                      % (nf, nb, lr_size, lr_size, dt, threads)}


def sweep(model, data, a):
    """interleaved A/B of the kernel variants: isolated dense-conv microbench + whole-step time"""
    from dasr_amd import _lib
    from dasr_amd.engine import OpList
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
    plan = (getattr(model, '_out_plans', None) or [model.netG.plan(a.batch, a.lr_size, a.lr_size)])[0]

    def time_ops(ops, reps=10):
        ol = OpList()
        for o in ops:
            ol.add(o)
        ol.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ol.run()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    c32 = [o for o in plan.fwd.ops if o.op == _lib.OP_CONV and o.conv.prec == 1 and o.conv.cout == 32][:4]
    c64 = [o for o in plan.fwd.ops if o.op == _lib.OP_CONV and o.conv.prec == 1 and o.conv.cout == 64][:1]
    cst = [o for o in plan.fwd.ops if o.op == _lib.OP_CONV and o.conv.prec == 3 and o.conv.Hout == 4 * a.lr_size and o.conv.cout == 64][:1]
    wg = [o for o in plan.bwd.ops if o.op in (_lib.OP_WGRAD, _lib.OP_WGRAD_REDUCE)]
    wg_rdb = wg[-4:-2]  # an RDB group (wgrad + reduce) near the end of the backward list
    fl = lambda ops: sum(2.0 * o.conv.N * o.conv.Hout * o.conv.Wout * 9 * o.conv.cin * o.conv.cout for o in ops)
    res = {}
    for rnd in range(2):
        for v in (0, 12):
            L.dasr_set_tuning(1, v)
            ms = time_ops(c32)
            res.setdefault(('rdb32', v), []).append(fl(c32) / ms / 1e9)
        L.dasr_set_tuning(1, 12)
        L.dasr_set_tuning(4, 1)
        for v in (0, 12):
            L.dasr_set_tuning(2, v)
            ms = time_ops(c64)
            res.setdefault(('rdb64', v), []).append(fl(c64) / ms / 1e9)
        L.dasr_set_tuning(2, 12)
        L.dasr_set_tuning(4, 1)
        for v in (0, 4):
            L.dasr_set_tuning(3, v)
            ms = time_ops(cst)
            res.setdefault(('stream', v), []).append(fl(cst) / ms / 1e9)
        L.dasr_set_tuning(3, 0)
        ms = time_ops(wg_rdb[:1])
        res.setdefault(('wgrad_rdb', 0), []).append(2.0 * plan.N * a.lr_size * a.lr_size * 239616 / ms / 1e9)
        ms = time_ops(wg_rdb[1:])
        res.setdefault(('wgrad_reduce_us', 0), []).append(ms * 1e3)
    if os.environ.get('DASR_HIP_LIB'):  # instrumented build: phase stamps of one RDB wgrad launch
        import ctypes
        import numpy as np
        grid = wg_rdb[0].i[0] * wg_rdb[0].i[1]
        buf = torch.zeros(grid * 16 + 64, dtype=torch.int64, device='cuda')
        L.dasr_debug_set_wtrace.argtypes = [ctypes.c_void_p]
        L.dasr_debug_set_wtrace(buf.data_ptr())
        time_ops(wg_rdb[:1], reps=1)
        L.dasr_debug_set_wtrace(None)
        t = buf[:grid * 16].view(grid, 16).cpu().numpy().astype(np.float64)
        t = t[t[:, 0] > 0]
        wall = (t[:, 14] - t[:, 15]) * 10.0
        log('wgrad3 trace: %d workgroups (parts %d x splits %d); wall p50 %.0f ns, span %.0f ns, clock %.2f GHz' % (
            len(t), wg_rdb[0].i[0], wg_rdb[0].i[1], np.percentile(wall, 50), (t[:, 14].max() - t[:, 15].min()) * 10.0, np.median((t[:, 9] - t[:, 0]) / wall)))
        for i, nm in enumerate(['entry->first prefetch issued', 'first prefetch -> tile 2 top (2 tiles)', 'tile2: barrier 1', 'tile2: commit (ds_write)', 'tile2: barrier 2',
                                'tile2: prefetch issue', 'tile2: compute 72 MFMA', 'tiles 3.. (rest of loop)', 'result store']):
            d = t[:, i + 1] - t[:, i]
            log('  %-40s cycles p10 %8.0f p50 %8.0f p90 %8.0f' % (nm, np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
    for k, v in res.items():
        log('sweep %-16s variant %d : %s (TFLOP/s algorithmic; stream = 1/3 of MFMA rate)' % (k[0], k[1], ' '.join('%.0f' % x for x in v)))
    for wm in (1, 5, 1, 5):  # wgrad3 staging requests: staggered inside the MFMA stream (1) vs all up front (5 = bit 2 set)
        L.dasr_wgrad_set_mode(wm)
        run_steps(1)
        t0 = time.perf_counter()
        run_steps(3)
        log('sweep step time, wgrad mode %d: %.2f ms/step' % (wm, (time.perf_counter() - t0) / 3 * 1e3))
    L.dasr_wgrad_set_mode(1)
    for combo in ((12, 12, 0, 1), (13, 13, 0, 1), (0, 0, 0, 1), (12, 12, 0, 1), (13, 13, 0, 1), (0, 0, 0, 1)):
        for k, v in zip((1, 2, 3, 4), combo):
            L.dasr_set_tuning(k, v)
        run_steps(1)
        t0 = time.perf_counter()
        run_steps(3)
        log('sweep step time, tuning %s: %.2f ms/step' % (combo, (time.perf_counter() - t0) / 3 * 1e3))


def bench_dsn(a):
    """second hot path: one DSN iteration = G fwd, D fwd on [fake; real], losses, D wgrad, G bwd, 2x Adam"""
    from dasr_amd.dist import DataParallelGroup
    from dasr_amd.dsn_model import DSNModel
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dp = DataParallelGroup() if world > 1 else None
    rank = dp.rank if dp else 0
    if dp:
        torch.cuda.set_device(dp.local_rank)
    torch.manual_seed(0)
    m = DSNModel(dict(filter=a.fs, w_per=0.01, per_type='VGG'))
    if dp:
        m.dp = dp
        for net in m.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    b = a.batch if a.batch != 16 else 8
    c = 4 * a.lr_size if a.lr_size != 128 else 256
    g = torch.Generator().manual_seed(1234 + rank)
    hr, bic, real = (torch.rand(b, 3, c, c, generator=g).cuda(), torch.rand(b, 3, c // 4, c // 4, generator=g).cuda(),
                     torch.rand(b, 3, c // 4, c // 4, generator=g).cuda())
    for _ in range(a.warmup):
        m.iteration(hr, bic, real)
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m.iteration(hr, bic, real)
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp:
        dt = dp.max_over_ranks(dt)
    if rank != 0:
        return
    # De_resnet: 39.5 GMAC fwd per 256 crop, x3 for fwd+dgrad+wgrad
    tf = 3 * 2 * 39.5e-3 * (c / 256.0) ** 2
    ips = b * world * a.steps / dt
    print(json.dumps({'metric': 'DSN train crops/sec (De_resnet + FSD, %dx%d HR crops)' % (c, c), 'value': round(ips, 2), 'unit': 'images/s',
                      'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True,
                      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'split-bf16 MFMA operands (~fp32), fp32 accumulate',
                      'data': 'synthetic (torch.rand, seed 1234+rank; default nn init, seed 0; VGG16 seeded random)',
                      'config': {'workload': 'configs[4]: DSN iteration, De_resnet(8 blocks) + FSD discriminator (%s filter) + colour/texture/VGG16 losses, '
                                             'batch %d of %dx%d crops per GPU' % (a.fs, b, c, c), 'global_batch': b * world, 'parallelism': 'dp%d' % world},
                      'generator_tflops': round(ips * tf, 1), 'log': m.get_current_log()}))


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
    ap.add_argument('--tune', type=str, default='', help='kernel variant knobs, e.g. 1=1,2=0 (dasr_set_tuning key=value)')
    ap.add_argument('--sweep', action='store_true', help='A/B the conv kernel variants (stderr table), then exit')
    ap.add_argument('--model', type=str, default='sr', choices=['sr', 'dasr', 'dsn'],
                    help="sr: configs[1] generator-only step (the headline line); dasr: configs[2] full GAN step (batch = G crops per GPU); "
                         "dsn: configs[4] DSN iteration (De_resnet + FSD discriminator, --batch HR crops of 4*lr-size per GPU)")
    ap.add_argument('--fs', type=str, default='wavelet', choices=['wavelet', 'gau', 'avg_pool'])
    a = ap.parse_args()
    if a.model == 'dsn':
        return bench_dsn(a)

    from dasr_amd import options
    from dasr_amd.dist import DataParallelGroup
    from dasr_amd.models import create_model

    from dasr_amd import _lib
    for kv in [x for x in a.tune.split(',') if x]:
        k, v = kv.split('=')
        _lib.check(_lib.lib().dasr_set_tuning(int(k), int(v)), 'set_tuning')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dp = DataParallelGroup() if world > 1 else None
    rank = dp.rank if dp else 0
    if dp:
        torch.cuda.set_device(dp.local_rank)
    torch.manual_seed(0)
    dasr = a.model == 'dasr'
    model = create_model(options.dict_to_nonedict(make_dasr_opt(a.nf, a.nb, a.fs) if dasr else make_opt(a.nf, a.nb)))
    if dp:
        model.dp = dp
        for net in model.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    g = torch.Generator().manual_seed(1234 + rank)
    if dasr:
        n, s = a.batch // 2, a.lr_size
        data = {'LR_fake': torch.rand(n, 3, s, s, generator=g).cuda(), 'LR_real': torch.rand(n, 3, s, s, generator=g).cuda(),
                'HR': torch.rand(n, 3, 4 * s, 4 * s, generator=g).cuda(), 'HR_unpair': torch.rand(n, 3, 4 * s, 4 * s, generator=g).cuda(),
                'fake_w': torch.rand(n, 1, s, s, generator=g).cuda()}
    else:
        data = {'LR': torch.rand(a.batch, 3, a.lr_size, a.lr_size, generator=g).cuda(),
                'HR': torch.rand(a.batch, 3, 4 * a.lr_size, 4 * a.lr_size, generator=g).cuda()}
    log('model built')
    if a.sweep:
        return sweep(model, data, a)
    step = 0
    for _ in range(a.warmup):
        step += 1
        model.update_learning_rate()
        model.feed_data(data)
        model.optimize_parameters(step)
    torch.cuda.synchronize()
    log('warm-up done')
    if dp:
        dp.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step += 1
        model.update_learning_rate()
        model.feed_data(data)
        model.optimize_parameters(step)
    torch.cuda.synchronize()
    if dp:
        dp.barrier()
    dt = time.perf_counter() - t0
    if dp:
        dt = dp.max_over_ranks(dt)
    logd = model.get_current_log()
    loss = logd.get('l_pix', logd.get('loss/l_g_pix'))
    log('timed steps done: %.1f ms/step' % (dt / a.steps * 1e3))
    if rank != 0:
        return
    n_gpus = world
    ips = a.batch * n_gpus * a.steps / dt
    full = (a.nf == 64 and a.nb == 23 and a.lr_size == 128)
    out = {'metric': 'SR train images/sec (4x, 128->512)', 'value': round(ips, 3), 'unit': 'images/s', 'n_gpus': n_gpus,
           'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 2), 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16 MFMA operands, fp32 accumulate / fp32 residual stream',
           'data': 'synthetic (torch.rand, seed 1234+rank; kaiming x0.1 weights, seed 0)',
           'config': {'workload': ('configs[2]: full SRN GAN step (RRDBNet nf=%d nb=%d + NLayer patch-D + VGG19-54 perceptual, fs=%s), %d G crops of '
                                   '%dx%d LR per GPU (n=%d source + %d target)' % (a.nf, a.nb, a.fs, a.batch, a.lr_size, a.lr_size, a.batch // 2, a.batch // 2))
                      if dasr else
                      'configs[1]: RRDBNet nf=%d nb=%d 4x SR, batch %d of %dx%d LR per GPU, generator-only L1 step '
                      '(fwd+bwd+Adam)' % (a.nf, a.nb, a.batch, a.lr_size, a.lr_size),
                      'global_batch': a.batch * n_gpus, 'parallelism': 'dp%d' % n_gpus},
           'final_loss': loss}
    if full and not dasr:
        out['mfma_util_step'] = round(ips * TFLOP_PER_IMAGE_TRAIN / PEAK_BF16_TFLOPS, 4)
    plans = getattr(model, '_out_plans', None) or [model.netG.plan(a.batch, a.lr_size, a.lr_size)]
    out['roofline'] = roofline_dominant_kernel(model, plans)
    log('roofline done')
    if n_gpus == 1 and not a.no_cpu_baseline and not dasr:
        out['cpu_baseline'] = cpu_baseline(a.nf, a.nb, a.lr_size)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
