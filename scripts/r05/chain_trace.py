"""Where a chained trunk launch spends its cycles: per-item phase accumulators of conv_chain_kernel (round-4 form) and
conv_chain2_kernel (round-5 form) from the -DDASR_TRACE build, plus the launch durations of both forms in the production build.

  python -m dasr_amd.build --trace && DASR_HIP_LIB=dasr_amd/libdasr_hip_trace.so python scripts/r05/chain_trace.py [--n 16] [--nb 23]
  python scripts/r05/chain_trace.py --time-only      (production library: launch durations only)

Phases per item (one layer of one tile), cycles of s_memtime as seen by thread 0 (wave 0, which also polls the neighbour flags):
  entry   : layer entry -> chunk 0 requested (incl. the up-front neighbour wait of conv1 of a dense block)
  chunk0  : chunk 0 requested -> in LDS and the workgroup through its barrier (0 when the previous item requested it early)
  loop    : main loop, all chunks (incl. `poll`: the neighbour-flag poll in front of the first dependent chunk)
  epilogue: bias / mask / residual loads, arithmetic, stores ISSUED (and, round-5 form, the early chunk-0 request of the next item)
  publish : stores acknowledged + barrier + flag store
"""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CH_BASE = 1 << 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=16)
    ap.add_argument('--nb', type=int, default=23)
    ap.add_argument('--lr', type=int, default=128)
    ap.add_argument('--time-only', action='store_true')
    ap.add_argument('--stagger', type=str, default='', help='comma list of start offsets (us) of the odd images: launch durations of both chains per value (form 1)')
    ap.add_argument('--forms', type=str, default='1,2')
    a = ap.parse_args()
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    os.environ['DASR_STREAMS'] = '1'
    import numpy as np
    import torch
    import bench
    from dasr_amd import _lib, options
    from dasr_amd.engine import OpList
    from dasr_amd.models import create_model
    L = _lib.lib()
    traced = bool(os.environ.get('DASR_HIP_LIB')) and not a.time_only
    torch.manual_seed(0)
    m = create_model(options.dict_to_nonedict(bench.make_opt(64, a.nb)))
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(a.n, 3, a.lr, a.lr, generator=g).cuda(), 'HR': torch.rand(a.n, 3, 4 * a.lr, 4 * a.lr, generator=g).cuda()}
    if a.stagger:
        _lib.check(L.dasr_set_tuning(7, 1))
        for st in (1, 2):
            m.update_learning_rate()
            m.feed_data(data)
            m.optimize_parameters(st)
        torch.cuda.synchronize()
        plan = m._out_plans[0]
        for rnd in (1, 2):
            for us_off in [int(x) for x in a.stagger.split(',')]:
                _lib.check(L.dasr_set_tuning(8, us_off))
                row = []
                for ch in (plan.chain, plan.chain_b):
                    one = OpList()
                    one.add(ch.op())
                    one.run()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        one.run()
                    e1.record()
                    torch.cuda.synchronize()
                    row.append(e0.elapsed_time(e1) / 4 * 1e3)
                print('stagger %4d us: forward chain %.1f us, data-gradient chain %.1f us per launch' % (us_off, row[0], row[1]))
                sys.stdout.flush()
        _lib.check(L.dasr_set_tuning(8, 0))
        plan.check_chain()
        return
    for form in [int(x) for x in a.forms.split(',')]:
        ntiles = a.n * ((a.lr + 15) // 16) * ((a.lr + 31) // 32)
        if form == 1 and ntiles != 512:
            continue
        _lib.check(L.dasr_set_tuning(7, form))
        for st in (1, 2):
            m.update_learning_rate()
            m.feed_data(data)
            m.optimize_parameters(st)
        torch.cuda.synchronize()
        plan = m._out_plans[0]
        assert plan.chain is not None and plan.chain_b is not None
        for name, ch in (('forward chain', plan.chain), ('data-gradient chain', plan.chain_b)):
            one = OpList()
            one.add(ch.op())
            one.run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                one.run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 3 * 1e3
            print('form %d  %-20s %d layers, %d tiles: %.1f us per launch = %.2f us per layer, %.0f TFLOP/s' % (form, name, ch.n, ntiles, us, us / ch.n, ch.flops / us / 1e6))
            if traced:
                buf = torch.zeros(CH_BASE + 512 * 64 + 64, dtype=torch.int64, device='cuda')
                L.dasr_debug_set_trace.argtypes = [ctypes.c_void_p]
                _lib.check(L.dasr_debug_set_trace(buf.data_ptr()))
                one.run()
                torch.cuda.synchronize()
                _lib.check(L.dasr_debug_set_trace(None))
                t = buf[CH_BASE:CH_BASE + 512 * 64].view(512, 8, 8).cpu().numpy().astype(np.float64)
                raw = buf[CH_BASE:CH_BASE + 512 * 64].view(512, 64).cpu().numpy()
                hw, tick, xcc = raw[:, 56], raw[:, 57], raw[:, 58] & 7
                cu_key = xcc * 4096 + ((hw >> 13) & 7) * 64 + ((hw >> 12) & 1) * 16 + ((hw >> 8) & 15)   # (XCC, SE_ID, SH_ID, CU_ID) of HW_REG_HW_ID
                T = ntiles // a.n
                img = tick // T
                cus = {}
                for k, im in zip(cu_key.tolist(), img.tolist()):
                    cus.setdefault(k, []).append(im)
                sizes = sorted(len(v) for v in cus.values())
                mixed = sum(1 for v in cus.values() if len(set(v)) > 1)
                print('    placement: %d workgroups on %d distinct (XCC, SE, SH, CU) ids, workgroups per id min %d max %d; ids that host tiles of two different images: %d' % (
                    len(tick), len(cus), sizes[0], sizes[-1], mixed))
                names = ['entry', 'chunk0', 'loop', '(poll)', 'epilogue', 'publish']
                for ty, tn in enumerate(('conv1 (64 ch in, waits up front)', 'conv2-4 (Cout 32)', 'conv5-class (Cout 64)')):
                    items = t[:, ty, 6]
                    if items.sum() == 0:
                        continue
                    per = t[:, ty, :6] / np.maximum(items[:, None], 1)
                    chunks = t[:, ty, 7].sum() / items.sum()
                    tot = per[:, [0, 1, 2, 4, 5]].sum(1)
                    print('    %-34s %5.0f items/WG, %.1f chunks/item | cycles per item (mean over workgroups; p10..p90 of the total): total %6.0f (%6.0f..%6.0f)' % (
                        tn, items.mean(), chunks, tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
                    print('        ' + '  '.join('%s %6.0f' % (nm, per[:, i].mean()) for i, nm in enumerate(names)) +
                          '  | loop per chunk %5.0f (MFMA-bound at 2 waves/SIMD: %d)' % ((per[:, 2].mean() - per[:, 3].mean()) / chunks, 2304 if ty < 2 else 4608))
            sys.stdout.flush()
    _lib.check(L.dasr_set_tuning(7, 1))


if __name__ == '__main__':
    main()
