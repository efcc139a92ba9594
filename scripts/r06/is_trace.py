"""Where an input-stationary chained launch (rdb_is_kernel, DASR_CHAIN_FORM=is) spends its cycles: per-workgroup accumulators of the -DDASR_TRACE build.

  python -m dasr_amd.build --trace && DASR_HIP_LIB=dasr_amd/libdasr_hip_trace.so python scripts/r06/is_trace.py [--n 16] [--nb 23]

Accumulators (cycles of thread 0 = wave 0, summed over the launch, per workgroup): 0 steps with one Cout-32 conv (18 MFMAs per wave), 1 steps with two m-tiles (36 MFMAs per wave),
2 end-of-step wait + barrier, 3 blocking waits (a group missing where it is needed), 4 item start (x landed + barrier), 5 epilogues conv1-4 (+ flush of an older flag),
6 epilogue conv5 (+ next item's x request), 7 residual scaling (waits for the fp32 residual), 8 / 9 step counts, 10 items, 11 blocking waits (count)."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CH_BASE = 1 << 20


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=16)
    ap.add_argument('--nb', type=int, default=23)
    ap.add_argument('--lr', type=int, default=128)
    ap.add_argument('--stagger', type=str, default='0')
    a = ap.parse_args()
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    os.environ['DASR_STREAMS'] = '1'
    os.environ['DASR_CHAIN_FORM'] = 'is'
    import numpy as np
    import torch
    import bench
    from dasr_amd import _lib, options
    from dasr_amd.engine import OpList
    from dasr_amd.models import create_model
    L = _lib.lib()
    traced = hasattr(L, 'dasr_debug_set_trace')
    torch.manual_seed(0)
    m = create_model(options.dict_to_nonedict(bench.make_opt(64, a.nb)))
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(a.n, 3, a.lr, a.lr, generator=g).cuda(), 'HR': torch.rand(a.n, 3, 4 * a.lr, 4 * a.lr, generator=g).cuda()}
    for st in (1, 2):
        m.update_learning_rate()
        m.feed_data(data)
        m.optimize_parameters(st)
    torch.cuda.synchronize()
    plan = m._out_plans[0]
    assert plan.chain is not None and plan.chain.form == 'is'
    for stg, (name, ch) in [(int(x), c) for x in a.stagger.split(',') for c in (('forward chain', plan.chain), ('data-gradient chain', plan.chain_b))]:
        _lib.check(L.dasr_set_tuning(9, stg))
        if stg:
            name += ' stagger %d' % stg
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
        nblk = ch.n // 5
        print('%-20s %d blocks: %.1f us per launch = %.2f us per dense block, %.0f TFLOP/s' % (name, nblk, us, us / nblk, ch.flops / us / 1e6))
        if traced:
            buf = torch.zeros(CH_BASE + 512 * 64 + 64, dtype=torch.int64, device='cuda')
            L.dasr_debug_set_trace.argtypes = [ctypes.c_void_p]
            _lib.check(L.dasr_debug_set_trace(buf.data_ptr()))
            one.run()
            torch.cuda.synchronize()
            _lib.check(L.dasr_debug_set_trace(None))
            t = buf[CH_BASE:CH_BASE + 256 * 64].view(256, 64).cpu().numpy().astype(np.float64)
            items = t[:, 10].mean()
            names = ['steps 1 m-tile', 'steps 2 m-tiles', 'end-of-step wait', 'blocking waits', 'item start', 'epilogue conv1-4', 'epilogue conv5', 'residual scale']
            tot = t[:, :8].sum(axis=1)
            print('    per item (one dense block of one tile), mean over 256 workgroups; items per workgroup %.0f; accounted total %.0f cycles per item (p10 %.0f p90 %.0f)' %
                  (items, tot.mean() / items, np.percentile(tot, 10) / items, np.percentile(tot, 90) / items))
            for k, nm in enumerate(names):
                extra = ''
                if k == 0:
                    extra = ' = %.0f per step (MFMA-bound: 1152)' % (t[:, 0].sum() / max(1.0, t[:, 8].sum()))
                if k == 1:
                    extra = ' = %.0f per step (MFMA-bound: 2304)' % (t[:, 1].sum() / max(1.0, t[:, 9].sum()))
                if k == 2:
                    extra = ' = %.0f per step | vmcnt wait %.0f / %.0f, barrier %.0f / %.0f per step (1 / 2 m-tiles)' % (
                        t[:, 2].sum() / max(1.0, t[:, 8].sum() + t[:, 9].sum()), t[:, 12].sum() / max(1.0, t[:, 8].sum()), t[:, 13].sum() / max(1.0, t[:, 9].sum()),
                        t[:, 14].sum() / max(1.0, t[:, 8].sum()), t[:, 15].sum() / max(1.0, t[:, 9].sum()))
                if k == 3:
                    extra = ' (%.2f blocking waits per item)' % (t[:, 11].sum() / max(1.0, t[:, 10].sum()))
                print('      %-18s %8.0f%s' % (nm, t[:, k].mean() / items, extra))
        sys.stdout.flush()
    plan.check_chain()


if __name__ == '__main__':
    main()
