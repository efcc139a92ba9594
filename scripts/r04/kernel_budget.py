"""Per-kernel time budget of the configs[1] step: for every kernel of the production step, launches, average launch duration on random and on all-zero
operands (same process, same plans), and the time an MFMA-only stream would need for the launch's algorithmic FLOPs at the rate the probe measures on
that operand kind -- the difference is the launch's non-MFMA time (prologue, LDS-DMA waits, epilogue, boundary), the quantity round 5 has to attack.
Run with DASR_STREAMS=1 for chip-exclusive launches (no second stream sharing the CUs) and with the default 2 for the production schedule.

    DASR_STREAMS=1 python scripts/r04/kernel_budget.py   > gpurun_out/r04_kernel_budget_1stream.txt
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    import torch
    import bench
    from dasr_amd import options, _lib
    from dasr_amd.models import create_model
    torch.manual_seed(0)
    m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
    g = torch.Generator().manual_seed(1234)
    rnd = {'LR': torch.rand(16, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(16, 3, 512, 512, generator=g).cuda()}
    zero = {k: torch.zeros_like(v) for k, v in rnd.items()}
    st = [0]
    cur = [rnd]

    def step():
        st[0] += 1
        m.update_learning_rate()
        m.feed_data(cur[0])
        m.optimize_parameters(st[0])

    def table():
        for _ in range(3):
            step()
        recs, wall, _ = bench.profiled_steps(step, 2)
        by = {}
        for tag, us, fl, bk in recs:
            r = by.setdefault(bench.kernel_name(tag), [0, 0.0, 0.0])
            r[0] += 1
            r[1] += us
            r[2] += fl
        return by, wall / 2 * 1e3

    BL = _lib.bench_lib()
    pk = C.c_float(0.0)
    rate = {}
    for mode, key in ((2, 'zero'), (0, 'random')):
        BL.dasr_probe_mfma_data(19968, mode, C.byref(pk), None)
        rate[key] = pk.value
    by_r, ms_r = table()
    P = m.netG.params
    P.flat.zero_(); P.m.zero_(); P.v.zero_()
    m.netG.repack()
    cur[0] = zero
    by_z, ms_z = table()
    print('streams %s | step %.2f ms random operands, %.2f ms zero operands | MFMA-only probe: %.0f TFLOP/s random bf16, %.0f zero' %
          (os.environ.get('DASR_STREAMS', '2'), ms_r, ms_z, rate['random'], rate['zero']))
    print('%-44s %8s %10s %10s %12s %12s %14s %14s' % ('kernel', 'launches', 'us random', 'us zero', 'MFMA-only r', 'MFMA-only z', 'non-MFMA ms r', 'non-MFMA ms z'))
    tot = [0.0, 0.0, 0.0, 0.0]
    for k, (n, us, fl) in sorted(by_r.items(), key=lambda kv: -kv[1][1]):
        n2, usz, _ = by_z.get(k, (n, 0.0, 0.0))
        a_r, a_z = us / n, usz / max(n2, 1)
        mr, mz = fl / n / (rate['random'] * 1e6), fl / n / (rate['zero'] * 1e6)   # us
        nm_r, nm_z = (a_r - mr) * n / 2 / 1e3, (a_z - mz) * n / 2 / 1e3           # ms per step (two profiled steps)
        tot[0] += us / 2 / 1e3; tot[1] += usz / 2 / 1e3; tot[2] += nm_r; tot[3] += nm_z
        if us / 2 > 50:
            print('%-44s %8d %10.1f %10.1f %12.1f %12.1f %14.2f %14.2f' % (k[:44], n // 2, a_r, a_z, mr, mz, nm_r, nm_z))
    print('%-44s %8s %10.2f %10.2f %12s %12s %14.2f %14.2f   (sums per step, ms; with 2 streams launch durations overlap)' % ('all kernels', '', tot[0], tot[1], '', '', tot[2], tot[3]))


if __name__ == '__main__':
    main()
