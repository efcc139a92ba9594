"""How much of the configs[1] step time is the data-dependent shader clock?  (profiles/r04_zero_data.txt)

The MFMA-only probe runs 2.3-2.5 PFLOP/s on all-zero operands and 1.7-1.8 PFLOP/s on random bf16 operands (same instruction stream,
lower clock under load).  This script times the PRODUCTION step of configs[1] (same plans, same launches, same bytes) three times in one process:
random data / weights as bench.py has them, then with every weight and every input image zero (all activations, gradients and partial sums are
zero: the instruction stream and the memory traffic are unchanged, only the operand toggling is gone), then random again.

    python scripts/r04/zero_data.py [--steps 8]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    a = ap.parse_args()
    import torch
    import bench
    from dasr_amd import options
    from dasr_amd.models import create_model
    torch.manual_seed(0)
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
    g = torch.Generator().manual_seed(1234)
    rnd = {'LR': torch.rand(16, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(16, 3, 512, 512, generator=g).cuda()}
    zero = {k: torch.zeros_like(v) for k, v in rnd.items()}
    st = [0]

    def run(data, n):
        for _ in range(n):
            st[0] += 1
            m.update_learning_rate()
            m.feed_data(data)
            m.optimize_parameters(st[0])

    def timed(data):
        run(data, 3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(data, a.steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    P = m.netG.params
    saved = (P.flat.clone(), P.m.clone(), P.v.clone())
    print('random data, random weights        : %.2f ms / step' % timed(rnd))
    P.flat.zero_(); P.m.zero_(); P.v.zero_()
    m.netG.repack()
    print('zero data, zero weights            : %.2f ms / step   (same launches and bytes, no operand toggling)' % timed(zero))
    print('   max |weight| after the zero steps: %.1e' % float(P.flat.abs().max()))
    print('random data, zero weights          : %.2f ms / step' % timed(rnd))
    P.flat.copy_(saved[0]); P.m.copy_(saved[1]); P.v.copy_(saved[2])
    m.netG.repack()
    print('zero data, random weights          : %.2f ms / step' % timed(zero))
    print('random data, random weights (again): %.2f ms / step' % timed(rnd))


if __name__ == '__main__':
    main()
