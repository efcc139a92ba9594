"""Register-window compute form of wgrad3_ld_kernel (round 6) against the round-3 form (wgrad3_ld6_kernel, six fragment reads per k-step), same process,
same box: libdasr_hip_ablate.so holds both (dasr_wgrad_set_mode bit 9).  Prints the configs[1] step time and the launch durations of the weight-gradient
kernels under each form, alternating, and checks that the two forms give BIT-IDENTICAL gradients (same MFMAs on the same fragments in the same order).
DASR_HIP_LIB=dasr_amd/libdasr_hip_ablate.so python scripts/r06/wgrad_ab.py [--steps 8] [--dsn]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--n', type=int, default=16)
    ap.add_argument('--lr', type=int, default=128)
    ap.add_argument('--regload', action='store_true', help='the loaders stage through registers (ds_write_b128) instead of LDS-DMA: correct results, A/B')
    ap.add_argument('--abl', action='store_true', help='ablations of the register-window kernel (WRONG results): no DMA after two tiles / no fragment reads / both')
    a = ap.parse_args()
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    import torch
    import bench
    from dasr_amd import options, _lib
    from dasr_amd.models import create_model
    torch.manual_seed(0)
    m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(a.n, 3, a.lr, a.lr, generator=g).cuda(), 'HR': torch.rand(a.n, 3, 4 * a.lr, 4 * a.lr, generator=g).cuda()}
    st = [0]

    def step():
        st[0] += 1
        m.update_learning_rate()
        m.feed_data(data)
        m.optimize_parameters(st[0])

    L = _lib.lib()
    grads = {}
    for rnd in range(2):
        cells = ((1, 'register window (wgrad3_ld_kernel)'), (1 | 512, 'six reads per k-step (wgrad3_ld6_kernel)'))
        if a.abl:
            cells = ((1, 'register window (wgrad3_ld_kernel)'), (1 | 1024, '  no LDS-DMA after the first two tiles'), (1 | 2048, '  no fragment reads inside the k-steps'),
                     (1 | 3072, '  neither (MFMA + barriers)'))
        if a.regload:
            cells = ((1, 'register window, LDS-DMA loaders (product)'), (1 | 4096, 'register window, register-staged loaders'))
        for mode, name in cells:
            assert L.dasr_wgrad_set_mode(mode) == 0, 'needs libdasr_hip_ablate.so (DASR_HIP_LIB)'
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            recs, wall, _ = bench.profiled_steps(step, 1)
            by = {}
            for r in recs:
                k = by.setdefault(r[0], [0, 0.0])
                k[0] += 1
                k[1] += r[1]
            wg = ['%s x%d %.1f us' % (str(k)[:40], v[0], v[1] / v[0]) for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]) if 'wgrad3' in str(k)]
            print('%-44s step %.2f ms | %s' % (name, ms, '  '.join(wg)), flush=True)
            if rnd == 0:   # gradients of ONE step from the same weights: reload the initial state first
                pass
    if a.abl:
        L.dasr_wgrad_set_mode(1)
        return
    # bit identity: same weights, same batch, one step under each form
    sd = {k: v.clone() for k, v in m.netG.state_dict().items()}
    other = (1 | 4096) if a.regload else (1 | 512)
    for mode in (1, other):
        L.dasr_wgrad_set_mode(mode)
        m.netG.load_state_dict(sd)
        m.feed_data(data)
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        grads[mode] = m.netG.params.grad.clone()
    d = (grads[1] - grads[other]).abs().max().item()
    print('gradients of one step, product vs the other form: max abs diff %.3e (%s), |grad| max %.3e' % (
        d, 'BIT-IDENTICAL' if torch.equal(grads[1], grads[other]) else 'different', grads[1].abs().max().item()))
    L.dasr_wgrad_set_mode(1)


if __name__ == '__main__':
    main()
