"""DASR_CHAIN=1 (the trunk's forward dense-block convs as one persistent chained launch) against the per-layer launches: bit-identical SR output and
gradients at small size, then step time at configs[1].   python scripts/r04/chain_check.py [--bench]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(chain, nf, nb):
    os.environ['DASR_CHAIN'] = '1' if chain else '0'
    import torch
    import bench
    from dasr_amd import options
    from dasr_amd.models import create_model
    torch.manual_seed(0)
    return create_model(options.dict_to_nonedict(bench.make_opt(nf, nb)))


def main():
    import torch
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    if '--bench' not in sys.argv:
        for (n, h, w, nb) in ((16, 128, 128, 1), (8, 256, 128, 2), (16, 128, 128, 3)):
            g = torch.Generator().manual_seed(5)
            data = {'LR': torch.rand(n, 3, h, w, generator=g).cuda(), 'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g).cuda()}
            outs = []
            for chain in (False, True):
                m = build(chain, 64, nb)
                for step in (1, 2):
                    m.update_learning_rate()
                    m.feed_data(data)
                    m.optimize_parameters(step)
                torch.cuda.synchronize()
                plans = m._out_plans
                used = [p.chain is not None for p in plans]
                if chain:
                    for p in plans:
                        if p.chain is not None:
                            p.chain.check()
                outs.append((m.fake_H.clone(), m.netG.params.grad.clone(), m.netG.params.flat.clone(), used))
            (s0, g0, w0, u0), (s1, g1, w1, u1) = outs
            print('N %d %dx%d nb %d: chain used %s | SR bit-identical %s (max |d| %.3e) | gradients bit-identical %s | weights after 2 steps bit-identical %s' %
                  (n, h, w, nb, u1, bool(torch.equal(s0, s1)), float((s0 - s1).abs().max()), bool(torch.equal(g0, g1)), bool(torch.equal(w0, w1))))
            sys.stdout.flush()
        return
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(16, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(16, 3, 512, 512, generator=g).cuda()}
    for rnd in (1, 2):
        for streams, chain in (('2', False), ('1', False), ('1', True)):
            if True:
                os.environ['DASR_STREAMS'] = streams
                m = build(chain, 64, 23)
                st = [0]

                def step():
                    st[0] += 1
                    m.update_learning_rate()
                    m.feed_data(data)
                    m.optimize_parameters(st[0])
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    step()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 8 * 1e3
                for p in m._out_plans:
                    if p.chain is not None:
                        p.chain.check()
                print('round %d streams %s chain %d: %.2f ms / step (chain plans: %s)' % (rnd, streams, chain, ms, [p.chain is not None for p in m._out_plans]))
                sys.stdout.flush()
                del m
                torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
