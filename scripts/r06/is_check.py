"""Round 6: the input-stationary chained launches (DASR_CHAIN_FORM=is, dasr_rdb_chain) against the per-layer launches: bit-identical SR output, gradients and
weights after two steps at small sizes, then (--bench) the configs[1] step time of per-layer / layer chain / input-stationary chain.   python scripts/r06/is_check.py [--bench]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build(chain, form, nf, nb):
    os.environ['DASR_CHAIN'] = '1' if chain else '0'
    os.environ['DASR_CHAIN_FORM'] = form
    os.environ['DASR_STREAMS'] = '1'
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
        shapes = ((16, 128, 128, 1), (8, 128, 128, 2), (32, 128, 128, 1), (16, 128, 128, 3), (16, 64, 128, 2), (64, 64, 64, 1), (8, 128, 112, 1), (16, 32, 32, 2), (16, 64, 64, 1))
        for (n, h, w, nb) in shapes:
            g = torch.Generator().manual_seed(5)
            data = {'LR': torch.rand(n, 3, h, w, generator=g).cuda(), 'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g).cuda()}
            outs = []
            for chain, form in ((False, 'layer'), (True, 'is')):
                m = build(chain, form, 64, nb)
                for step in (1, 2):
                    m.update_learning_rate()
                    m.feed_data(data)
                    m.optimize_parameters(step)
                torch.cuda.synchronize()
                plans = m._out_plans
                used = [(p.chain is not None and p.chain.form) for p in plans]
                err = int(m.netG.chain_err.item())
                outs.append((m.fake_H.clone(), m.netG.params.grad.clone(), m.netG.params.flat.clone(), used, err))
                del m
            (s0, g0, w0, u0, e0), (s1, g1, w1, u1, e1) = outs
            print('N %d %dx%d nb %d: chain %s err %d | SR bit-identical %s (max |d| %.3e) | gradients bit-identical %s (max |d| %.3e of %.3e) | weights after 2 steps bit-identical %s' %
                  (n, h, w, nb, u1, e1, bool(torch.equal(s0, s1)), float((s0 - s1).abs().max()), bool(torch.equal(g0, g1)), float((g0 - g1).abs().max()), float(g0.abs().max()),
                   bool(torch.equal(w0, w1))))
            sys.stdout.flush()
        return
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(16, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(16, 3, 512, 512, generator=g).cuda()}
    import bench
    for rnd in (1, 2):
        for chain, form in ((True, 'layer'), (True, 'is')):
            m = build(chain, form, 64, 23)
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
            recs, wall, _ = bench.profiled_steps(step, 1)
            by = {}
            for r in recs:
                k = by.setdefault(r[0], [0, 0.0])
                k[0] += 1
                k[1] += r[1]
            top = sorted(by.items(), key=lambda kv: -kv[1][1])[:4]
            print('round %d form %-5s: %.2f ms / step | err %d | %s' % (rnd, form, ms, int(m.netG.chain_err.item()),
                                                                     '  '.join('%s x%d %.1fus' % (str(k)[:32], v[0], v[1] / v[0]) for k, v in top)))
            sys.stdout.flush()
            del m
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
