"""Round 6: tile heights of the input-stationary chained launch (16 / 8 / 4 rows of 32 pixels = 8 x 2, 8 x 1, 4 x 1 waves x rows; 2-row tiles were built and dropped) at the small shapes: per shape,
the step of one launch per conv (the bit-identity reference: SR output, gradients, weights after two steps), then every feasible height forced through
dasr_set_tuning(10, th) and the launcher's own choice (0): bit identity + SR step time (nf 64, nb 23) + the two chained launches' durations.
python scripts/r06/is_tiles.py [--shapes 16x32x32,...] [--steps 6]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='16x32x32,32x32x32,8x32x32,16x64x64,8x64x64,16x48x48')
    ap.add_argument('--steps', type=int, default=6)
    a = ap.parse_args()
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    os.environ['DASR_STREAMS'] = '1'
    import torch
    import bench
    from dasr_amd import options, _lib
    from dasr_amd.models import create_model
    from dasr_amd.rrdbnet import RRDBNetHIP
    from dasr_amd.engine import ceil_div
    L = _lib.lib()
    for shp in a.shapes.split(','):
        n, h, w = [int(x) for x in shp.split('x')]
        g = torch.Generator().manual_seed(1234)
        data = {'LR': torch.rand(n, 3, h, w, generator=g).cuda(), 'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g).cuda()}
        ref = None
        cells = [('per-layer', '0', 0)] + [('th %d' % th, '1', th) for th in (16, 8, 4) if RRDBNetHIP.is_geometry(n, ceil_div(h, th) * ceil_div(w, 32)) is not None] + [('auto', '1', 0)]
        for name, chain, th in cells:
            os.environ['DASR_CHAIN'], os.environ['DASR_CHAIN_FORM'] = chain, 'is'
            assert L.dasr_set_tuning(10, th) == 0
            torch.manual_seed(0)
            m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
            st = [0]

            def step():
                st[0] += 1
                m.update_learning_rate()
                m.feed_data(data)
                m.optimize_parameters(st[0])
            step(), step()
            torch.cuda.synchronize()
            out = (m.fake_H.clone(), m.netG.params.grad.clone(), m.netG.params.flat.clone())
            if ref is None:
                ref = out
            same = all(torch.equal(x, y) for x, y in zip(out, ref))
            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            recs, wall, _ = bench.profiled_steps(step, 1)
            ch = ['%.0f us' % r[1] for r in recs if 'rdb_is' in str(r[0])]
            gl = RRDBNetHIP.is_geometry(n, ceil_div(h, th) * ceil_div(w, 32)) if th else RRDBNetHIP.is_launch(n, h, w)
            print('%-10s %-10s step %6.2f ms | chains %-20s | %s | err %d | %s' % (shp, name, ms, ' + '.join(ch) or '-', 'bit-identical' if same else 'DIFFERENT',
                                                                             int(m.netG.chain_err.item()), ('workgroups %d x tpw %d' % (8 * gl[-2], gl[-1])) if (chain == '1' and gl) else ''))
            sys.stdout.flush()
            del m
            torch.cuda.empty_cache()
    L.dasr_set_tuning(10, 0)


if __name__ == '__main__':
    main()
