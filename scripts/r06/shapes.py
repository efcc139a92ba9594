"""Round 6: SR step time (nf 64, nb 23) at several batch shapes under the three trunk schedules: per-layer launches (DASR_CHAIN=0), the layer-by-layer chained launches
(DASR_CHAIN_FORM=layer, eligible at 512 k tiles only) and the input-stationary chained launch (DASR_CHAIN_FORM=is: N * tiles a multiple of 256, tiles per image divides 32).
python scripts/r06/shapes.py [--shapes 8x128x128,16x128x128,...]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='8x128x128,16x128x128,24x128x128,32x128x128,16x64x128,32x64x64')
    ap.add_argument('--steps', type=int, default=6)
    a = ap.parse_args()
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    os.environ['DASR_STREAMS'] = '1'
    import torch
    import bench
    from dasr_amd import options
    from dasr_amd.models import create_model
    for shp in a.shapes.split(','):
        n, h, w = [int(x) for x in shp.split('x')]
        g = torch.Generator().manual_seed(1234)
        data = {'LR': torch.rand(n, 3, h, w, generator=g).cuda(), 'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g).cuda()}
        row = []
        for name, chain, form in (('per-layer', '0', 'layer'), ('layer chain', '1', 'layer'), ('is chain', '1', 'is')):
            os.environ['DASR_CHAIN'], os.environ['DASR_CHAIN_FORM'] = chain, form
            torch.manual_seed(0)
            m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
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
            for _ in range(a.steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.steps * 1e3
            used = m._out_plans[0].chain.form if m._out_plans[0].chain is not None else '-'
            row.append('%s %.2f ms (%s, err %d)' % (name, ms, used, int(m.netG.chain_err.item())))
            del m
            torch.cuda.empty_cache()
        print('%-12s %s' % (shp, ' | '.join(row)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
