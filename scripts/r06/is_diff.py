"""Round 6 debugging aid: where do the input-stationary chained launches first differ from the per-layer launches?  Compares every dense slab (16-bit planes) of the forward pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
os.environ['DASR_STREAMS'] = '1'


def build(chain, form, nb):
    os.environ['DASR_CHAIN'] = '1' if chain else '0'
    os.environ['DASR_CHAIN_FORM'] = form
    import bench
    from dasr_amd import options
    from dasr_amd.models import create_model
    torch.manual_seed(0)
    return create_model(options.dict_to_nonedict(bench.make_opt(64, nb)))


n, h, w, nb = 16, 128, 128, 2
g = torch.Generator().manual_seed(5)
data = {'LR': torch.rand(n, 3, h, w, generator=g).cuda(), 'HR': torch.rand(n, 3, 4 * h, 4 * w, generator=g).cuda()}
res = []
for chain, form in ((False, 'layer'), (True, 'is')):
    m = build(chain, form, nb)
    m.update_learning_rate(); m.feed_data(data); m.optimize_parameters(1)
    torch.cuda.synchronize()
    p = m._out_plans[0]
    res.append(([s.t.clone() if hasattr(s, 't') else None for s in p.slabs], [g_.t.clone() if hasattr(g_, 't') else None for g_ in p.gslab], m.fake_H.clone()))
    print(type(p.slabs[0]), [a for a in dir(p.slabs[0]) if not a.startswith('_')][:12])
(a, ga, sa), (b, gb, sb) = res
for name, A, B in (('slab', a, b), ('gslab', ga, gb)):
    for i, (x, y) in enumerate(zip(A, B)):
        if x is None:
            continue
        xf, yf = x.float(), y.float()
        for pl in range(x.shape[1]):
            d = (xf[:, pl] - yf[:, pl]).abs()
            if float(d.max()) != 0.0:
                idx = torch.nonzero(d > 0)
                print('%s %d plane %d: %d elements differ, max |d| %.3e; first at %s (a %.6e b %.6e)' % (name, i, pl, idx.shape[0], float(d.max()), idx[0].tolist(),
                      float(xf[:, pl][tuple(idx[0].tolist())]), float(yf[:, pl][tuple(idx[0].tolist())])))
                break
        else:
            continue
        break
print('SR max |d|', float((sa - sb).abs().max()))
