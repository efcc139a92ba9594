"""Step time + per-kernel launch durations of the configs[1] step in THIS process's library / tuning (DASR_HIP_LIB, DASR_TUNE): the cell of the
same-box A/B loops in scripts/r04/alias_ab.sh.   python scripts/r04/step_time.py [--steps 8] [--label text]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--label', default='')
    a = ap.parse_args()
    os.environ.setdefault('DASR_ALLOW_NONFINITE', '1')
    import torch
    import bench
    from dasr_amd import options
    from dasr_amd.models import create_model
    torch.manual_seed(0)
    m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
    g = torch.Generator().manual_seed(1234)
    data = {'LR': torch.rand(16, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(16, 3, 512, 512, generator=g).cuda()}
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
    recs, wall, _ = bench.profiled_steps(step, 1)
    by = {}
    for r in recs:
        k = by.setdefault(r[0], [0, 0.0])
        k[0] += 1
        k[1] += r[1]
    top = sorted(by.items(), key=lambda kv: -kv[1][1])[:6]
    print('%-44s step %.2f ms | k/wall %.2f | %s' % (a.label or os.environ.get('DASR_TUNE', 'default'), ms, sum(r[1] for r in recs) / (wall * 1e6),
                                                      '  '.join('%s x%d %.1fus' % (str(k)[:28], v[0], v[1] / v[0]) for k, v in top)))


if __name__ == '__main__':
    main()
