"""Soak test of the chained trunk launches at configs[1]: (1) 300 training steps on changing batches with DASR_CHAIN=1 and with per-layer launches
(one plan each): the weights must stay BIT-identical -- a single stale halo read in any of 300 x 688 chained layers would show; (2) 2000 more chained steps with
the device error word checked every 250 steps and the step time of every block printed (a broken wait would also show as a ~1 s stall).
    python scripts/r04/chain_soak.py [--steps 300] [--more 2000]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--more', type=int, default=2000)
    ap.add_argument('--batch', type=int, default=16, help='32: two chained launches over 16-image sub-batches per direction (RRDBNetHIP.chain_split, round 5)')
    ap.add_argument('--lr', type=int, default=128, help='LR crop size (round 6: 32 = the shipped shape, input-stationary launches of 4-row tiles; 64: 8-row tiles at batch 16)')
    a = ap.parse_args()
    os.environ['DASR_STREAMS'] = '1'
    import torch
    import bench
    from dasr_amd import options
    from dasr_amd.models import create_model
    g = torch.Generator().manual_seed(99)
    pool = [{'LR': torch.rand(a.batch, 3, a.lr, a.lr, generator=g).cuda(), 'HR': torch.rand(a.batch, 3, 4 * a.lr, 4 * a.lr, generator=g).cuda()} for _ in range(4)]
    finals = []
    for chain in ('1', '0'):
        os.environ['DASR_CHAIN'] = chain
        torch.manual_seed(0)
        m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
        t0 = time.perf_counter()
        for step in range(1, a.steps + 1):
            m.update_learning_rate()
            m.feed_data(pool[step % 4])
            m.optimize_parameters(step)
        loss = m.get_current_log()['l_pix']   # (host sync + error-word / non-finite checks)
        dt = (time.perf_counter() - t0) / a.steps * 1e3
        finals.append(m.netG.params.flat.clone())
        print('chain %s: %d steps, %.2f ms / step, last l_pix %.6f, chained plan %s' % (chain, a.steps, dt, loss, m._out_plans[0].chain.form if m._out_plans[0].chain is not None else None))
        sys.stdout.flush()
        if chain == '1':
            keep = m
        else:
            del m
    same = bool(torch.equal(finals[0], finals[1]))
    print('weights after %d steps bit-identical: %s (max |d| %.3e)' % (a.steps, same, float((finals[0] - finals[1]).abs().max())))
    m = keep
    step = a.steps
    for blk in range(a.more // 250):
        t0 = time.perf_counter()
        for _ in range(250):
            step += 1
            m.update_learning_rate()
            m.feed_data(pool[step % 4])
            m.optimize_parameters(step)
        loss = m.get_current_log()['l_pix']
        print('chained steps %d-%d: %.2f ms / step, l_pix %.6f, error word %d' % (step - 249, step, (time.perf_counter() - t0) / 250 * 1e3, loss, int(m._out_plans[0].chain.err.item())))
        sys.stdout.flush()
    sys.exit(0 if same else 1)


if __name__ == '__main__':
    main()
