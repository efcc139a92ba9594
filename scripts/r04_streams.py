"""Same-box A/B of the sub-batch schedule of the configs[1] step: replica streams {1, 2, 3, 4} x enqueue {chunk = one host
thread interleaving chunks, mt = one enqueue thread per replica inside the library (dasr_run_ops_mt), graph = hipGraph replay of the gradient
part} x GPU_MAX_HW_QUEUES {runtime default, 8}.  Every cell is its own process (the knobs are process-level); cells are visited round-robin
`--rounds` times so that box drift does not favour a column.

    python scripts/r04_streams.py [--rounds 2] [--steps 8] > gpurun_out/r04_streams.txt
    python scripts/r04_streams.py --cell 4 mt 8      # one cell (used by the driver loop above)

Columns: full step ms (feed + fwd + loss + bwd + wgrad phase + Adam + repack; the number bench.py reports), host ms to enqueue one step,
gradient-part ms eager / under hipGraph replay (Adam + repack excluded: their arguments change every step), kernel_time_over_wall of one
profiled step.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cell(streams, enq, steps, graph):
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

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    for _ in range(3):
        step()
    out = {'streams': streams, 'enq': enq, 'hwq': os.environ.get('GPU_MAX_HW_QUEUES', 'default'), 'plans': len(m._out_plans),
           'sub_batches': [p.N for p in m._out_plans]}
    out['step_ms'] = round(timed(step, steps), 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    out['host_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
    torch.cuda.synchronize()
    recs, wall, _ = bench.profiled_steps(step, 1)
    out['kernel_time_over_wall'] = round(sum(r[1] for r in recs) / (wall * 1e6), 3)
    out['launches'] = len(recs)
    if graph:
        adam, repack = m.optimizer_G.step, m.netG.repack
        m.optimizer_G.step = lambda lr: None
        m.netG.repack = lambda: None
        for _ in range(2):
            step()
        out['grad_eager_ms'] = round(timed(step, steps), 3)
        try:
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s):
                    step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            for _ in range(2):
                gr.replay()
            out['grad_graph_ms'] = round(timed(gr.replay, steps), 3)
            t0 = time.perf_counter()
            gr.replay()
            out['graph_host_ms'] = round((time.perf_counter() - t0) * 1e3, 3)
            torch.cuda.synchronize()
        except Exception as e:   # a capture failure must not lose the eager numbers of the cell
            out['graph_error'] = repr(e)[:200]
        m.optimizer_G.step, m.netG.repack = adam, repack
    print('CELL ' + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--cell', nargs=3, default=None, metavar=('STREAMS', 'ENQ', 'HWQ'))
    ap.add_argument('--cells', type=str, default='1:chunk:d,2:chunk:d,2:mt:d,2:mt:8,3:mt:d,3:mt:8,4:chunk:d,4:chunk:8,4:mt:d,4:mt:8')
    ap.add_argument('--no-graph', action='store_true')
    a = ap.parse_args()
    if a.cell:
        cell(int(a.cell[0]), a.cell[1], a.steps, not a.no_graph)
        return
    cells = [c.split(':') for c in a.cells.split(',')]
    rows = {}
    for rnd in range(a.rounds):
        for s, e, q in cells:
            env = dict(os.environ, DASR_STREAMS=s, DASR_ENQ=e)
            env.pop('GPU_MAX_HW_QUEUES', None)
            if q != 'd':
                env['GPU_MAX_HW_QUEUES'] = q
            cmd = [sys.executable, os.path.abspath(__file__), '--cell', s, e, q, '--steps', str(a.steps)] + (['--no-graph'] if (a.no_graph or rnd > 0) else [])
            try:
                p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
                line = [l for l in p.stdout.splitlines() if l.startswith('CELL ')]
                r = json.loads(line[-1][5:]) if line else {'error': (p.stderr or '')[-300:]}
            except subprocess.TimeoutExpired:
                r = {'error': 'timeout'}
            rows.setdefault((s, e, q), []).append(r)
            print('round %d streams %s enq %s hwq %s: %s' % (rnd, s, e, q, json.dumps(r)))
            sys.stdout.flush()
    print('\n%-8s %-6s %-4s %-12s %-22s %-8s %-8s %-12s %-12s %-10s' % ('streams', 'enq', 'hwq', 'sub-batches', 'step ms (per round)', 'host ms', 'k/wall', 'grad eager', 'grad graph', 'graph host'))
    for (s, e, q), rs in rows.items():
        ok = [r for r in rs if 'step_ms' in r]
        if not ok:
            print('%-8s %-6s %-4s failed: %s' % (s, e, q, rs[0].get('error', '')[:120]))
            continue
        r0 = ok[0]
        print('%-8s %-6s %-4s %-12s %-22s %-8s %-8s %-12s %-12s %-10s' % (s, e, q, '+'.join(str(x) for x in r0['sub_batches']), ' '.join('%.2f' % r['step_ms'] for r in ok),
                                                                    '%.2f' % r0['host_ms'], '%.2f' % r0['kernel_time_over_wall'], r0.get('grad_eager_ms', '-'),
                                                                    r0.get('grad_graph_ms', r0.get('graph_error', '-')), r0.get('graph_host_ms', '-')))


if __name__ == '__main__':
    main()
