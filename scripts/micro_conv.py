"""Isolated dense-block conv launches for profiling: python scripts/micro_conv.py [--cin 160] [--cout 32] [--n 16] [--hw 128] [--reps 40] [--tune k=v,...]"""
import argparse, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dasr_amd import _lib, engine
from dasr_amd.engine import BTensor, ParamStore, PackRegistry, OpList, conv_op

ap = argparse.ArgumentParser()
ap.add_argument('--cin', type=int, default=160); ap.add_argument('--cout', type=int, default=32)
ap.add_argument('--n', type=int, default=16); ap.add_argument('--hw', type=int, default=128)
ap.add_argument('--reps', type=int, default=40); ap.add_argument('--tune', type=str, default='')
ap.add_argument('--streams', type=int, default=1)
ap.add_argument('--noout', type=int, default=0)
ap.add_argument('--zero', type=int, default=0, help='1: zero-filled activations and weights (DVFS: how much of the time is clock, not cycles)')
ap.add_argument('--mode', type=str, default='fwd', help='fwd: bias+lrelu->bf16 | dgrad: mask->bf16 | conv5: bias, alpha, res1 -> f32+bf16')
ap.add_argument('--alias5', type=int, default=0, help='conv5 mode: bit 0 residual aliased, bit 1 fp32 output aliased, bit 2 in-place')
ap.add_argument('--alias', type=int, default=0, help='1: all images alias image 0 on the input, 2: also on the output (cache-resident working set)')
a = ap.parse_args()
engine.ensure_runtime_ready()
dev = torch.device('cuda')
L = _lib.lib()
for kv in [x for x in a.tune.split(',') if x]:
    k, v = kv.split('=')
    _lib.check(L.dasr_set_tuning(int(k), int(v)))
mt = 2 if a.cout == 64 else 1
P = ParamStore([('w', (a.cout, a.cin, 3, 3)), ('b', (a.cout,))], dev)
if not a.zero:
    P.flat.normal_(0, math.sqrt(2.0 / (9 * a.cin)))
pack = PackRegistry(P)
ref = pack.add(a.cout, a.cin, 9, mt, 1, [(0, a.cout, a.cin, 0, a.cin, 0, 0)])
pack.finalize(); pack.run()
lists = []
for s in range(a.streams):
    n = a.n // a.streams
    x = BTensor(n, a.cin, a.hw, a.hw, False, dev)
    if not a.zero:
        x.t.normal_()
    y = BTensor(n, a.cout, a.hw, a.hw, False, dev); y.t.normal_()
    ym = BTensor(n, a.cout, a.hw, a.hw, False, dev)
    rf = BTensor(n, a.cout, a.hw, a.hw, True, dev); of = BTensor(n, a.cout, a.hw, a.hw, True, dev)
    ol = OpList()
    for _ in range(a.reps):
        xv, yv = x.view(), y.view()
        if a.alias in (1, 2):
            xv.n_stride = 0
        if a.alias >= 2:   # 3: only the mask / output tensor aliases image 0
            yv.n_stride = 0
        if a.mode == 'dgrad':
            ol.add(conv_op(pack, ref, xv, False, a.cin, a.hw, a.hw, a.hw, a.hw, n, mask=yv, mask_f32=0, out_bf16=ym.view()))
        elif a.mode == 'conv5':
            rv, ov = rf.view(), of.view()
            if a.alias5 & 1:   # fp32 residual: every image reads image 0's
                rv.n_stride = 0
            if a.alias5 & 2:   # fp32 output: every image writes image 0's
                ov.n_stride = 0
            if a.alias5 & 4:   # fp32 output written over the residual (in place, as a read-modify-write stream would)
                ov = rf.view()
            ol.add(conv_op(pack, ref, xv, False, a.cin, a.hw, a.hw, a.hw, a.hw, n, bias=P.ptr('b'), alpha=0.2, res1=rv, beta1=1.0, out_f32=ov, out_bf16=yv))
        else:
            ol.add(conv_op(pack, ref, xv, False, a.cin, a.hw, a.hw, a.hw, a.hw, n, bias=P.ptr('b'), act=1, out_bf16=None if a.noout else yv, out_f32=None))
    ol.keep += [x, y, ym, rf, of]
    lists.append(ol)
streams = [torch.cuda.Stream() for _ in lists]
def run():
    if len(lists) == 1:
        lists[0].run()
    else:
        engine.run_interleaved(lists, streams, chunk=8)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
run(); torch.cuda.synchronize()
dt = time.perf_counter() - t0
fl = 2.0 * a.n * a.hw * a.hw * 9 * a.cin * a.cout * a.reps
print('%s alias %d/%d zero %d ' % (a.mode, a.alias, a.alias5, a.zero) + 'cin %d cout %d N %d streams %d tune [%s]: %.1f us/launch-set, %.0f TFLOP/s' % (a.cin, a.cout, a.n, a.streams, a.tune, dt / a.reps * 1e6, fl / dt / 1e12))

if os.environ.get('DASR_HIP_LIB'):
    import ctypes, numpy as np
    n = a.n // a.streams
    grid = (1 if a.cout <= 32 else (a.cout + 63) // 64 if mt == 2 else (a.cout + 31) // 32) * n * ((a.hw + 15) // 16) * ((a.hw + 31) // 32)
    buf = torch.zeros(grid * 16 + 64, dtype=torch.int64, device=dev)
    L.dasr_debug_set_trace.argtypes = [ctypes.c_void_p]
    one = OpList(); one.add(lists[0].ops[0])
    one.run(); torch.cuda.synchronize()
    _lib.check(L.dasr_debug_set_trace(buf.data_ptr()))
    one.run(); torch.cuda.synchronize()
    _lib.check(L.dasr_debug_set_trace(None))
    t = buf[:grid * 16].view(grid, 16).cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    rt = t[:, 15] * 10.0  # s_memrealtime: 100 MHz -> ns
    print('trace: %d workgroups; entry spread (ns): p0 %.0f p50 %.0f p90 %.0f p100 %.0f' % (len(t), 0, np.percentile(rt - rt.min(), 50), np.percentile(rt - rt.min(), 90), (rt - rt.min()).max()))
    names = ['entry->loads issued', 'loads issued->chunk0 in LDS', 'chunk0 compute', 'rest of main loop', 'bias/epilogue setup', 'epilogue to stores issued', 'stores issued -> retired']
    for i, nm in enumerate(names):
        d = t[:, i + 1] - t[:, i]
        print('  %-30s cycles p10 %7.0f p50 %7.0f p90 %7.0f' % (nm, np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
    glds = t[:, 12].max() == 0
    for i, nm in zip(range(8, 13), ['chunk2: reads + MFMAs + DMA issue', 'chunk2: wait for own DMA (vmcnt 0)', 'chunk2: barrier', '-', '-'] if glds else
                     ['chunk2: store_chunk (vmcnt wait + ds_write)', 'chunk2: barrier 1', 'chunk2: issue loads', 'chunk2: compute', 'chunk2: barrier 2']):
        d = t[:, i + 1] - t[:, i]
        if t[:, i].min() > 0 and t[:, i + 1].min() > 0:
            print('  %-44s cycles p10 %7.0f p50 %7.0f p90 %7.0f' % (nm, np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
    wall = (t[:, 14] - t[:, 15]) * 10.0
    cyc = t[:, 7] - t[:, 0]
    print('  workgroup wall time (s_memrealtime) p50 %.0f ns -> shader clock %.2f GHz; last exit - first entry %.0f ns' % (np.percentile(wall, 50), np.median(cyc / wall), (t[:, 14].max() - t[:, 15].min()) * 10.0))
    d = t[:, 7] - t[:, 0]
    print('  %-30s cycles p10 %7.0f p50 %7.0f p90 %7.0f' % ('whole workgroup', np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90)))
