"""back-to-back launch cost of a near-empty kernel (dasr_fill_f32 of 256 floats) through the op-list executor"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dasr_amd import _lib, engine
from dasr_amd.engine import OpList, Op
engine.ensure_runtime_ready()
buf = torch.zeros(1 << 20, device='cuda')
for n in (256, 1 << 20):
    ol = OpList()
    for _ in range(2000):
        o = Op(); o.op = _lib.OP_FILL; o.p[0], o.l[0], o.f[0] = buf.data_ptr(), n, 1.0
        ol.add(o)
    ol.run(); torch.cuda.synchronize()
    t0 = time.perf_counter(); ol.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('fill %7d floats: host enqueue %.2f us/launch, GPU-complete %.2f us/launch' % (n, (t1 - t0) / 2000 * 1e6, (t2 - t0) / 2000 * 1e6))
