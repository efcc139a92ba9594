"""host enqueue time vs GPU time of one SR step (is the step launch-bound?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dasr_amd import options
from dasr_amd.models import create_model
torch.manual_seed(0)
m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
g = torch.Generator().manual_seed(1234)
data = {'LR': torch.rand(16, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(16, 3, 512, 512, generator=g).cuda()}
step = 0
for _ in range(3):
    step += 1; m.update_learning_rate(); m.feed_data(data); m.optimize_parameters(step)
torch.cuda.synchronize()
for _ in range(4):
    step += 1
    t0 = time.perf_counter()
    m.update_learning_rate(); m.feed_data(data); m.optimize_parameters(step)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue %.2f ms, + wait for GPU %.2f ms = %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
plans = m._out_plans if hasattr(m, '_out_plans') else []
print('ops per step:', sum(len(p.fwd.ops) + len(p.bwd.ops) for p in plans))
