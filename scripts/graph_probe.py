"""hipGraph probe: the gradient part of the SR step (forward + loss + backward of both sub-batch streams, replica
gradient sum; Adam / repack left out: their arguments change every step) captured once with torch.cuda.graph and replayed, against the
same work launched eagerly through dasr_run_ops.   python scripts/graph_probe.py [--batch 16] [--steps 10]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dasr_amd import options
from dasr_amd.models import create_model

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=16); ap.add_argument('--steps', type=int, default=10)
a = ap.parse_args()
m = create_model(options.dict_to_nonedict(bench.make_opt(64, 23)))
g = torch.Generator().manual_seed(1)
data = {'LR': torch.rand(a.batch, 3, 128, 128, generator=g).cuda(), 'HR': torch.rand(a.batch, 3, 512, 512, generator=g).cuda()}
m.optimizer_G.step = lambda lr: None          # gradient part only
m.netG.repack = lambda: None


def step():
    m.feed_data(data)
    m.optimize_parameters(1)


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(3):
    step()
eager = timed(step, a.steps)
t0 = time.perf_counter(); step(); host = (time.perf_counter() - t0) * 1e3   # host time to enqueue one step (no sync)
torch.cuda.synchronize()
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
graph = timed(gr.replay, a.steps)
t0 = time.perf_counter(); gr.replay(); hostg = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
print('batch %d: gradient part eager %.2f ms/step (host enqueue %.2f ms), hipGraph replay %.2f ms/step (host %.2f ms)' % (a.batch, eager, host, graph, hostg))
