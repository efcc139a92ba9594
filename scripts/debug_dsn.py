import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dasr_amd import engine, _lib
engine.ensure_runtime_ready()
from dasr_amd.dsn_model import DeResnetHIP
from oracle import dsn
from oracle.gen_golden_dsn import dsn_state
from tests.test_gpu_dsn import to_blocked
dev = torch.device('cuda')
def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
ref = dsn.DeResnet(); sd = dsn_state(ref.state_dict(), 21, 0.5); ref.load_state_dict(sd)
G = DeResnetHIP(8, device=dev); G.load_state_dict(sd)
g = torch.Generator().manual_seed(11)
x = torch.rand(2, 3, 48, 64, generator=g)
p = G.plan(2, 48, 64); p.x_nchw.copy_(x); p.fwd.run()
acts = {}
def hook(name):
    def f(m, i, o):
        o.retain_grad(); acts[name] = o
    return f
ref.block_input.register_forward_hook(hook('s0'))
for k, rb in enumerate(ref.res_blocks):
    rb.register_forward_hook(hook('s%d' % (k + 1)))
    rb.conv1.register_forward_hook(hook('z%d' % k))
y = ref(x)
gy = torch.randn(y.shape, generator=g)
p.g_fake.t.copy_(to_blocked(gy, dev).t)
(y * gy).sum().backward()
ops = p.bwd.ops
gs_ptrs = {p.g_s[0].t.data_ptr(): p.g_s[0], p.g_s[1].t.data_ptr(): p.g_s[1]}
k_s, k_h = 8, 7
for i, o in enumerate(ops):
    p.bwd.run(i, i + 1)
    torch.cuda.synchronize()
    if o.op != _lib.OP_CONV:
        continue
    outp = o.conv.out_f32.p
    if outp in gs_ptrs and (o.conv.res1.p or (o.conv.out_stride == 2 and o.conv.out_oy == 1 and o.conv.out_ox == 1)):
        got = gs_ptrs[outp].nchw().cpu(); want = acts['s%d' % k_s].grad
        d = (got - want).abs()
        idx = d.flatten().argmax().item()
        print('op %3d  dL/ds%d rel %.3e  maxabs %.3e at %s (ref max %.3e)' % (i, k_s, rel(got, want), d.max(), tuple(torch.unravel_index(torch.tensor(idx), d.shape)), want.abs().max()))
        # error by row / col bands
        print('      row-err', ['%.1e' % v for v in d.amax(dim=(0, 1, 3))[::6].tolist()], 'col-err', ['%.1e' % v for v in d.amax(dim=(0, 1, 2))[::8].tolist()])
        k_s -= 1
    elif outp == p.g_h.t.data_ptr():
        got = p.g_h.nchw().cpu(); want = acts['z%d' % k_h].grad
        d = (got - want).abs()
        print('op %3d  dL/dz%d rel %.3e maxabs %.3e' % (i, k_h, rel(got, want), d.max()))
        k_h -= 1
