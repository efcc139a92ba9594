import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch.nn.functional as F
import test_gpu_kernels as T
from dasr_amd.engine import BTensor, OpList, conv_op, PackRegistry
dev = torch.device('cuda')
for cin, cout in ((40, 64), (64, 32), (16, 32)):
    N, H, W = 2, 20, 36
    mt = 2 if cout % 64 == 0 else 1
    c16 = lambda c: (c + 15) // 16 * 16
    w, b, P, pack, ref = T.make_conv(cout, cin, 3, mt, 1, dev, 29)
    pack = PackRegistry(P)
    ref = pack.add(cout, 3 * c16(cin), 9, mt, 5, [(0, cout, cin, 0, cin, 0, 0)])
    pack.finalize(); pack.run()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, cin, H, W, generator=g)
    xs = T._to_split(x, dev, True)
    xv = T._from_split(xs, cin).double()
    wv = (w.half().float() + (w - w.half().float()).half().float()).double()
    y = F.conv2d(xv, wv, None, padding=1)
    y1 = F.conv2d(xs.t[:, :c16(cin)//16].float().permute(0,1,4,2,3).reshape(N, -1, H, W)[:, :cin].cpu().double(), w.half().double(), None, padding=1)
    kin = c16(cin) // 16
    for lo in (0, c16(cout) // 16):
        out = BTensor(N, 2 * c16(cout), H, W, False, dev, f16=True)
        ops = OpList()
        ops.add(conv_op(pack, ref, xs.view(), False, 3 * c16(cin), H, W, H, W, N, out_bf16=out.view(), out16_f16=1, in_wrap=2 * kin, out16_lo=lo))
        ops.run(); torch.cuda.synchronize()
        K = c16(cout) // 16
        hi = out.t[:, :K].float().permute(0,1,4,2,3).reshape(N, K*16, H, W)[:, :cout].cpu()
        lov = out.t[:, K:].float().permute(0,1,4,2,3).reshape(N, K*16, H, W)[:, :cout].cpu()
        print('cin %d cout %d lo %d: hi vs y %.3e  hi vs hi-only conv %.3e  hi+lo vs y %.3e' % (cin, cout, lo, T.rel(hi, y.float()), T.rel(hi, y1.float()), T.rel(hi + lov, y.float())))
        e = (hi + lov - y.float()).abs()
        print('   per image', e.amax(dim=(1,2,3)).tolist(), 'per 16ch', [round(float(e[:, c:c+16].max()), 4) for c in range(0, cout, 16)], 'rows>=16', float(e[:, :, 16:].max()), 'cols>=32', float(e[:, :, :, 32:].max()))
    print('y  ', y[0, 0, 0, :6].tolist())
    print('hi ', hi[0, 0, 0, :6].tolist())
    print('lo ', lov[0, 0, 0, :6].tolist())
    print('y-hi', (y.float() - hi)[0, 0, 0, :6].tolist())
    print('lo plane raw ch 0..15 px0', out.t[0, K, 0, 0].float().tolist())
    print('hi plane raw ch 0..15 px0', out.t[0, 0, 0, 0].float().tolist())
    print('y ch 0..15 px0', y[0, :16, 0, 0].tolist())
