"""Randomised parity sweep of the conv / weight-gradient kernels through the C ABI (dasr_run_ops) against fp64 torch on the CPU: shapes, channel
counts, kernel sizes / strides, epilogue term sets and tensor formats drawn at random inside the domain the host code uses -- the parametrised
tests pin chosen cases, this looks for the ones nobody chose (partial tiles, channel tails, one-pixel images, odd strides, unusual term sets).

    python scripts/fuzz_kernels.py [--cases 300] [--seed 0]        # prints one line per failure and a summary; exit code 1 on any failure
"""
import argparse
import math
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def bf16r(x):
    return x.to(torch.bfloat16).float()


def f16r(x):
    return x.to(torch.float16).float()


def to_blocked(x, f32, dev, f16=False):
    from dasr_amd.engine import BTensor
    N, C_, H, W = x.shape
    b = BTensor(N, C_, H, W, f32, dev, f16=f16)
    xp = torch.zeros(N, b.planes * 16, H, W)
    xp[:, :C_] = x
    b.t.copy_(xp.view(N, b.planes, 16, H, W).permute(0, 1, 3, 4, 2).to(b.t.dtype))
    return b


# (kh, stride, pad) the generic kernel is instantiated for (csrc/conv.hip dispatch keys 1110-1116 / 3110-3116)
GENERIC_GEOM = [(3, 1, 1), (4, 1, 1), (4, 2, 1), (5, 1, 2), (1, 1, 0), (3, 2, 1)]


def draw_conv(rng):
    fam = rng.choice(['g3', 'g3', 'g4', 'dense', 'dense', 'dense16', 'vggf32'])
    c = {'fam': fam, 'N': rng.choice([1, 1, 2, 3])}
    if fam in ('g3', 'g4'):
        c['prec'], c['in_f32'], c['mt'] = (3 if fam == 'g3' else 4), True, 1
        c['kh'], c['stride'], c['pad'] = rng.choice(GENERIC_GEOM)
        c['cin'], c['cout'] = rng.choice([1, 3, 9, 16, 17, 40, 64, 80]), rng.choice([1, 3, 16, 24, 32, 33, 64, 100])
    elif fam == 'vggf32':
        c['prec'], c['in_f32'], c['mt'] = 1, True, rng.choice([1, 2])
        c['kh'], c['stride'], c['pad'] = 3, 1, 1
        c['cin'], c['cout'] = rng.choice([3, 16, 48, 64]), rng.choice([16, 32, 64, 128])
    else:
        c['prec'], c['in_f32'] = (1 if fam == 'dense' else 2), False
        c['kh'], c['stride'], c['pad'] = 3, 1, 1
        c['cin'], c['cout'] = rng.choice([16, 32, 64, 96, 160, 192]), rng.choice([3, 16, 32, 48, 64, 128])
        c['mt'] = 1 if c['cout'] <= 32 else rng.choice([1, 2])
    c['ups'] = int(fam in ('g3', 'dense16') and c['kh'] == 3 and c['stride'] == 1 and rng.random() < 0.25)
    lo = max(1, c['kh'] - 2 * c['pad'])
    c['H'], c['W'] = rng.randint(lo, 45), rng.randint(lo, 70)
    if rng.random() < 0.15:
        c['H'], c['W'] = rng.choice([(lo, lo), (16, 32), (32, 32), (17, 33), (1 + lo, 64)])
    c['bias'], c['act'] = rng.random() < 0.6, rng.choice([0, 1])
    c['mask'], c['res1'], c['res2'] = rng.random() < 0.3, rng.random() < 0.4, rng.random() < 0.2
    c['alpha'], c['gamma'] = rng.choice([1.0, 0.2]), rng.choice([1.0, 0.5])
    c['out_f32'], c['out_16'] = True, rng.random() < 0.6
    if fam in ('dense', 'dense16') and rng.random() < 0.5:
        c['out_f32'], c['out_16'] = False, True
    return c


def run_conv(c, dev, seed):
    from dasr_amd.engine import BTensor, OpList, ParamStore, PackRegistry, conv_op
    g = torch.Generator().manual_seed(seed)
    N, cin, cout, kh, stride, pad, H, W = c['N'], c['cin'], c['cout'], c['kh'], c['stride'], c['pad'], c['H'], c['W']
    f16t = c['fam'] == 'dense16'
    w = torch.randn(cout, cin, kh, kh, generator=g) * math.sqrt(2.0 / (cin * kh * kh))
    b = torch.randn(cout, generator=g) * 0.1
    P = ParamStore([('w', (cout, cin, kh, kh)), ('b', (cout,))], dev)
    P.load_state_dict({'w': w, 'b': b})
    pack = PackRegistry(P)
    cin_pad = (cin + 15) // 16 * 16
    ref = pack.add(cout, cin_pad, kh * kh, c['mt'], c['prec'], [(0, cout, cin, 0, cin, 0, 0)])
    pack.finalize()
    pack.run()
    x = torch.randn(N, cin, H, W, generator=g)
    HL, WL = (2 * H, 2 * W) if c['ups'] else (H, W)
    Ho, Wo = (HL + 2 * pad - kh) // stride + 1, (WL + 2 * pad - kh) // stride + 1
    rnd16 = f16r if f16t else bf16r
    xin = x if c['in_f32'] else rnd16(x)
    xb = to_blocked(xin, c['in_f32'], dev, f16=f16t)
    res1, res2, msk = (torch.randn(N, cout, Ho, Wo, generator=g) for _ in range(3))
    kw, keep = {}, []   # (every blocked tensor an op points at stays referenced until the op has run: the ops hold raw device pointers)
    if c['mask']:
        mq = msk if c['in_f32'] else rnd16(msk)
        keep.append(to_blocked(mq, c['in_f32'], dev, f16=f16t))
        kw.update(mask=keep[-1].view(), mask_f32=int(c['in_f32']))
    if c['res1']:
        keep.append(to_blocked(res1, True, dev))
        kw.update(res1=keep[-1].view(), beta1=1.0)
    if c['res2']:
        keep.append(to_blocked(res2, True, dev))
        kw.update(res2=keep[-1].view(), beta2=0.5)
    of = BTensor(N, cout, Ho, Wo, True, dev) if c['out_f32'] else None
    ob = BTensor(N, cout, Ho, Wo, False, dev, f16=f16t) if c['out_16'] else None
    ops = OpList()
    ops.add(conv_op(pack, ref, xb.view(), c['in_f32'], cin_pad, H, W, Ho, Wo, N, bias=P.ptr('b') if c['bias'] else None, kh=kh, stride=stride, pad=pad,
                    ups=c['ups'], act=c['act'], alpha=c['alpha'], gamma=c['gamma'], out_f32=of.view() if of else None, out_bf16=ob.view() if ob else None,
                    out16_f16=int(f16t), **kw))
    ops.run()
    torch.cuda.synchronize()
    prec = c['prec']
    xr = xin if prec in (3, 4) else (f16r(xin) if prec == 2 else bf16r(xin))
    wr = w if prec in (3, 4) else (f16r(w) if prec == 2 else bf16r(w))
    if c['ups']:
        xr = F.interpolate(xr, scale_factor=2, mode='nearest')
    y = F.conv2d(xr.double(), wr.double(), b.double() if c['bias'] else None, stride=stride, padding=pad)
    if c['act']:
        y = F.leaky_relu(y, 0.2)
    if c['mask']:
        y = torch.where((msk if c['in_f32'] else rnd16(msk)).double() > 0, y, y * 0.2)
    y = y * c['alpha']
    if c['res1']:
        y = y + res1.double()
    if c['res2']:
        y = y + 0.5 * res2.double()
    errs = {}
    tol = 3e-5 if prec != 4 else 3e-6
    if of is not None:
        errs['f32'] = (rel(of.nchw().cpu(), y.float()), tol)
        if cout % 16:
            errs['pad'] = (float(of.t[:, -1, :, :, cout % 16:].abs().max()), 1e-30)
    if ob is not None:
        errs['16'] = (rel(ob.nchw().cpu(), rnd16((y * c['gamma']).float())), 6e-3 if not f16t else 8e-4)
    return errs


def draw_wgrad(rng):
    c = {'N': rng.choice([1, 2, 3])}
    c['kh'], c['stride'], c['pad'] = rng.choice([(3, 1, 1), (3, 1, 1), (4, 1, 1), (4, 2, 1), (5, 1, 2), (1, 1, 0), (3, 2, 1)])
    c['cin'], c['cout'] = rng.choice([3, 9, 16, 40, 64, 100, 128]), rng.choice([1, 3, 32, 48, 64, 128])
    lo = max(2, c['kh'] - 2 * c['pad'])
    c['H'], c['W'] = rng.randint(lo, 40), rng.randint(lo, 56)
    c['f16'] = rng.random() < 0.5
    return c


def run_wgrad(c, dev, seed):
    from dasr_amd.engine import OpList, ParamStore, WgradGroup, Workspace
    g = torch.Generator().manual_seed(seed)
    N, cin, cout, kh, stride, pad, H, W = c['N'], c['cin'], c['cout'], c['kh'], c['stride'], c['pad'], c['H'], c['W']
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kh) // stride + 1
    x = torch.randn(N, cin, H, W, generator=g)
    go = torch.randn(N, cout, Ho, Wo, generator=g)
    P = ParamStore([('w', (cout, cin, kh, kh)), ('b', (cout,))], dev)
    xb, gb = to_blocked(x, True, dev), to_blocked(go, True, dev)
    ws = Workspace(dev)
    grp = WgradGroup(kh, stride)
    grp.add_conv(gb.view, True, gb.planes, xb.view, True, xb.planes, cout, cin, H, W, Ho, Wo, N, P.off('w'), P.off('b'), pad=pad, f16=c['f16'], g_scale=1.0 if c['f16'] else 0.0)
    grp.finalize(ws, dev)
    wl = OpList()
    for o in grp.ops(P.grad.data_ptr()):
        wl.add(o)
    ws.finalize()
    wl.run()
    torch.cuda.synchronize()
    rq = f16r if c['f16'] else bf16r   # the kernel stages both f32 operands as 16-bit values (fp32 accumulation); the bias sums the unrounded gradient
    wr = torch.zeros(cout, cin, kh, kh, dtype=torch.float64, requires_grad=True)
    (F.conv2d(rq(x).double(), wr, None, stride=stride, padding=pad) * rq(go).double()).sum().backward()
    gd = P.grad_dict()
    return {'w': (rel(gd['w'], wr.grad.float()), 3e-5), 'b': (rel(gd['b'], go.double().sum((0, 2, 3)).float()), 3e-5)}


def draw_wgrad3(rng):
    """the GROUPED 3x3 / stride-1 weight gradient on 16-bit tensors (wgrad3_ld_kernel, register-window form of round 6): one 64-channel input block x one to three
    32-oc tiles per part, several parts per launch, partial pixel tiles, 32-channel input blocks, nearest-x2 input, f16 / bf16"""
    c = {'N': rng.choice([1, 2, 3]), 'cin': rng.choice([32, 64, 96, 128]), 'cout': rng.choice([32, 64, 96, 128, 160]), 'f16': rng.random() < 0.5,
         'ups': rng.random() < 0.2}
    c['H'], c['W'] = rng.randint(2, 40), rng.randint(2, 56)
    return c


def run_wgrad3(c, dev, seed):
    from dasr_amd.engine import OpList, ParamStore, WgradGroup3, Workspace, BTensor, ceil_div
    g = torch.Generator().manual_seed(seed)
    N, cin, cout, H, W, ups, f16 = c['N'], c['cin'], c['cout'], c['H'], c['W'], int(c['ups']), c['f16']
    Hi, Wi = H, W
    Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
    rq = f16r if f16 else bf16r
    x, go = rq(torch.randn(N, cin, Hi, Wi, generator=g)), rq(torch.randn(N, cout, Ho, Wo, generator=g))
    P = ParamStore([('w', (cout, cin, 3, 3)), ('b', (cout,))], dev)
    xb, gb = BTensor(N, cin, Hi, Wi, False, dev, f16=f16), BTensor(N, cout, Ho, Wo, False, dev, f16=f16)
    for bt, src, C_ in ((xb, x, cin), (gb, go, cout)):
        t = torch.zeros(src.shape[0], bt.planes * 16, src.shape[2], src.shape[3])
        t[:, :C_] = src
        bt.t.copy_(t.reshape(src.shape[0], bt.planes, 16, src.shape[2], src.shape[3]).permute(0, 1, 3, 4, 2).contiguous().to(bt.t.dtype))
    ws = Workspace(dev)
    grp = WgradGroup3()
    octs = list(range(0, cout, 32))
    for c0 in range(0, cin, 64):
        blk = min(64, cin - c0)
        for k0 in range(0, len(octs), 3):
            sub = octs[k0:k0 + 3]
            tiles = [dict(dst_w_off=P.off('w'), dst_b_off=P.off('b') if c0 == 0 else None, cout=cout, cin=cin, oc0=oc0, c0=c0, n_ctiles=min(2, ceil_div(blk, 32))) for oc0 in sub]
            grp.add_block(gb.view(sub[0]), min(2 * len(sub), gb.planes - sub[0] // 16), xb.view(c0), ceil_div(blk, 16), ceil_div(blk, 32), Hi, Wi, Ho, Wo, N, tiles,
                          want_bias=(c0 == 0), ups=ups)
    grp.f16, grp.g_scale, grp.flops = f16, 1.0, 0.0
    grp.finalize(ws, dev, target_wgs=rng_targets[seed % len(rng_targets)])
    wl = OpList()
    for o in grp.ops(P.grad.data_ptr()):
        wl.add(o)
    ws.finalize()
    wl.run()
    torch.cuda.synchronize()
    xr = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(xr.double(), wr, None, padding=1) * go.double()).sum().backward()
    gd = P.grad_dict()
    return {'w': (rel(gd['w'], wr.grad.float()), 3e-5), 'b': (rel(gd['b'], go.double().sum((0, 2, 3)).float()), 3e-5)}


rng_targets = (256, 64, 17, 1)   # workgroup targets -> split counts from "every tile its own workgroup" down to ONE workgroup walking all tiles


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=300)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    dev = torch.device('cuda')
    rng = random.Random(a.seed)
    fails, worst, n = 0, {}, {'conv': 0, 'wgrad': 0, 'wgrad3': 0}
    for i in range(a.cases):
        u = rng.random()
        kind = 'conv' if u < 0.65 else ('wgrad' if u < 0.82 else 'wgrad3')
        c = draw_conv(rng) if kind == 'conv' else (draw_wgrad(rng) if kind == 'wgrad' else draw_wgrad3(rng))
        try:
            errs = (run_conv if kind == 'conv' else (run_wgrad if kind == 'wgrad' else run_wgrad3))(c, dev, 1000 + i)
        except Exception as e:   # a refused geometry is reported like a wrong result: the drawn domain is the one the host code uses
            print('CASE %d %s %s raised %s' % (i, kind, c, repr(e)[:200]))
            fails += 1
            continue
        n[kind] += 1
        for k, (e, tol) in errs.items():
            key = (kind, c.get('fam', 'f16' if c.get('f16') else 'bf16'), k)
            worst[key] = max(worst.get(key, 0.0), e)
            if not e <= tol:
                print('CASE %d %s %s: %s error %.3e (tol %.1e)' % (i, kind, c, k, e, tol))
                fails += 1
    print('ran %d conv + %d weight-gradient + %d grouped 3x3 weight-gradient cases, %d failures' % (n['conv'], n['wgrad'], n['wgrad3'], fails))
    for key in sorted(worst):
        print('  worst %-28s %.2e' % (' / '.join(key), worst[key]))
    sys.exit(1 if fails else 0)


if __name__ == '__main__':
    main()
