"""Where does the gradient error of the BatchNorm source-discriminator step come from?  (TEST INFRASTRUCTURE, dev tool; VERDICT r03 weak #1.)

Case dasr_srcVGG128_gau5_nf32_nb1_n3_32 on the CPU oracle in fp64 = truth.  Variants degrade ONE component at a time to the arithmetic the
HIP path uses there and report the worst per-tensor gradient error of G / D_source against the truth:

  G-bf16      dense-block convs of the generator with bf16 operands (forward, data gradient, weight gradient), rest fp64
  D-x22       D_source conv operands rounded to 22 bits (f16 hi + lo) in forward and data gradient, weight-gradient operands to `wg` format
  D-bn32      D_source BatchNorm evaluated in fp32 (statistics, normalisation, backward)
  all         everything together (what the GPU step does)

    python -m oracle.bn_probe
"""
import copy
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fixtures, nets, trainers

CASE = 'dasr_srcVGG128_gau5_nf32_nb1_n3_32'


def rnd(x, fmt, scale=1.0):
    if fmt == 'exact':
        return x
    d = x.dtype
    if fmt == 'f32':
        return x.float().to(d)
    if fmt == 'bf16':
        return x.to(torch.bfloat16).to(d)
    if fmt == 'bf16x2':
        hi = x.to(torch.bfloat16).to(d)
        return hi + (x - hi).to(torch.bfloat16).to(d)
    if fmt == 'bf16x3':
        hi = x.to(torch.bfloat16).to(d)
        mid = (x - hi).to(torch.bfloat16).to(d)
        return hi + mid + (x - hi - mid).to(torch.bfloat16).to(d)
    if fmt == 'f16':
        return (x * scale).to(torch.float16).to(d) / scale
    if fmt == 'f16x2':
        xs = x * scale
        hi = xs.to(torch.float16).to(d)
        return (hi + (xs - hi).to(torch.float16).to(d)) / scale
    raise ValueError(fmt)


def autoscale(g):
    """power of two that brings max|g| to ~2^-1 (what the HIP path's fixed pre-scales aim at)"""
    m = float(g.abs().max())
    if m == 0.0:
        return 1.0
    import math
    return 2.0 ** (-1 - math.floor(math.log2(m)))


class RConv(torch.autograd.Function):
    """conv2d whose operands are rounded like the HIP kernels round them; accumulation in the tensor dtype (fp64 in this probe)"""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, fx, fw, fg, wgx, wgg):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, fx, fw, fg, wgx, wgg, b is not None)
        return F.conv2d(rnd(x, fx), rnd(w, fw), b, stride, pad)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, pad, fx, fw, fg, wgx, wgg, has_b = ctx.cfg
        s = autoscale(g)
        gx = torch.nn.grad.conv2d_input(x.shape, rnd(w, fw), rnd(g, fg, s), stride, pad)
        gw = torch.nn.grad.conv2d_weight(rnd(x, wgx), w.shape, rnd(g, wgg, s), stride, pad)
        return gx, gw, (g.sum((0, 2, 3)) if has_b else None), None, None, None, None, None, None, None


def patch_convs(mods, fmts):
    for m in mods:
        if isinstance(m, nn.Conv2d):
            m.forward = (lambda mm: (lambda x: RConv.apply(x, mm.weight, mm.bias, mm.stride, mm.padding, *fmts)))(m)


class BN32(torch.autograd.Function):
    """training-mode BatchNorm2d + nothing else, evaluated in fp32 like csrc/gan.hip::bnorm_lrelu_* (two-pass variance, fp32 sums)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, acc):
        xf = x.to(acc)
        mean = xf.mean((0, 2, 3), keepdim=True)
        var = ((xf - mean) ** 2).mean((0, 2, 3), keepdim=True)
        rstd = 1.0 / torch.sqrt(var + eps)
        xh = ((x.float() - mean.float()) * rstd.float())
        ctx.save_for_backward(xh, rstd.float(), gamma)
        ctx.acc = acc
        return (xh * gamma.float().view(1, -1, 1, 1) + beta.float().view(1, -1, 1, 1)).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        xh, rstd, gamma = ctx.saved_tensors
        acc = ctx.acc
        gz = g.float()
        m1 = gz.to(acc).mean((0, 2, 3), keepdim=True).float()
        m2 = (gz * xh).to(acc).mean((0, 2, 3), keepdim=True).float()
        gx = gamma.float().view(1, -1, 1, 1) * rstd * (gz - m1 - xh * m2)
        dg = (gz * xh).to(acc).sum((0, 2, 3))
        db = gz.to(acc).sum((0, 2, 3))
        return gx.to(g.dtype), dg.to(g.dtype), db.to(g.dtype), None, None


def patch_bn(mods, acc):
    for m in mods:
        if isinstance(m, nn.BatchNorm2d):
            m.forward = (lambda mm: (lambda x: BN32.apply(x, mm.weight, mm.bias, mm.eps, acc)))(m)


def build(variant):
    c = fixtures.CASES[CASE]
    netG = nets.RRDBNet(3, 3, c['nf'], c['nb'], 4)
    netG.load_state_dict(fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1))
    netD = nets.NLayerDiscriminator(c['d_in_nc'], n_layers=2)
    netD.load_state_dict(fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0))
    netD2 = nets.Discriminator_VGG_128(c['d_in_nc'], 64)
    netD2.load_state_dict(fixtures.seeded_state_dict(netD2.state_dict(), 3, 1.0))
    g64, d64, s64 = netG.double(), netD.double(), netD2.double()
    t = trainers.DASRTrainer(fixtures.make_opt(CASE), netG=g64, netD=d64, netF=None, vgg_seed=77, netD_source=s64)
    for v in vars(t).values():
        if isinstance(v, nn.Module):
            v.double()
    v = variant
    if v.get('g'):   # generator dense blocks: one format for all five roundings, or a tuple (fwd x, fwd w / dgrad w, dgrad g, wgrad x, wgrad g)
        rdb = [m for n, m in g64.named_modules() if 'RDB' in n]
        patch_convs(rdb, v['g'] if isinstance(v['g'], tuple) else (v['g'],) * 5)
    if v.get('g_stream'):
        oth = [m for n, m in g64.named_modules() if 'RDB' not in n]
        patch_convs(oth, (v['g_stream'],) * 5)
    if v.get('d'):
        fx, wg = v['d'], v.get('d_wg', v['d'])
        patch_convs(list(s64.modules()), (fx, fx, fx, wg, wg))
        # the two Linear layers run as convs on the same kernels
        for lin in (s64.linear1, s64.linear2):
            lin.forward = (lambda mm: (lambda x: RConv.apply(x.reshape(x.shape[0], -1, 1, 1), mm.weight.reshape(mm.out_features, -1, 1, 1), mm.bias, 1, 0,
                                                             fx, fx, fx, wg, wg).reshape(x.shape[0], -1)))(lin)
    if v.get('bn'):
        patch_bn(list(s64.modules()), v['bn'])
    batch = fixtures.make_batch(CASE)
    t.update_learning_rate()
    t.feed_data({k: x.double() for k, x in batch.items()})
    t.optimize_parameters(1)
    return ({n: p.grad.detach().clone() for n, p in g64.named_parameters()}, {n: p.grad.detach().clone() for n, p in s64.named_parameters()},
            {n: p.grad.detach().clone() for n, p in d64.named_parameters()})


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def worst(got, ref):
    e = [(rel(got[k], ref[k]), k) for k in ref]
    return max(e)


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    refG, refS, refD = build({})
    variants = [
        ('fp32 everything (the CPU reference)', None),
        ('G-bf16 (dense blocks bf16, stream convs bf16x2)', dict(g='bf16', g_stream='bf16x2')),
        ('G dense blocks: only the WEIGHTS bf16 in the forward', dict(g=('exact', 'bf16', 'exact', 'exact', 'exact'))),
        ('G dense blocks: only the ACTIVATIONS bf16 in the forward', dict(g=('bf16', 'exact', 'exact', 'exact', 'exact'))),
        ('G dense blocks: data gradient, g bf16 (w exact)', dict(g=('exact', 'exact', 'bf16', 'exact', 'exact'))),
        ('G dense blocks: weight gradient operands bf16', dict(g=('exact', 'exact', 'exact', 'bf16', 'bf16'))),
        ('G dense blocks all f16 (11-bit) instead of bf16', dict(g='f16', g_stream='bf16x2')),
        ('D-x22 fwd/dgrad f16x2, wgrad f16', dict(d='f16x2', d_wg='f16')),
        ('D-x22 fwd/dgrad f16x2, wgrad f16x2', dict(d='f16x2', d_wg='f16x2')),
        ('D bf16x3 everywhere', dict(d='bf16x3', d_wg='bf16x3')),
        ('D-bn32 (fp32 BatchNorm, fp32 sums)', dict(bn=torch.float32)),
        ('D-bn32 with fp64 sums', dict(bn=torch.float64)),
        ('all: G-bf16 + D f16x2/f16 + bn32', dict(g='bf16', g_stream='bf16x2', d='f16x2', d_wg='f16', bn=torch.float32)),
        ('all, wgrad f16x2, bn sums fp64', dict(g='bf16', g_stream='bf16x2', d='f16x2', d_wg='f16x2', bn=torch.float64)),
        ('all, D bf16x3, bn sums fp64', dict(g='bf16', g_stream='bf16x2', d='bf16x3', d_wg='bf16x3', bn=torch.float64)),
    ]
    sel = sys.argv[1:]
    for name, v in variants:
        if sel and not any(s in name for s in sel):
            continue
        if v is None:
            # plain fp32 run of the same trainer
            c = fixtures.CASES[CASE]
            netG = nets.RRDBNet(3, 3, c['nf'], c['nb'], 4)
            netG.load_state_dict(fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1))
            netD = nets.NLayerDiscriminator(c['d_in_nc'], n_layers=2)
            netD.load_state_dict(fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0))
            netD2 = nets.Discriminator_VGG_128(c['d_in_nc'], 64)
            netD2.load_state_dict(fixtures.seeded_state_dict(netD2.state_dict(), 3, 1.0))
            t = trainers.DASRTrainer(fixtures.make_opt(CASE), netG=netG, netD=netD, netF=None, vgg_seed=77, netD_source=netD2)
            t.update_learning_rate()
            t.feed_data(fixtures.make_batch(CASE))
            t.optimize_parameters(1)
            gG = {n: p.grad for n, p in netG.named_parameters()}
            gS = {n: p.grad for n, p in netD2.named_parameters()}
        else:
            gG, gS, _ = build(v)
        wg, ws = worst(gG, refG), worst(gS, refS)   # (fw also rounds the data gradient's weights: RConv passes one weight format)
        print('%-52s G worst %.2e (%s)   D_source worst %.2e (%s)' % (name, wg[0], wg[1], ws[0], ws[1]))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
