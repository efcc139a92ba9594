"""Operand-precision emulation on the CPU oracle (TEST INFRASTRUCTURE; dev tool behind the numerics recipe in DESIGN.md).

Every conv of the oracle RRDBNet is replaced by an autograd function that rounds its operands to a chosen format (fp32 accumulate,
fp32 storage), separately for the forward product, the data-gradient and the weight gradient -- the same places the HIP kernels round.
Prints the normwise relative error of the tapped activations and the worst per-tensor weight-gradient error against the fp32 run.

    python -m oracle.precision_probe [nb] [lr_size]
"""
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fixtures, nets


def rnd(x, fmt, scale=1.0):
    if fmt == 'f32':
        return x
    if fmt == 'bf16':
        return x.to(torch.bfloat16).float()
    if fmt == 'f16':      # scaled so that tiny gradients stay in the normal range (exact power of two)
        return (x * scale).to(torch.float16).float() / scale
    if fmt == 'bf16x2':   # hi + lo split: ~2^-17
        hi = x.to(torch.bfloat16).float()
        return hi + (x - hi).to(torch.bfloat16).float()
    raise ValueError(fmt)


_BT = torch.tensor([[1., 0., -1., 0.], [0., 1., 1., 0.], [0., -1., 1., 0.], [0., 1., 0., -1.]])
_G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]])
_AT = torch.tensor([[1., 1., 1., 0.], [0., 1., -1., -1.]])


def winograd_conv(x, w, op_fmt='bf16'):
    """3x3 / pad 1 conv as Winograd F(2x2, 3x3) with the rounding an MFMA implementation would have: the operands arrive in bf16 (storage), the
    TRANSFORMED input tiles B^T d B and weights G g G^T are rounded to `op_fmt` again (they are what the matrix cores multiply), products and the
    output transform A^T m A stay in fp32.  ('emulate the transform rounding ... if gradients stay < 1e-2 build a micro-kernel')"""
    N, C_, H, W = x.shape
    K = w.shape[0]
    assert H % 2 == 0 and W % 2 == 0
    xp = F.pad(rnd(x, 'bf16'), (1, 1, 1, 1))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                               # [N, C, H/2, W/2, 4, 4]
    V = rnd(torch.einsum('ij,nchwjk,lk->nchwil', _BT, d, _BT), op_fmt)
    U = rnd(torch.einsum('ij,kcjl,ml->kcim', _G, rnd(w, 'bf16'), _G), op_fmt)
    M = torch.einsum('kcim,nchwim->nkhwim', U, V)
    Y = torch.einsum('ai,nkhwim,bm->nkhwab', _AT, M, _AT)                # [N, K, H/2, W/2, 2, 2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, K, H, W)


class RConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, fx, fw, fg):
        ctx.save_for_backward(x, w)
        ctx.f = (fx, fw, fg)
        if fx == 'wino':   # Winograd on bf16 operands (forward and data gradient), direct bf16 weight gradient
            return winograd_conv(x, w) + b.view(1, -1, 1, 1)
        return F.conv2d(rnd(x, fx), rnd(w, fw), b, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        fx, fw, fg = ctx.f
        if fx == 'wino':
            gx = winograd_conv(g, w.transpose(0, 1).flip(2, 3))
            gw = torch.nn.grad.conv2d_weight(rnd(x, 'bf16'), w.shape, rnd(g, 'bf16'), 1, 1)
            return gx, gw, g.sum((0, 2, 3)), None, None, None
        gs = 2.0 ** 24 if fg == 'f16' else 1.0
        gr = rnd(g, fg, gs)
        gx = torch.nn.grad.conv2d_input(x.shape, rnd(w, fw), gr, 1, 1)
        gw = torch.nn.grad.conv2d_weight(rnd(x, fx), w.shape, gr, 1, 1)
        return gx, gw, g.sum((0, 2, 3)), None, None, None


def patch(net, recipe):
    """recipe: name-class -> (fmt_x, fmt_w, fmt_g); classes: rdb, fea, lr, up, hr0, hr1"""
    convs = [(n, m) for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
    nb = len(net.model[1].sub.sub) - 1 if hasattr(net.model[1].sub, 'sub') else None
    for n, m in convs:
        if 'RDB' in n:
            cls = 'rdb'
        elif n == 'model.0':
            cls = 'fea'
        elif n.startswith('model.1.sub'):
            cls = 'lr'
        elif n in ('model.3', 'model.6'):
            cls = 'up'
        elif n == 'model.8':
            cls = 'hr0'
        else:
            cls = 'hr1'
        f = recipe[cls]
        m.forward = (lambda mm, ff: (lambda x: RConv.apply(x, mm.weight, mm.bias, *ff)))(m, f)


def run(nb, lr, recipe):
    torch.manual_seed(0)
    net = nets.RRDBNet(3, 3, 64, nb, 4)
    net.load_state_dict(fixtures.seeded_state_dict(net.state_dict(), 1, 0.1))
    if recipe is not None:
        patch(net, recipe)
    g = torch.Generator().manual_seed(1234)
    x, hr = torch.rand(1, 3, lr, lr, generator=g), torch.rand(1, 3, 4 * lr, 4 * lr, generator=g)
    taps = {}
    hs = [net.model[1].sub[i].register_forward_hook(lambda m, i_, o, k=i: taps.__setitem__('rrdb%d' % k, o.detach())) for i in (0, nb // 2, nb - 1)]
    hs.append(net.model[1].register_forward_hook(lambda m, i_, o: taps.__setitem__('trunk', o.detach())))
    sr = net(x)
    taps['sr'] = sr.detach()
    (sr - hr).abs().mean().backward()
    return taps, [p.grad.detach().clone() for p in net.parameters()], [n for n, _ in net.named_parameters()]


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 23
    lr = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    torch.set_num_threads(8)
    ref_t, ref_g, names = run(nb, lr, None)
    S3, BF = ('bf16x2', 'bf16x2', 'bf16x2'), ('bf16', 'bf16', 'bf16')
    F16 = ('f16', 'f16', 'f16')
    F16X = ('f16', 'bf16x2', 'f16')      # activations / gradients rounded once to f16, weights exact (2 MFMA passes)
    WINO = ('wino', 'wino', 'wino')
    recipes = {
        'W  rdb WINOGRAD F(2x2,3x3) on bf16 (transformed operands re-rounded to bf16) | others as A': dict(rdb=WINO, fea=S3, lr=S3, up=S3, hr0=S3, hr1=S3),
        'A  rdb bf16 | fea lr up hr0 hr1 split-bf16 (round 1)': dict(rdb=BF, fea=S3, lr=S3, up=S3, hr0=S3, hr1=S3),
        'E  rdb bf16 | fea lr hr1 split | up hr0 f16 1-pass': dict(rdb=BF, fea=S3, lr=S3, up=F16, hr0=F16, hr1=S3),
        'E2 rdb bf16 | fea lr hr1 split | up hr0 f16 x exact-w 2-pass': dict(rdb=BF, fea=S3, lr=S3, up=F16X, hr0=F16X, hr1=S3),
        'D  rdb bf16 | all six f16 1-pass': dict(rdb=BF, fea=F16, lr=F16, up=F16, hr0=F16, hr1=F16),
        'F  rdb bf16 | fea lr split | up hr0 hr1 f16': dict(rdb=BF, fea=S3, lr=S3, up=F16, hr0=F16, hr1=F16),
    }
    for name, rc in recipes.items():
        t, g, _ = run(nb, lr, rc)
        acts = ' '.join('%s %.1e' % (k, rel(t[k], ref_t[k])) for k in ref_t)
        errs = [(rel(a, b), n) for a, b, n in zip(g, ref_g, names)]
        w = max(errs)
        tail = max((e, n) for e, n in errs if n.split('.')[1] in ('3', '6', '8', '10'))
        print('%-62s acts: %s | worst grad %.1e (%s); worst HR-tail grad %.1e (%s)' % (name, acts, w[0], w[1], tail[0], tail[1]))


if __name__ == '__main__':
    main()
