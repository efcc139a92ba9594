"""How many MFMA passes do the split-f16 forward convs need?  (TEST INFRASTRUCTURE, dev tool.)

The HIP path evaluates the DSN generator and the perceptual VGG on split f16 tensors with three MFMA passes (x_hi w_hi + x_hi w_lo + x_lo w_hi:
22-bit operands).  A two-pass product keeps the 22-bit ACTIVATIONS and rounds the frozen / packed WEIGHTS to one f16 (x_hi w_hi + x_lo w_hi).
This probe runs the oracle's DSN iteration in fp64 (= truth), then with the weights of the selected convs rounded to f16 (forward and backward
see the same rounded weights, as the HIP plan would), and reports the error of the generated image, of the generator gradients and of the
perceptual gradient dL/d(fake).

    python -m oracle.pass_probe
"""
import sys

import torch
import torch.nn as nn

from . import dsn, fixtures, nets
from .gen_golden_dsn import DSN_CASES, dsn_batch, dsn_state


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


def round_w(mods, fmt):
    for m in mods:
        if isinstance(m, nn.Conv2d):
            w = m.weight.data
            if fmt == 'f16':
                m.weight.data = w.to(torch.float16).to(w.dtype)
            elif fmt == 'bf16x2':
                hi = w.to(torch.bfloat16).to(w.dtype)
                m.weight.data = hi + (w - hi).to(torch.bfloat16).to(w.dtype)


def run(case, g_fmt=None, vgg_fmt=None, crop=None):
    c = dict(DSN_CASES[case])
    if crop:
        c['crop'] = crop
    torch.manual_seed(0)
    G = dsn.DeResnet()
    D = dsn.Discriminator(c['k'], c['norm'], c['filter'])
    G.load_state_dict(dsn_state(G.state_dict(), 21, 0.5))
    D.load_state_dict(dsn_state(D.state_dict(), 22, 1.0))
    G.double(), D.double()
    if g_fmt:
        round_w([m for r in G.res_blocks for m in r.modules()], g_fmt)
    t = dsn.DSNTrainer(netG=G, netD=D, filter_type=c['filter'], kernel_size=c['k'], norm_layer=c['norm'])
    t.per.double()
    if vgg_fmt:
        round_w(list(t.per.modules()), vgg_fmt)
    if not callable(getattr(t.color_filter, 'double', None)):
        pass
    else:
        t.color_filter.double()
    hr, bic, real = [x.double() for x in dsn_batch(c)]
    fake = G(hr)
    ft = D(fake)
    tex = torch.mean(-torch.log(ft + 1e-8))
    col = torch.nn.functional.l1_loss(t.color_filter(fake), t.color_filter(bic))
    per = torch.nn.functional.mse_loss(t.per(fake), t.per(bic))
    g_per = torch.autograd.grad(per, fake, retain_graph=True)[0]
    loss = col + 0.005 * tex + 0.01 * per
    gg = torch.autograd.grad(loss, list(G.parameters()))
    return fake.detach(), [g.detach() for g in gg], g_per.detach(), float(per)


def main():
    torch.set_num_threads(16)
    case = 'dsn_gau5_inst_b2_128'
    crop = int(sys.argv[1]) if len(sys.argv) > 1 else None
    f0, g0, p0, l0 = run(case, crop=crop)
    names = [n for n, _ in dsn.DeResnet().named_parameters()]
    for label, kw in (('generator residual-block weights f16 (two passes)', dict(g_fmt='f16')),
                      ('perceptual VGG16 weights f16 (two passes)', dict(vgg_fmt='f16')),
                      ('both', dict(g_fmt='f16', vgg_fmt='f16'))):
        f, g, p, l = run(case, crop=crop, **kw)
        e = [(rel(a, b), n) for a, b, n in zip(g, g0, names)]
        print('%-52s fake %.2e   G grads worst %.2e (%s) median %.2e   dL_per/dfake %.2e   per loss %.2e' %
              (label, rel(f, f0), max(e)[0], max(e)[1], sorted(x[0] for x in e)[len(e) // 2], rel(p, p0), abs(l - l0) / abs(l0)))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
