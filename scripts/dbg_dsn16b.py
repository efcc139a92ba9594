import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden_dsn import dsn_state, dsn_batch
def rel(a, b): return float((a.double() - b.double()).norm() / max(1e-30, float(b.double().norm())))
def run(lo, hi, crop=256, w_per=0.01):
    from dasr_amd.dsn_model import DSNModel
    hr, bic, real = dsn_batch(dict(n=8, crop=crop))
    torch.manual_seed(0)
    m = DSNModel(dict(filter='wavelet', w_per=w_per, vgg_seed=78, allow_random_perceptual=True), device='cuda')
    m.netG.load_state_dict(dsn_state(m.netG.state_dict(), 21, 0.5))
    m.netD.load_state_dict(dsn_state(m.netD.state_dict(), 22, 1.0))
    m.iteration(hr[lo:hi].cuda(), bic[lo:hi].cuda(), real[lo:hi].cuda())
    torch.cuda.synchronize()
    g = m.netG.params.grad_dict()
    del m; torch.cuda.empty_cache()
    return g
for b16, vp, wper in ((0, 4, 0.01), (0, 5, 0.01), (1, 4, 0.01), (1, 5, 0.01), (1, 5, 0.0)):
    os.environ['DASR_DSN_BWD16'] = str(b16); os.environ['DASR_VGG_PREC'] = str(vp)
    f, a, b = run(0, 8, w_per=wper), run(0, 4, w_per=wper), run(4, 8, w_per=wper)
    rows = sorted(((k, rel(f[k], 0.5 * (a[k] + b[k]))) for k in f), key=lambda r: -r[1])
    print('BWD16 %d VGG_PREC %d w_per %g: batch 8 vs mean of halves, worst:' % (b16, vp, wper), ['%s %.1e' % r for r in rows[:4]])
