import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden_dsn import dsn_state, dsn_batch
def rel(a, b): return float((a.double() - b.double()).norm() / max(1e-30, float(b.double().norm())))
def run(b16, lo, hi, crop=256):
    os.environ['DASR_DSN_BWD16'] = str(b16)
    from dasr_amd.dsn_model import DSNModel
    hr, bic, real = dsn_batch(dict(n=8, crop=crop))
    torch.manual_seed(0)
    m = DSNModel(dict(filter='wavelet', w_per=0.01, vgg_seed=78, allow_random_perceptual=True), device='cuda')
    m.netG.load_state_dict(dsn_state(m.netG.state_dict(), 21, 0.5))
    m.netD.load_state_dict(dsn_state(m.netD.state_dict(), 22, 1.0))
    m.iteration(hr[lo:hi].cuda(), bic[lo:hi].cuda(), real[lo:hi].cuda())
    torch.cuda.synchronize()
    g = m.netG.params.grad_dict()
    del m; torch.cuda.empty_cache()
    return g
crop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f0 = run(0, 0, 8, crop); f1 = run(1, 0, 8, crop); a1 = run(1, 0, 4, crop); b1 = run(1, 4, 8, crop)
rows = []
for k in f0:
    rows.append((k, rel(f1[k], f0[k]), rel(f1[k], 0.5 * (a1[k] + b1[k]))))
rows.sort(key=lambda r: -r[2])
print('tensor, 16-bit bwd vs fp32-tensor bwd (batch 8), 16-bit batch 8 vs mean of halves')
for r in rows[:12]: print('  %-34s %.2e  %.2e' % r)
print('worst 16 vs 32:', max(rows, key=lambda r: r[1]))
