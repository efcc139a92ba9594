import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.gen_golden_dsn import dsn_state, dsn_batch
from dasr_amd.dsn_model import DSNModel
for n, crop in ((1, 160), (2, 128)):
    junk = torch.full((3 * 2**28,), float('nan'), device='cuda'); j2 = torch.full((2**28,), float('nan'), dtype=torch.float16, device='cuda'); del junk, j2   # poison the allocator's free blocks
    hr, bic, real = dsn_batch(dict(n=n, crop=crop))
    torch.manual_seed(0)
    m = DSNModel(dict(filter='avg_pool', kernel_size=5, w_per=0.01, vgg_seed=78, allow_random_perceptual=True), device='cuda')
    m.netG.load_state_dict(dsn_state(m.netG.state_dict(), 21, 0.5))
    m.netD.load_state_dict(dsn_state(m.netD.state_dict(), 22, 1.0))
    try:
        m.iteration(hr.cuda(), bic.cuda(), real.cuda())
        torch.cuda.synchronize()
        print('G grads finite:', {k: bool(torch.isfinite(v).all()) for k, v in m.netG.params.grad_dict().items() if not torch.isfinite(v).all()})
        print(n, crop, 'log', {k: round(v, 5) for k, v in m.get_current_log().items()})
    except Exception as e:
        print(n, crop, 'EXC', str(e)[:80])
    P = m.plan.g if hasattr(m, 'plan') else None
    for name in dir(m):
        pass
    import gc
    plans = [o for o in gc.get_objects() if type(o).__name__ == '_GPlan']
    for P in plans[-1:]:
        for nm in ('x_s', 'g_fake', 'gz_out', 'g_h', 'g_h16', 'fake', 'x_in', 'd1', 'd2', 'g_d1', 'g_d2'):
            t = getattr(P, nm, None)
            if t is not None: print('   ', nm, 'finite', bool(torch.isfinite(t.t.float()).all()), 'absmax', float(t.t.float().abs().max()))
        for i, t in enumerate(P.s16): print('    s16[%d] finite %s absmax %.3g' % (i, bool(torch.isfinite(t.t.float()).all()), float(t.t.float().abs().max())))
        for i, t in enumerate(P.h16): print('    h16[%d] finite %s absmax %.3g' % (i, bool(torch.isfinite(t.t.float()).all()), float(t.t.float().abs().max())))
        for i, t in enumerate(P.g_s16): print('    g_s16[%d] finite %s absmax %.3g' % (i, bool(torch.isfinite(t.t.float()).all()), float(t.t.float().abs().max())))
        print('    gscale', P.gscale)
        for i, t in enumerate(P.s):
            if t is not None: print('    s[%d] finite %s' % (i, bool(torch.isfinite(t.t).all())))
        for i, t in enumerate(P.g_s): print('    g_s[%d] finite %s' % (i, bool(torch.isfinite(t.t).all())))
    del m
