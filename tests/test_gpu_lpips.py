"""GPU parity of the LPIPS(alex) perceptual loss (dasr_amd/lpips.py + csrc/lpips.hip) against fp32 torch restatements of each layer, the
oracle (oracle/lpips.py) and the fixture made by the reference's own PerceptualLossLPIPS.  Tolerances: activations / loss 1e-3, gradients 1e-2."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ACT_TOL, GRAD_TOL = 1e-3, 1e-2


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def to_blocked(x, dev):
    from dasr_amd.engine import BTensor
    N, Cc, H, W = x.shape
    b = BTensor(N, Cc, H, W, True, dev)
    xp = torch.zeros(N, b.planes * 16, H, W)
    xp[:, :Cc] = x
    b.t.copy_(xp.view(N, b.planes, 16, H, W).permute(0, 1, 3, 4, 2))
    return b


def _call(name, *args):
    from dasr_amd import _lib
    _lib.check(getattr(_lib.lib(), name)(*args))
    torch.cuda.synchronize()


def test_space_to_depth_input_and_adjoint():
    dev = _gpu()
    from dasr_amd.engine import BTensor
    g = torch.Generator().manual_seed(0)
    n, H, W = 2, 40, 52
    x = torch.rand(n, 3, H, W, generator=g)
    sc, sh = (C.c_float * 4)(2.1, 2.2, 2.3, 0), (C.c_float * 4)(-1.0, -0.9, -0.8, 0)
    xb = to_blocked(x, dev)
    Hs, Ws = (H + 4) // 4, (W + 4) // 4
    y = BTensor(n, 48, Hs, Ws, True, dev)
    _call('dasr_lpips_s2d', xb.view(), n, H, W, sc, sh, y.view(), 0, None)
    xs = x * torch.tensor([2.1, 2.2, 2.3]).view(1, 3, 1, 1) + torch.tensor([-1.0, -0.9, -0.8]).view(1, 3, 1, 1)
    xp = F.pad(xs, (2, 2, 2, 2))
    want = xp.view(n, 3, Hs, 4, Ws, 4).permute(0, 1, 3, 5, 2, 4).reshape(n, 48, Hs, Ws)
    assert rel(y.nchw().cpu(), want) < 1e-6 and torch.equal(y.nchw().cpu() == 0, want == 0)   # fma vs mul+add: 1 ulp; the zero padding is exact
    # adjoint: <s2d(x) - shift part, gy> == <x, adj(gy)>, accumulated on top of what is already there
    gy = torch.randn(n, 48, Hs, Ws, generator=g)
    gx0 = torch.randn(n, 3, H, W, generator=g)
    gxb = to_blocked(gx0, dev)
    _call('dasr_lpips_s2d', gxb.view(), n, H, W, sc, sh, to_blocked(gy, dev).view(), 1, None)
    gyp = gy.view(n, 3, 4, 4, Hs, Ws).permute(0, 1, 4, 2, 5, 3).reshape(n, 3, 4 * Hs, 4 * Ws)[:, :, 2:2 + H, 2:2 + W]
    want_g = gx0 + gyp * torch.tensor([2.1, 2.2, 2.3]).view(1, 3, 1, 1)
    assert rel(gxb.nchw(3).cpu(), want_g) < 1e-6


@pytest.mark.parametrize('C_,H,W', [(64, 31, 31), (192, 15, 18), (16, 7, 9)])
def test_maxpool3s2_forward_backward(C_, H, W):
    dev = _gpu()
    from dasr_amd.engine import BTensor
    g = torch.Generator().manual_seed(1)
    n = 2
    z = torch.randn(n, C_, H, W, generator=g)
    z[0, :, :5, :5] = -1.0                      # a dead region: all-zero windows after the ReLU (ties: first element wins, and the ReLU' kills it)
    x = torch.relu(z).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2)
    Ho, Wo = y.shape[2:]
    xb = to_blocked(x.detach(), dev)
    yb = BTensor(n, C_, Ho, Wo, True, dev)
    _call('dasr_maxpool3s2', xb.view(), n, C_, H, W, yb.view(), None)
    assert torch.equal(yb.nchw().cpu(), y.detach())
    gy = torch.randn(y.shape, generator=g)
    zz = z.clone().requires_grad_(True)
    (F.max_pool2d(torch.relu(zz), 3, 2) * gy).sum().backward()      # gradient w.r.t. the pre-activation
    base = torch.randn(n, C_, H, W, generator=g)
    gxb = to_blocked(base, dev)
    _call('dasr_maxpool3s2_bwd', xb.view(), to_blocked(gy, dev).view(), n, C_, H, W, gxb.view(), 1, 1, None)
    assert rel(gxb.nchw().cpu(), base + zz.grad) < 1e-6
    _call('dasr_maxpool3s2_bwd', xb.view(), to_blocked(gy, dev).view(), n, C_, H, W, gxb.view(), 1, 0, None)
    assert rel(gxb.nchw().cpu(), zz.grad) < 1e-6


@pytest.mark.parametrize('C_,H,W', [(64, 15, 15), (384, 7, 9)])
def test_lpips_head_loss_and_gradient(C_, H, W):
    dev = _gpu()
    from dasr_amd.engine import BTensor
    g = torch.Generator().manual_seed(2)
    n = 2
    z = torch.randn(2 * n, C_, H, W, generator=g)
    lin = torch.rand(C_, generator=g)
    zz = z.clone().requires_grad_(True)
    f = torch.relu(zz)
    f0, f1 = f[:n], f[n:]
    u = f0 / (torch.sqrt((f0 ** 2).sum(1, keepdim=True)) + 1e-10)
    v = f1 / (torch.sqrt((f1 ** 2).sum(1, keepdim=True)) + 1e-10)
    val = (((u - v) ** 2) * lin.view(1, -1, 1, 1)).sum(1).mean([1, 2]).mean()
    (0.7 * val).backward()
    fb = to_blocked(f.detach(), dev)
    g0 = BTensor(n, C_, H, W, True, dev)
    acc = torch.zeros(1, device=dev)
    lind = lin.to(dev)
    cnt = float(n * H * W)
    _call('dasr_lpips_head', fb.view(), n, n, C_, H, W, lind.data_ptr(), 1e-10, 1.0 / cnt, 0.7 / cnt, acc.data_ptr(), g0.view(), 1, None)
    assert abs(float(acc) - float(val)) < 1e-5 * float(val)
    assert rel(g0.nchw().cpu(), zz.grad[:n]) < 1e-5


def _product_and_oracle(dev, golden_dir):
    from dasr_amd.lpips import LPIPSAlexHIP
    from oracle import lpips
    from oracle.gen_golden_lpips import SEED
    gold = np.load(os.path.join(golden_dir, 'lpips_alex.npz'))
    lin = [torch.from_numpy(gold['lin%d' % i]) for i in range(5)]
    feats = lpips.alexnet_init_(lpips.alexnet_features(), SEED)
    crit = lpips.PerceptualLossLPIPS(lpips.LPIPSAlex(feats, lin))
    sd = {'features.' + k: v for k, v in feats.state_dict().items()}
    sd.update({'lin%d.model.1.weight' % i: w.reshape(1, -1, 1, 1) for i, w in enumerate(lin)})
    net = LPIPSAlexHIP(device=dev)
    net.load_state_dict(sd)
    return net, crit, gold


def _run(net, x, y, dev, weight=1.0):
    from dasr_amd.engine import BTensor, OpList
    n, _, H, W = x.shape
    p = net.plan(2 * n, n, H, W)
    acc = torch.zeros(1, device=dev)
    ops = OpList()
    xb, yb = to_blocked(x, dev), to_blocked(y, dev)
    ops.add(p.input_op(xb.view(), 0, n))
    ops.add(p.input_op(yb.view(), n, n))
    ops.extend(p.fwd)
    for o in p.head_ops(acc.data_ptr(), weight):
        ops.add(o)
    ops.extend(p.bwd)
    gimg = BTensor(n, 16, H, W, True, dev)
    ops.add(p.adjoint_op(gimg.view()))
    ops.keep += [xb, yb, gimg, acc]
    ops.run()
    torch.cuda.synchronize()
    return float(acc), gimg.nchw(3).cpu(), p


@pytest.mark.parametrize('case', ['a', 'b'])
def test_lpips_loss_and_image_gradient_match_oracle_and_reference_fixture(case, golden_dir, margins):
    dev = _gpu()
    from oracle import fixtures
    from oracle.gen_golden_lpips import CASES, lpips_batch
    net, crit, gold = _product_and_oracle(dev, golden_dir)
    x, y = lpips_batch(CASES[case])
    xr = x.clone().requires_grad_(True)
    l = crit(xr, y)
    gx, = torch.autograd.grad(l, xr)
    loss, g, p = _run(net, x, y, dev)
    feats = crit.net.slices(2 * torch.cat([x, y]) - 1)
    worst_f = max(rel(r.nchw().cpu(), f.detach()) for r, f in zip(p.relu, feats))
    e_l, e_ref, e_g = abs(loss - float(l)) / float(l), abs(loss - float(gold[case + '_loss'][0])) / float(l), rel(g, gx)
    margins('LPIPS case %s: relu1..5 worst rel err %.2e (tol 1e-3); loss vs oracle %.2e, vs reference fixture %.2e (tol 1e-3); dL/dimage rel err %.2e (tol 1e-2)'
            % (case, worst_f, e_l, e_ref, e_g))
    assert worst_f < ACT_TOL and e_l < ACT_TOL and e_ref < ACT_TOL and e_g < GRAD_TOL
    np.testing.assert_allclose(float(g.double().norm()), gold[case + '_gx_norm'][0], rtol=GRAD_TOL)
    sub, gsub = fixtures.subsample(g).numpy(), gold[case + '_gx_sub']
    assert np.linalg.norm(sub - gsub) <= GRAD_TOL * np.linalg.norm(gsub)


def test_lpips_at_training_size_is_zero_for_identical_pairs_and_deterministic(golden_dir):
    """the configs[2] shape of the loss: 512 x 512 HR crops; identical inputs -> (numerically) zero loss and gradient; run-to-run bit-exact"""
    dev = _gpu()
    net, crit, gold = _product_and_oracle(dev, golden_dir)
    g = torch.Generator().manual_seed(4)
    y = torch.rand(2, 3, 512, 512, generator=g)
    loss, gimg, _ = _run(net, y, y, dev)
    assert loss < 1e-12 and float(gimg.abs().max()) < 1e-9     # not bit-zero: the two images of a pair sit in different tiles of a launch
    x = (y + 0.1 * (torch.rand(y.shape, generator=g) - 0.5)).clamp(0, 1)
    l1, g1, _ = _run(net, x, y, dev, 0.5)
    l2, g2, _ = _run(net, x, y, dev, 0.5)
    assert l1 > 0 and torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(g1, g2) and abs(l1 - l2) <= 1e-6 * l1     # the loss sum uses atomics; the gradient is order-free


def test_val_lpips_metric_in_sr_model_matches_oracle(golden_dir):
    """`val_lpips: true` (SR_model.py:95-99): LPIPS of the 8-bit SR and HR images after test(), reported through get_current_visuals()"""
    dev = _gpu()
    from oracle import fixtures, lpips
    from dasr_amd import options
    from dasr_amd.models import create_model
    opt = fixtures.make_opt(dict(kind='sr', nf=32, nb=1, n=1, lr=16))
    opt['gpu_ids'], opt['val_lpips'], opt['is_train'] = [0], True, False
    m = create_model(options.dict_to_nonedict(opt))
    m.netG.load_state_dict(fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1))
    crit, sd = lpips.golden_criterion(77, golden_dir)
    from dasr_amd.lpips import LPIPSAlexHIP
    m.cri_fea_lpips = LPIPSAlexHIP(device=dev)
    m.cri_fea_lpips.load_state_dict(sd)
    g = torch.Generator().manual_seed(8)
    data = {'LR': torch.rand(1, 3, 16, 20, generator=g), 'HR': torch.rand(1, 3, 64, 80, generator=g)}
    m.feed_data(data)
    m.test()
    vis = m.get_current_visuals()
    q = lambda t: (t.float().clamp(0, 1) * 255.0).round() / 255.0
    want = float(crit(q(vis['SR'][None]), q(data['HR'])))
    assert abs(float(vis['LPIPS']) - want) < 1e-3 * want, (float(vis['LPIPS']), want)
