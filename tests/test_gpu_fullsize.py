"""Full-size (BASELINE.json configs[1] shapes: 16 x 64..192 ch x 128 x 128, nf 64) checks through size-independent
properties, where the CPU oracle would take minutes:
  * adjointness:   <conv_W(x), g> == <x, dgrad_W(g)>  (forward kernel vs the same kernel on transposed/flipped packs)
  * wgrad pairing: <conv_W'(x), g> == <W', wgrad(x, g)> for a random direction W' (the op is linear in W)
  * linearity:     conv(a x) == a conv(x) (no bias / activation)
  * determinism:   two identical training steps give bit-identical gradients (fixed-order split reductions)
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    return torch.device('cuda')


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize('cin,cout,prec', [(160, 32, 1), (192, 64, 1), (64, 64, 3)])
def test_conv_adjoint_wgrad_linearity_fullsize(cin, cout, prec):
    dev = _gpu()
    from dasr_amd.engine import BTensor, ParamStore, PackRegistry, OpList, WgradGroup, Workspace, conv_op
    N, H, W = 16, 128, 128
    f32 = prec == 3
    dt = torch.float32 if f32 else torch.bfloat16
    g = torch.Generator(device='cuda').manual_seed(1)
    P = ParamStore([('w', (cout, cin, 3, 3)), ('w2', (cout, cin, 3, 3)), ('b', (cout,))], dev)
    P.view('w').copy_(torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.05)
    P.view('w2').copy_(torch.randn(cout, cin, 3, 3, device=dev, generator=g) * 0.05)
    if not f32:  # make the weights exactly representable so that prec-1 rounding does not enter the identities
        P.flat.copy_(P.flat.to(torch.bfloat16).float())
    pack = PackRegistry(P)
    mt = 2 if (cout == 64 and prec == 1) else 1
    fw = pack.add(cout, cin, 9, mt, prec, [(P.off('w'), cout, cin, 0, cin, 0, 0)])
    fw2 = pack.add(cout, cin, 9, mt, prec, [(P.off('w2'), cout, cin, 0, cin, 0, 0)])
    bw = pack.add(cin, cout, 9, 2 if (cin % 64 == 0 and prec == 1) else 1, prec, [(P.off('w'), cout, cin, 0, cout, 0, 1)])
    pack.finalize()
    pack.run()
    x, gy = BTensor(N, cin, H, W, f32, dev), BTensor(N, cout, H, W, f32, dev)
    x.t.copy_(torch.randn(x.t.shape, device=dev, generator=g).to(dt))
    gy.t.copy_(torch.randn(gy.t.shape, device=dev, generator=g).to(dt))
    y, y2, ya, gx = (BTensor(N, cout, H, W, True, dev) for _ in range(3)), None, None, None
    y, y2, ya = y
    gx = BTensor(N, cin, H, W, True, dev)
    ops = OpList()
    ops.add(conv_op(pack, fw, x.view(), f32, cin, H, W, H, W, N, out_f32=y.view()))
    ops.add(conv_op(pack, fw2, x.view(), f32, cin, H, W, H, W, N, out_f32=y2.view()))
    ops.add(conv_op(pack, bw, gy.view(), f32, cout, H, W, H, W, N, out_f32=gx.view()))
    ws = Workspace(dev)
    grp = WgradGroup(3, 1)
    grp.add_conv(gy.view, f32, gy.planes, x.view, f32, x.planes, cout, cin, H, W, H, W, N, P.off('w'), P.off('b'))
    grp.finalize(ws, dev)
    for o in grp.ops(P.grad.data_ptr()):
        ops.add(o)
    ws.finalize()
    ops.run()
    torch.cuda.synchronize()
    lhs, rhs = _dot(y.t, gy.t), _dot(x.t, gx.t)
    assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), 1.0) + 1e-3 * float(y.t.double().norm() * gy.t.double().norm()) * 1e-3, (lhs, rhs)
    # <conv_{w2}(x), g> == <w2, dW>: bf16 operands are exact here (x, g bf16 for prec 1; f32 inputs are rounded by wgrad)
    lhs2, rhs2 = _dot(y2.t, gy.t), _dot(P.view('w2'), P.view('w', P.grad))
    tol = 5e-3 if f32 else 2e-4  # f32 path: wgrad rounds x and g to bf16 (documented), the forward does not
    scale = float(y2.t.double().norm() * gy.t.double().norm())
    assert abs(lhs2 - rhs2) <= tol * scale, (lhs2, rhs2, scale)
    # bias gradient == sum of g
    assert torch.allclose(P.view('b', P.grad), gy.nchw().sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-2)
    # linearity in x
    x.t.mul_(2)
    ops2 = OpList()
    ops2.add(conv_op(pack, fw, x.view(), f32, cin, H, W, H, W, N, out_f32=ya.view()))
    ops2.run()
    torch.cuda.synchronize()
    assert float((ya.t - 2 * y.t).abs().max()) <= 1e-5 * float(y.t.abs().max())


def test_training_step_is_deterministic():
    dev = _gpu()
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    outs = []
    for _ in range(2):
        opt = fixtures.make_opt('sr_nf64_nb2_b2_32')
        opt['gpu_ids'] = [0]
        m = create_model(options.dict_to_nonedict(opt))
        sd = fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1)
        m.netG.load_state_dict(sd)
        g = torch.Generator().manual_seed(3)
        data = {'LR': torch.rand(8, 3, 48, 48, generator=g), 'HR': torch.rand(8, 3, 192, 192, generator=g)}  # 2 sub-batch streams
        m.update_learning_rate()
        m.feed_data(data)
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        outs.append((m.netG.params.grad.clone(), m.netG.params.flat.clone(), m.fake_H.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
