"""GPU parity of the generator and of the SRModel training step against the oracle (fp32 CPU restatement,
itself pinned to the reference by tests/golden) and against the committed reference fixtures.

Tolerances are the north_star's: generator activations within 1e-3 relative (normwise per tap), gradients
within 1e-2 relative (normwise per parameter tensor)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT_TOL = 1e-3
GRAD_TOL = 1e-2


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda')


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


TAP_RRDBS = (0, 11, 22)   # RRDB outputs compared at full depth (nb = 23)


def _oracle_run(case, steps=2):
    from oracle import fixtures, nets, trainers
    c = fixtures.CASES[case] if isinstance(case, str) else case
    opt = fixtures.make_opt(case)
    netG = nets.RRDBNet(3, 3, c['nf'], c['nb'], 4, upsample_mode=c.get('upsample_mode', 'upconv'))
    sd0 = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
    netG.load_state_dict(sd0)
    t = trainers.SRTrainer(opt, netG=netG)
    batch = fixtures.make_batch(case)
    taps = {}
    def hook(name):
        def f(m, i, o):  # must return None (a returned tensor would replace the module output)
            taps.setdefault(name, o.detach().clone())
        return f

    hs = [netG.model[0].register_forward_hook(hook('fea')), netG.model[1].register_forward_hook(hook('trunk'))]
    hs += [netG.model[1].sub[i].register_forward_hook(hook('rrdb%d' % i)) for i in TAP_RRDBS if i < c['nb']]
    out = {'sd0': sd0, 'batch': batch, 'logs': []}
    for step in range(1, steps + 1):
        t.update_learning_rate()
        t.feed_data(batch)
        t.optimize_parameters(step)
        out['logs'].append(t.log['l_pix'])
        if step == 1:
            for h in hs:
                h.remove()
            out['taps'] = taps
            out['sr'] = t.fake_H.detach().clone()
            out['grads'] = [p.grad.detach().clone() for p in netG.parameters()]
    out['sdN'] = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    return out


# sr_nf64_nb2_b8_32: batch 8 -> the PRODUCTION schedule (two sub-batch replicas on two streams, run_interleaved, private replica
# gradient buffer + add_flat; SR_model.py:77-85 is the reference step).  sr_nf64_nb23_b2_32: the full ESRGAN depth
# (architecture.py:174-205), RRDB outputs 0 / 11 / 22 tapped in fp32.
# sr_ps_nf64_nb1_b2_32: the PixelShuffle upsampler (block.py:838-851) instead of nearest + conv.
# sr_l2_nf32_nb1_b2_32: pixel_criterion 'l2' (nn.MSELoss, SR_model.py:33-36).
# FULL_128 (round 3): the bench's depth AND spatial size -- nf64 / nb23 at 128 x 128 LR, where the size-dependent
# kernel paths switch on (XCD tile remap at >= 512 workgroups is not reached at batch 2, but 128-wide rows, 8 x 4 tiles per image, the 16-way
# pixel splits of the weight gradients and the deferred weight-gradient phase over 128^2 are) -- against the oracle, all 702 gradients.
FULL_128 = dict(kind='sr', nf=64, nb=23, n=2, lr=128)


@pytest.mark.parametrize('case', ['sr_nf64_nb1_b1_24x40', 'sr_nf64_nb2_b2_32', 'cfg1_sr_nf32_nb4_b2_64', 'sr_nf64_nb2_b8_32', 'sr_nf64_nb23_b2_32',
                                  'sr_ps_nf64_nb1_b2_32', 'sr_l2_nf32_nb1_b2_32', FULL_128],
                         ids=lambda c: c if isinstance(c, str) else 'sr_nf64_nb23_b2_128')
def test_sr_step_matches_oracle_and_reference_fixture(case, golden_dir, margins):
    dev = _gpu()
    torch.set_num_threads(8)
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    cfg = fixtures.CASES[case] if isinstance(case, str) else case
    nsteps = 2 if isinstance(case, str) else 1   # the 128^2 case has no reference fixture (the reference would need minutes): oracle only, one step
    want = _oracle_run(case, nsteps)
    opt = fixtures.make_opt(case)
    opt['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(opt))
    m.netG.load_state_dict(want['sd0'])
    m.netG.debug_taps = tuple(i for i in TAP_RRDBS if i < cfg['nb'])
    gold = np.load(os.path.join(golden_dir, case + '.npz')) if isinstance(case, str) else None
    case = case if isinstance(case, str) else 'sr_nf64_nb23_b2_128'
    logs = []
    for step in range(1, nsteps + 1):
        m.update_learning_rate()
        m.feed_data(want['batch'])
        m.optimize_parameters(step)
        logs.append(m.get_current_log()['l_pix'])
        if step == 1:
            plans = m._out_plans
            assert len(plans) == (2 if cfg['n'] >= 8 else 1)   # batch >= 8: the two-stream schedule really ran
            cat = lambda f: torch.cat([f(p).cpu() for p in plans], 0)
            # activations
            errs = {'fea': rel(cat(lambda p: p.fea.nchw()), want['taps']['fea']), 'trunk': rel(cat(lambda p: p.t0.nchw()), want['taps']['trunk']),
                    'sr': rel(m.fake_H.cpu(), want['sr'])}
            for i in m.netG.debug_taps:
                errs['rrdb%d' % i] = rel(cat(lambda p: p.taps[i].nchw()), want['taps']['rrdb%d' % i])
            margins('%s activations: %s (tol %.0e)' % (case, ' '.join('%s %.2e' % kv for kv in errs.items()), ACT_TOL))
            for k, e in errs.items():
                assert e < ACT_TOL, (k, e)
            # gradients, per parameter tensor
            gd = m.netG.params.grad_dict()
            worst, wk = 0.0, None
            for (k, gv), gw in zip(gd.items(), want['grads']):
                r = rel(gv, gw)
                if r > worst:
                    worst, wk = r, k
            margins('%s gradients: worst normwise rel err %.2e at %s (tol %.0e, %d tensors)' % (case, worst, wk, GRAD_TOL, len(gd)))
            assert worst < GRAD_TOL, (wk, worst)
            # against the REFERENCE's own numbers
            if gold is not None:
                np.testing.assert_allclose(np.array([float(g.double().norm()) for g in gd.values()]), gold['gradG_norm'], rtol=GRAD_TOL)
                for i in m.netG.debug_taps:
                    got_n = float(cat(lambda p: p.taps[i].nchw()).double().norm())
                    np.testing.assert_allclose(got_n, float(gold['tap_norm/trunk_%d' % i]), rtol=ACT_TOL)
    np.testing.assert_allclose(logs, want['logs'], rtol=1e-4)
    if gold is not None:
        np.testing.assert_allclose(logs, gold['logs'][:, 0], rtol=1e-4)
    # weights after 2 Adam steps: Adam normalises the update (a sign flip of a ~0 gradient moves a weight by 2*lr), so the bound is
    # absolute; the worst observed values are logged so the margin stays visible
    sdN = m.netG.state_dict()
    dmax, frac = 0.0, 0.0
    for k, v in sdN.items():
        d = (v - want['sdN'][k]).abs()
        dmax, frac = max(dmax, float(d.max())), max(frac, float((d > 2e-5).float().mean()))
        assert float(d.max()) <= 3.2e-4, k
        assert float((d > 2e-5).float().mean()) < 0.02, (k, float((d > 2e-5).float().mean()))
    margins('%s weights after %d Adam steps: max |dw| %.2e (bound 3.2e-4 at lr 1e-4), worst fraction of elements off by > 2e-5: %.4f (bound 0.02)'
            % (case, nsteps, dmax, frac))


def test_checkpoint_layout_roundtrip(tmp_path):
    dev = _gpu()
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = 'sr_nf64_nb1_b1_24x40'
    opt = fixtures.make_opt(case)
    opt['gpu_ids'] = [0]
    opt['path']['models'] = str(tmp_path)
    opt['path']['training_state'] = str(tmp_path)
    m = create_model(options.dict_to_nonedict(opt))
    batch = fixtures.make_batch(case)
    m.update_learning_rate()
    m.feed_data(batch)
    m.optimize_parameters(1)
    m.save(1)
    m.save_training_state(0, 1)
    sd = torch.load(os.path.join(str(tmp_path), '1_G.pth'))
    from oracle import nets
    ref_net = nets.RRDBNet(3, 3, 64, 1, 4)
    ref_net.load_state_dict(sd)  # strict: same keys and shapes as the reference module tree
    st = torch.load(os.path.join(str(tmp_path), '1.state'), weights_only=False)
    assert set(st) == {'epoch', 'iter', 'schedulers', 'optimizers'} and st['iter'] == 1
    # the optimizer entry loads into a real torch Adam over the reference-shaped parameters
    o = torch.optim.Adam(ref_net.parameters(), lr=1e-4)
    o.load_state_dict(st['optimizers'][0])
    m2 = create_model(options.dict_to_nonedict(opt))
    m2.netG.load_state_dict(sd)
    m2.resume_training(st)
    assert m2.optimizers[0].step_count == 1 and m2.schedulers[0].last_epoch == 1
    for a, b in ((m, m2),):
        a.update_learning_rate(); b.update_learning_rate()
        a.feed_data(batch); b.feed_data(batch)
        a.optimize_parameters(2); b.optimize_parameters(2)
    for k, v in m.netG.state_dict().items():
        assert torch.equal(v, m2.netG.state_dict()[k]), k


def test_sr_step_with_split_bf16_hr_tail(golden_dir, margins, monkeypatch):
    """DASR_HR_PREC=3: the HR tail on f32 tensors in split-bf16 with sub-pixel upconvs -- the fall-back the non-finite-gradient guard points to when
    f16's range is not enough (models.py::AdamHIP.check_finite); same fixture, same tolerances as the default f16-storage tail"""
    monkeypatch.setenv('DASR_HR_PREC', '3')
    test_sr_step_matches_oracle_and_reference_fixture('sr_nf64_nb2_b2_32', golden_dir, margins)


@pytest.mark.parametrize('case', ['sr_nf64_nb2_b8_32', 'sr_nf64_nb23_b2_32', 'cfg1_sr_nf32_nb4_b2_64'])
def test_sr_step_with_f16_dense_blocks(case, golden_dir, margins, monkeypatch):
    """DASR_RDB_PREC=2 (round 4): dense slabs and their gradients in f16 storage (11-bit operands, gradients pre-scaled by a power of two) instead of
    bf16 -- the numerics switch DASR_Model selects for the BatchNorm source discriminator.  Same fixtures and tolerances as the bf16 default: the
    two-stream production schedule (batch 8), the full depth nb = 23, and nf = 32 (conv5 on the Cout-32 path)"""
    monkeypatch.setenv('DASR_RDB_PREC', '2')
    test_sr_step_matches_oracle_and_reference_fixture(case, golden_dir, margins)


def test_f16_dense_block_gradient_scale_is_consistently_patched(monkeypatch):
    """DASR_RDB_PREC=2: the dense-block gradient slabs hold gscale * dL/d(.), gscale a power of two calibrated from max |dL/d(trunk output)| and patched
    into every op that carries it (rrdbnet.TrunkStore.set_gscale_from).  The gradients must not depend on the scale beyond f16 rounding: the same step with the
    calibration forced 2^5 off (both directions) gives the same weight gradients; the two-replica schedule (batch 8) calibrates on a dry run of replica 0"""
    dev = _gpu()
    monkeypatch.setenv('DASR_RDB_PREC', '2')
    from oracle import fixtures
    from dasr_amd import options, rrdbnet
    from dasr_amd.models import create_model
    case = 'sr_nf64_nb2_b8_32'
    grads, scales = [], []
    orig = rrdbnet.TrunkStore.set_gscale_from
    for factor in (1.0, 32.0, 1.0 / 32.0):
        monkeypatch.setattr(rrdbnet.TrunkStore, 'set_gscale_from', lambda self, a, f=factor: orig(self, a * f))
        opt = fixtures.make_opt(case)
        opt['gpu_ids'] = [0]
        m = create_model(options.dict_to_nonedict(opt))
        m.netG.load_state_dict(fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1))
        m.update_learning_rate()
        m.feed_data(fixtures.make_batch(case))
        m.optimize_parameters(1)
        torch.cuda.synchronize()
        assert len(m._out_plans) == 2 and m._out_plans[0].store is m._out_plans[1].store
        scales.append(m._out_plans[0].store.gscale)
        grads.append(m.netG.params.grad_dict())
        m.get_current_log()   # (raises if a gradient overflowed)
    assert scales[1] == scales[0] / 32.0 and scales[2] == scales[0] * 32.0 and scales[0] > 1.0
    for other in grads[1:]:
        worst = max(rel(other[k], v) for k, v in grads[0].items())
        assert worst < 2e-3, worst


@pytest.mark.parametrize('shape', [(16, 128, 128, 2, 'layer'), (8, 256, 128, 1, 'layer'), (32, 128, 128, 1, 'layer'), (24, 128, 256, 1, 'layer'),
                                   (16, 128, 128, 2, 'is'), (8, 128, 128, 2, 'is'), (32, 128, 128, 1, 'is'), (24, 128, 128, 1, 'is'), (64, 64, 64, 1, 'is'), (8, 128, 112, 3, 'is'), (16, 32, 32, 2, 'is'), (16, 64, 64, 1, 'is')],
                         ids=['16x128x128', '8x256x128', '32x128x128-two_sub_batches', '24x128x256-three_sub_batches',
                              'is-16x128x128-two_tiles_per_workgroup', 'is-8x128x128-one_tile', 'is-32x128x128-four_tiles', 'is-24x128x128-three_tiles', 'is-64x64x64-four_images_per_xcd',
                              'is-8x128x112-partial_tiles-rrdb_residual', 'is-16x32x32-shipped_shape-32_workgroups', 'is-16x64x64-128_workgroups'])
def test_chained_trunk_launches_are_bit_identical_to_per_layer_launches(shape, monkeypatch):
    """DASR_CHAIN (default on where the batch fills the chip exactly, RRDBNetHIP.chain_ok): the 15 nb dense-block convs of the forward and of the data
    gradient each run as ONE persistent launch in which a tile waits for its neighbour tiles only before the input chunks the previous layer wrote
    (dasr_conv_chain).  Same arithmetic in the same order: SR output, every gradient and the weights after two Adam steps must be BIT-identical to the
    per-layer launches, and the device error word stays zero (no neighbour wait gave up).  Round 5: batches of k x 512 tiles run k chained launches back to
    back over image ranges (RRDBNetHIP.chain_split: configs[2]'s 32 crops = two launches of 16).  Round 6: the INPUT-STATIONARY form (dasr_rdb_chain, csrc/rdb_is.h: every slab
    chunk staged once per dense block, every accumulator receives its products in the order of the per-layer kernel) -- one launch over all 15 nb convs, one to eight tiles per
    workgroup, one to four images per XCD and slot, partial tiles, the RRDB residual (nb 3: three dense blocks with the second fp32 residual)."""
    _gpu()
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip('the chained launches need a whole 256-CU MI355X (RRDBNetHIP.chain_ok)')
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    n, h, w, nb, form = shape
    monkeypatch.setenv('DASR_CHAIN_FORM', form)
    case = dict(kind='sr', nf=64, nb=nb, n=n, lr=(h, w))
    batch = fixtures.make_batch(case, seed=11)
    outs = []
    monkeypatch.setenv('DASR_STREAMS', '1')   # like for like: one plan over the whole batch on both sides (the sub-batch schedule sums the weight gradients in another order)
    for chain in ('0', '1'):
        monkeypatch.setenv('DASR_CHAIN', chain)
        torch.manual_seed(0)
        o = fixtures.make_opt(case)
        o['gpu_ids'] = [0]
        m = create_model(options.dict_to_nonedict(o))
        assert m.netG.chain_ok(n, h, w) == (chain == '1')
        for step in (1, 2):
            m.update_learning_rate()
            m.feed_data(batch)
            m.optimize_parameters(step)
        m.check_finite()                         # includes the chains' error word
        plans = m._out_plans
        if chain == '1':
            assert len(plans) == 1 and plans[0].chain is not None and plans[0].chain_b is not None and plans[0].chain.form == plans[0].chain_b.form == form
            if form == 'is':
                assert plans[0].chain.n == 15 * nb and plans[0].chain_b.n == 15 * nb and len(plans[0].chains) == len(plans[0].chains_b) == m.netG.chain_split(n, h, w) == 1
            else:
                assert plans[0].chain.n == 15 * nb - 1 and plans[0].chain_b.n == 15 * nb - 1
                assert len(plans[0].chains) == len(plans[0].chains_b) == m.netG.chain_split(n, h, w) == n * ((h + 15) // 16) * ((w + 31) // 32) // 512
        else:
            assert all(getattr(p, 'chain', None) is None for p in plans)
        outs.append((m.fake_H.clone(), m.netG.params.grad.clone(), m.netG.params.flat.clone()))
    (s0, g0, w0), (s1, g1, w1) = outs
    assert torch.equal(s0, s1) and torch.equal(g0, g1) and torch.equal(w0, w1)


def test_chain_refuses_shapes_that_do_not_fill_the_chip():
    _gpu()
    from dasr_amd.rrdbnet import RRDBNetHIP
    net = RRDBNetHIP(3, 3, 64, 1, device='cuda')
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        assert not net.chain_ok(16, 128, 128)   # a partitioned device: never
        return
    assert net.chain_ok(16, 128, 128) and net.chain_ok(8, 128, 256) and net.chain_ok(32, 64, 128)
    assert not net.chain_ok(4, 256, 256) and not net.chain_ok(12, 128, 128) and not net.chain_ok(16, 192, 192)
    assert net.chain_choice(24, 128, 128)[:2] == ('is', 1) and net.chain_choice(8, 128, 128)[:2] == ('is', 1) and net.chain_choice(16, 128, 128)[:2] == ('layer', 1)
    assert (net.chain_split(16, 128, 128), net.chain_split(32, 128, 128), net.chain_split(48, 128, 128), net.chain_split(20, 128, 128)) == (1, 2, 3, 0)   # (20: 640 tiles)
    assert net.chain_split(80, 128, 128) == 0   # five sub-batches: over the limit of four (DASR_CHAIN_SPLIT)
