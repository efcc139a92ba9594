"""Generate tests/golden/*.npz by running the REFERENCE code (imported from /root/reference).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Run in the build container only:
    python -m oracle.gen_golden
The reference never travels; only these small vectors (inputs are re-derivable from
seeds, outputs are digests / sub-samples) are committed.
"""
import os
import sys

import numpy as np
import torch

from . import fixtures, nets, ref_import

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def collect(case, netG, netD, update_lr, feed, step_fn, get_log, steps=2, netD2=None):
    """Drive a trainer for `steps` steps and return the fixture dict (numpy)."""
    out = {}
    taps = {}

    def hook(name):
        def f(m, i, o):
            if name not in taps:
                taps[name] = o.detach().clone()
        return f

    hs = [netG.model[0].register_forward_hook(hook('fea_conv'))]
    for i, blk in enumerate(netG.model[1].sub):
        hs.append(blk.register_forward_hook(hook('trunk_%d' % i)))
    hs.append(netG.model[1].register_forward_hook(hook('trunk_out')))
    hs.append(netG.register_forward_hook(hook('sr_out')))
    out['w0_digest'] = np.array([nets.tensor_digest(v) for v in netG.state_dict().values()])
    batch = fixtures.make_batch(case)
    logs = []
    for step in range(1, steps + 1):
        update_lr()
        feed(batch)
        step_fn(step)
        logs.append(dict(get_log()))
        if step == 1:
            for h in hs:
                h.remove()
            for k, v in taps.items():
                out['tap_norm/' + k] = np.array(float(v.double().norm()))
                out['tap_sub/' + k] = fixtures.subsample(v).numpy()
            out['gradG_norm'] = np.array([float(p.grad.double().norm()) for p in netG.parameters()])
            out['gradG_sub'] = np.concatenate([fixtures.subsample(p.grad, 4).numpy() for p in netG.parameters()])
            if netD is not None:
                out['gradD_norm'] = np.array([float(p.grad.double().norm()) for p in netD.parameters()])
            if netD2 is not None:
                out['gradD2_norm'] = np.array([float(p.grad.double().norm()) for p in netD2.parameters()])
    keys = sorted(logs[0].keys())
    out['log_keys'] = np.array(keys)
    out['logs'] = np.array([[l[k] for k in keys] for l in logs], dtype=np.float64)
    out['wN_digest'] = np.array([nets.tensor_digest(v) for v in netG.state_dict().values()])
    if netD is not None:
        out['dN_digest'] = np.array([nets.tensor_digest(v) for v in netD.state_dict().values()])
    if netD2 is not None:
        out['d2N_digest'] = np.array([nets.tensor_digest(v) for v in netD2.state_dict().values()])
    out['state_keys'] = np.array(list(netG.state_dict().keys()))
    return out


def run_reference(case):
    option, SRModel, DASR_Model, arch, networks = ref_import.import_srn()
    c = fixtures.CASES[case]
    opt = option.dict_to_nonedict(fixtures.make_opt(case))
    torch.manual_seed(0)
    orig_define_G = networks.define_G
    if c.get('upsample_mode', 'upconv') != 'upconv':
        # the reference's define_G hard-wires 'upconv'; its own RRDBNet class implements the other mode: hand THAT class to its trainer
        def define_G(o):
            g = o['network_G']
            return arch.RRDBNet(in_nc=g['in_nc'], out_nc=g['out_nc'], nf=g['nf'], nb=g['nb'], gc=g['gc'], upscale=g['scale'], norm_type=g['norm_type'],
                                act_type='leakyrelu', mode=g['mode'], upsample_mode=c['upsample_mode'])
        networks.define_G = define_G
    try:
        m = (SRModel if c['kind'] == 'sr' else DASR_Model)(opt)
    finally:
        networks.define_G = orig_define_G
    m.netG.load_state_dict(fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1))
    netD = netD2 = None
    if c['kind'] == 'dasr':
        netD = m.netD_target
        netD.load_state_dict(fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0))
        if c.get('gan_src', 0) > 0:
            netD2 = m.netD_source
            netD2.load_state_dict(fixtures.seeded_state_dict(netD2.state_dict(), 3, 1.0))
        feed = lambda b: m.feed_data(b, True)
    else:
        feed = lambda b: m.feed_data(b)
    return collect(case, m.netG, netD, m.update_learning_rate, feed, m.optimize_parameters, m.get_current_log, netD2=netD2)


def run_oracle(case):
    from . import trainers
    c = fixtures.CASES[case]
    opt = fixtures.make_opt(case)
    netG = nets.RRDBNet(3, 3, c['nf'], c['nb'], 4, upsample_mode=c.get('upsample_mode', 'upconv'))
    netG.load_state_dict(fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1))
    if c['kind'] == 'sr':
        t = trainers.SRTrainer(opt, netG=netG)
        return collect(case, netG, None, t.update_learning_rate, t.feed_data, t.optimize_parameters, lambda: t.log)
    netD = nets.NLayerDiscriminator(c['d_in_nc'], n_layers=2)
    netD.load_state_dict(fixtures.seeded_state_dict(netD.state_dict(), 2, 1.0))
    netF = None
    if c.get('fea') == 'LPIPS':
        from . import lpips
        netF = lpips.golden_criterion(77)[0]
    netD2 = None
    if c.get('gan_src', 0) > 0:
        netD2 = nets.Discriminator_VGG_128(c['d_in_nc'], 64) if c.get('pairD') == 'discriminator_vgg_128' else nets.NLayerDiscriminator(c['d_in_nc'], 64, n_layers=2)
        netD2.load_state_dict(fixtures.seeded_state_dict(netD2.state_dict(), 3, 1.0))
    t = trainers.DASRTrainer(opt, netG=netG, netD=netD, netF=netF, vgg_seed=77, netD_source=netD2)
    return collect(case, netG, netD, t.update_learning_rate, t.feed_data, t.optimize_parameters, lambda: t.log, netD2=netD2)


def misc_reference():
    """Stand-alone module vectors from the reference: NLayerD, gaussian filters, init rule."""
    option, SRModel, DASR_Model, arch, networks = ref_import.import_srn()
    out = {}
    g = torch.Generator().manual_seed(4321)
    for nc in (3, 9):
        d = arch.NLayerDiscriminator(nc, n_layers=2)
        d.load_state_dict(fixtures.seeded_state_dict(d.state_dict(), 10 + nc, 1.0))
        x = torch.rand(2, nc, 64, 64, generator=g)
        y = d(x)
        out['nld%d_shape' % nc] = np.array(y.shape)
        out['nld%d_sub' % nc] = fixtures.subsample(y).detach().numpy()
    for k in (5, 9):
        x = torch.rand(1, 3, 40, 40, generator=g)
        out['flow_gau%d' % k] = fixtures.subsample(arch.FilterLow(kernel_size=k, gaussian=True)(x)).numpy()
        out['fhigh_gau%d' % k] = fixtures.subsample(arch.FilterHigh(kernel_size=k, gaussian=True)(x)).numpy()
        out['fhigh_avg%d' % k] = fixtures.subsample(arch.FilterHigh(kernel_size=k)(x)).numpy()
    # init rule (networks.py:30-44): same seed -> same weights as oracle.init_kaiming_ when the
    # construction order matches; digest of define_G(nf32, nb1) under torch.manual_seed(5)
    opt = option.dict_to_nonedict(fixtures.make_opt(dict(kind='sr', nf=32, nb=1, n=1, lr=8)))
    torch.manual_seed(5)
    netG = networks.define_G(opt)
    out['init_digest'] = np.array([nets.tensor_digest(v) for v in netG.state_dict().values()])
    f = networks.define_F(opt)
    x = torch.rand(1, 3, 64, 64, generator=g)
    out['vgg_shape'] = np.array(f(x).shape)
    out['vgg_sub'] = fixtures.subsample(f(x)).detach().numpy()
    return out


def main():
    if not ref_import.available():
        sys.exit('reference tree missing; fixtures can only be generated in the build container')
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = [a for a in sys.argv[1:] if a in fixtures.CASES]   # `python -m oracle.gen_golden CASE...`: just these (plus nothing else)
    for case in (only or fixtures.CASES):
        fx = run_reference(case)
        np.savez_compressed(os.path.join(OUT, case + '.npz'), **fx)
        print(case, 'logs', fx['logs'][0])
    if not only:
        np.savez_compressed(os.path.join(OUT, 'misc_modules.npz'), **misc_reference())
    print('written to', OUT)


if __name__ == '__main__':
    main()
