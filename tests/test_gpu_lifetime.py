"""No op of a recorded plan may point at freed memory.  The ops hold raw device pointers; the tensors behind them must stay referenced by the
plan (round 3: the VGG plans kept the intermediate gradient tensors of their backward chain only as locals of the builder -- freed on return, the
memory stayed intact just as long as the caching allocator did not hand it out again: a rare non-finite-gradient failure of the DSN fixtures).
The check: every device pointer found in the ops of a trainer's plans lies inside a CUDA tensor that is still alive after a garbage collection."""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu


def _live_intervals():
    gc.collect()
    iv = []
    for o in gc.get_objects():
        try:
            if torch.is_tensor(o) and o.is_cuda:
                st = o.untyped_storage()
                iv.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
        except Exception:
            pass
    iv.sort()
    return iv


def _inside(iv, p):
    import bisect
    i = bisect.bisect_right(iv, (p, float('inf'))) - 1
    return i >= 0 and iv[i][0] <= p < iv[i][1]


def _op_lists():
    from dasr_amd.engine import OpList
    gc.collect()
    return [o for o in gc.get_objects() if isinstance(o, OpList)]


def _check_all_plans(tag):
    from dasr_amd import _lib
    iv = _live_intervals()
    n_ptr, bad = 0, []
    for ol in _op_lists():
        for k, o in enumerate(ol.ops):
            ptrs = [o.t[j].p for j in range(5)]
            if o.op == _lib.OP_CONV:
                c = o.conv
                ptrs += [c.inp.p, c.mask.p, c.res1.p, c.res2.p, c.out_f32.p, c.out_bf16.p, c.w, c.bias, c.slope_ptr]
            if o.op in (_lib.OP_CONV_CHAIN, _lib.OP_RDB_CHAIN):   # p[1] is the HOST copy of the layer table (validated by the launcher): kept alive by a ConvChain in ol.keep
                import ctypes as C
                hosts = [C.cast(ch.host, C.c_void_p).value for ch in ol.keep if hasattr(ch, 'host')]
                assert o.p[1] in hosts, (tag, k, 'host layer table of a chained launch is not kept by its list')
                ptrs += [o.p[0], o.p[2], o.p[3], o.l[0]]
            elif o.op not in (_lib.OP_EVENT_RECORD, _lib.OP_STREAM_WAIT, _lib.OP_SET_STREAM):   # (those carry host-side event / stream handles)
                ptrs += [o.p[j] for j in range(4)]
            for p in ptrs:
                if p:
                    n_ptr += 1
                    if not _inside(iv, int(p)):
                        bad.append((k, o.op, hex(int(p))))
        for grp in ol.keep:   # weight-gradient groups: the part descriptors (copied into a device table) point at gradient / input tensors
            for entry in getattr(grp, 'parts', []):
                wp = entry[0]
                for p in (wp.g.p, wp.inp.p):
                    if p:
                        n_ptr += 1
                        if not _inside(iv, int(p)):
                            bad.append(('wgrad part', type(grp).__name__, hex(int(p))))
    assert n_ptr > 50, (tag, n_ptr)
    assert not bad, (tag, len(bad), bad[:8])
    return n_ptr


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda')


def test_dasr_trainer_plans_keep_their_tensors(margins):
    """the full GAN trainer with the VGG19-54 criterion (generator, patch discriminator, perceptual network plans)"""
    _gpu()
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = 'dasr_wavelet_nf32_nb2_n2_32'
    opt = fixtures.make_opt(case)
    opt['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(opt))
    m.feed_data(fixtures.make_batch(case), True)
    m.optimize_parameters(1)
    torch.cuda.synchronize()
    n = _check_all_plans('DASR')
    margins('plan lifetime check (DASR trainer, VGG criterion): %d device pointers in the recorded ops, all inside live tensors' % n)


def test_sr_trainer_plans_keep_their_tensors():
    """the L1 trainer (two sub-batch streams, deferred grouped weight gradients, TrunkStore views) and its inference plan"""
    _gpu()
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.models import create_model
    case = 'sr_nf64_nb2_b8_32'
    opt = fixtures.make_opt(case)
    opt['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(opt))
    m.feed_data(fixtures.make_batch(case))
    m.optimize_parameters(1)
    m.test()
    torch.cuda.synchronize()
    _check_all_plans('SR')


@pytest.mark.parametrize('per', ['VGG', 'LPIPS'])
def test_dsn_model_plans_keep_their_tensors(per):
    _gpu()
    from dasr_amd.dsn_model import DSNModel
    from oracle.gen_golden_dsn import dsn_batch
    m = DSNModel(dict(filter='wavelet', w_per=0.01, vgg_seed=78, per_type=per, allow_random_perceptual=True), device='cuda')
    hr, bic, real = dsn_batch(dict(n=1, crop=160))
    m.iteration(hr.cuda(), bic.cuda(), real.cuda())
    torch.cuda.synchronize()
    _check_all_plans('DSN ' + per)


@pytest.mark.parametrize('prec', [5, 4, 2])
def test_vgg_plan_keeps_its_tensors(prec):
    dev = _gpu()
    from dasr_amd.gan_nets import VGGFeatureHIP
    V = VGGFeatureHIP(34, device=dev, prec=prec)
    p = V.plan(2, 1, 40, 40)
    junk = [torch.zeros(1 << 20, device=dev) for _ in range(64)]   # would re-use freed blocks
    del junk
    _check_all_plans('VGG prec %d' % prec)
    assert p is not None
