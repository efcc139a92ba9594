"""GPU parity tests of the individual HIP kernels against plain fp32 PyTorch-CPU math (the same ops the
oracle is built from), called through the C ABI (ctypes).  Run on the MI355X box with `-m gpu`."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd import engine
    engine.ensure_runtime_ready()
    return torch.device('cuda')


def to_blocked(x, f32, dev):
    """NCHW cpu tensor -> BTensor on device (layout plumbing only)."""
    from dasr_amd.engine import BTensor
    N, Cc, H, W = x.shape
    b = BTensor(N, Cc, H, W, f32, dev)
    xp = torch.zeros(N, b.planes * 16, H, W)
    xp[:, :Cc] = x
    b.t.copy_(xp.view(N, b.planes, 16, H, W).permute(0, 1, 3, 4, 2).to(b.t.dtype))
    return b


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def bf16r(x):
    return x.to(torch.bfloat16).float()


def make_conv(cout, cin, kh, mt, prec, dev, seed, transpose_src=None):
    from dasr_amd.engine import ParamStore, PackRegistry
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(cout, cin, kh, kh, generator=g) * math.sqrt(2.0 / (cin * kh * kh))
    b = torch.randn(cout, generator=g) * 0.1
    P = ParamStore([('w', (cout, cin, kh, kh)), ('b', (cout,))], dev)
    P.load_state_dict({'w': w, 'b': b})
    pack = PackRegistry(P)
    cin_pad = (cin + 15) // 16 * 16
    ref = pack.add(cout, cin_pad, kh * kh, mt, prec, [(0, cout, cin, 0, cin, 0, 0)])
    pack.finalize()
    pack.run()
    return w, b, P, pack, ref


CONV_CASES = [
    # name, prec, in_f32, mt, kh, stride, cin, cout, H, W, ups
    ('rdb_c32', 1, False, 1, 3, 1, 64, 32, 24, 40, 0),
    ('rdb_c64', 1, False, 2, 3, 1, 96, 64, 20, 36, 0),
    ('rdb_wide', 1, False, 1, 3, 1, 160, 32, 16, 32, 0),
    ('stream', 3, True, 1, 3, 1, 64, 64, 18, 34, 0),
    ('stream_ups', 3, True, 1, 3, 1, 32, 32, 9, 17, 1),
    ('stream_c3', 3, True, 1, 3, 1, 16, 64, 16, 32, 0),
    ('stream_out3', 3, True, 1, 3, 1, 64, 3, 20, 20, 0),
    ('vgg_like', 1, True, 2, 3, 1, 64, 128, 16, 16, 0),
    ('d_k4s2', 3, True, 1, 4, 2, 16, 64, 32, 48, 0),
    ('d_k4s1', 3, True, 1, 4, 1, 64, 32, 17, 21, 0),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward_matches_torch(case):
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, OpList, conv_op
    name, prec, in_f32, mt, kh, stride, cin, cout, H, W, ups = case
    N = 2
    real_cin = 3 if name == 'stream_c3' else cin
    w, b, P, pack, ref = make_conv(cout, real_cin, kh, mt, prec, dev, 11)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, real_cin, H, W, generator=g)
    HL, WL = (2 * H, 2 * W) if ups else (H, W)
    Ho = (HL + 2 - kh) // stride + 1
    Wo = (WL + 2 - kh) // stride + 1
    res = torch.randn(N, cout, Ho, Wo, generator=g)
    msk = torch.randn(N, cout, Ho, Wo, generator=g)
    xin = x if in_f32 else bf16r(x)
    xb = to_blocked(xin, in_f32, dev)
    rb = to_blocked(res, True, dev)
    mb = to_blocked(bf16r(msk), in_f32, dev)  # the mask tensor has the input's dtype (bf16 slabs / f32 stream)
    of = BTensor(N, cout, Ho, Wo, True, dev)
    ob = BTensor(N, cout, Ho, Wo, False, dev)
    ops = OpList()
    ops.add(conv_op(pack, ref, xb.view(), in_f32, (real_cin + 15) // 16 * 16, H, W, Ho, Wo, N, bias=P.ptr('b'), kh=kh, stride=stride,
                    pad=1, ups=ups, act=1, mask=mb.view(), mask_f32=int(in_f32), alpha=0.2, res1=rb.view(), beta1=1.0,
                    out_f32=of.view(), out_bf16=ob.view(), gamma=0.5))
    ops.run()
    torch.cuda.synchronize()
    # fp32 reference (operands rounded exactly as the kernel rounds them for prec 1)
    xr = xin if prec == 3 else bf16r(xin)
    wr = w if prec == 3 else bf16r(w)
    if ups:
        xr = F.interpolate(xr, scale_factor=2, mode='nearest')
    y = F.conv2d(xr.double(), wr.double(), b.double(), stride=stride, padding=1)
    y = F.leaky_relu(y, 0.2)
    y = torch.where(bf16r(msk).double() > 0, y, y * 0.2)
    y = (0.2 * y + res.double()).float()
    got = of.nchw().cpu()
    tol = 2e-5 if prec == 3 else 2e-5  # operands pre-rounded -> only accumulation-order differences remain
    assert rel(got, y) < tol, (name, rel(got, y))
    got_b = ob.nchw().cpu()
    assert rel(got_b, bf16r(y * 0.5)) < 5e-3
    # padded channels of the last plane must be exactly zero
    if cout % 16:
        pad_part = of.t[:, -1, :, :, cout % 16:]
        assert float(pad_part.abs().max()) == 0.0


def test_prelu_slope_partials_from_the_mask_epilogue():
    """round 6 (ABI 20): dasr_conv_params::prelu_part -- the 64-channel mask-only data-gradient conv of the LDS-DMA kernel (f16 tensors, learned slope) leaves
    slope * sum_{h <= 0} conv * h per workgroup; dasr_prelu_final turns the partials of several such convs into dL/dslope = sum conv * h / slope.  Against fp64 torch
    on an odd-sized image (partial tiles), and refused (DASR_EINVAL) by every launch that has no such epilogue"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, OpList, conv_op, ceil_div, Op
    L = _lib.lib()
    N, C_, H, W = 2, 64, 21, 37
    w, b, P, pack, ref = make_conv(C_, C_, 3, 2, 2, dev, 17)   # prec 2: f16 operands, one pass
    g = torch.Generator().manual_seed(23)
    f16r = lambda t: t.half().float()
    x, h = f16r(torch.randn(N, C_, H, W, generator=g)), f16r(torch.randn(N, C_, H, W, generator=g))
    slopes = torch.tensor([0.25, 0.1], device=dev)
    nblk = N * ceil_div(H, 4) * ceil_div(W, 16)
    part = torch.zeros(2 * nblk, device=dev)
    dst = torch.zeros(2, device=dev)
    xb, hb = BTensor(N, C_, H, W, False, dev, f16=True), BTensor(N, C_, H, W, False, dev, f16=True)
    for bt, src in ((xb, x), (hb, h)):
        t = src.reshape(N, C_ // 16, 16, H, W).permute(0, 1, 3, 4, 2).contiguous()
        bt.t.copy_(t.half())
    ob = BTensor(N, C_, H, W, False, dev, f16=True)
    ops = OpList()
    for k in range(2):   # two "blocks": the same conv under two slopes
        o = conv_op(pack, ref, xb.view(), False, C_, H, W, H, W, N, mask=hb.view(), mask_f32=0, slope_ptr=slopes.data_ptr() + 4 * k, out_bf16=ob.view(), out16_f16=1)
        o.conv.prelu_part = part.data_ptr() + 4 * k * nblk
        ops.add(o)
    sl = torch.tensor([slopes.data_ptr(), slopes.data_ptr() + 4], dtype=torch.int64, device=dev)
    ds = torch.tensor([dst.data_ptr(), dst.data_ptr() + 4], dtype=torch.int64, device=dev)
    ops.run()
    _lib.check(L.dasr_prelu_final(part.data_ptr(), nblk, nblk, 2, sl.data_ptr(), ds.data_ptr(), 0.5, None))
    torch.cuda.synchronize()
    y = F.conv2d(x.double(), f16r(w).double(), None, padding=1)
    want = 0.5 * (y * h.double())[h <= 0].sum()   # dL/da = sum_{h <= 0} dL/dh * h / a, times `scale`
    for k, a in enumerate((0.25, 0.1)):
        assert abs(float(dst[k]) - float(want) / a) < 2e-4 * abs(float(want) / a) + 1e-3, (k, float(dst[k]), float(want) / a)
    # ... and nowhere else: an f32-tensor conv (register-staged kernel), a 32-channel LDS-DMA conv
    w2, b2, P2, pack2, ref2 = make_conv(32, 64, 3, 1, 1, dev, 5)
    xb16, o16 = BTensor(N, 64, H, W, False, dev), BTensor(N, 32, H, W, False, dev)
    bad = conv_op(pack2, ref2, xb16.view(), False, 64, H, W, H, W, N, mask=o16.view(), mask_f32=0, out_bf16=o16.view())
    bad.conv.prelu_part = part.data_ptr()
    assert L.dasr_conv(C.byref(bad.conv), None) == -22
    w3, b3, P3, pack3, ref3 = make_conv(64, 64, 3, 1, 3, dev, 6)
    xf, of = BTensor(N, 64, H, W, True, dev), BTensor(N, 64, H, W, True, dev)
    bad = conv_op(pack3, ref3, xf.view(), True, 64, H, W, H, W, N, mask=xf.view(), mask_f32=1, out_f32=of.view())
    bad.conv.prelu_part = part.data_ptr()
    assert L.dasr_conv(C.byref(bad.conv), None) == -22


def test_product_library_has_one_dense_conv_family():
    """round 4: the A/B variants of rounds 1-3 (first-generation tilings, ring / loader / flag forms, dasr_set_tuning keys 1-6) live in
    libdasr_hip_ablate.so only; the product library accepts the defaults and the workgroup-shape rule of the Cout = 64 launches (key 2)"""
    _gpu()
    from dasr_amd import _lib
    L = _lib.lib()
    for key, value in ((1, 0), (1, 13), (1, 14), (1, 15), (2, 0), (2, 8), (3, 1), (4, 0), (5, 0), (6, 1)):
        assert L.dasr_set_tuning(key, value) == -22, (key, value)
    for key, value in ((1, 12), (2, 12), (2, 13), (3, 0), (4, 1), (5, 1), (6, 0)):
        assert L.dasr_set_tuning(key, value) == 0, (key, value)
    assert L.dasr_wgrad_set_mode(1 | 128) == -22 and L.dasr_wgrad_set_mode(1 | 2) == -22 and L.dasr_wgrad_set_mode(1) == 0


@pytest.mark.parametrize('case', [c for c in CONV_CASES[:3] if c[3] == 2], ids=[c[0] for c in CONV_CASES[:3] if c[3] == 2])
def test_conv64_four_wave_shape_matches_torch(case):
    """dasr_set_tuning(2, 12): Cout = 64 launches always in the 4-wave shape (default 13: 8 waves for launches of <= 256 four-wave workgroups)"""
    _gpu()
    from dasr_amd import _lib
    L = _lib.lib()
    _lib.check(L.dasr_set_tuning(2, 12))
    try:
        test_conv_forward_matches_torch(case)
    finally:
        _lib.check(L.dasr_set_tuning(2, 13))


@pytest.mark.parametrize('tune64', [12, 13], ids=['4waves', '8waves'])
@pytest.mark.parametrize('bias,res2', [(False, False), (True, False), (False, True), (True, True)], ids=['e232', 'e233', 'e248', 'e249'])
@pytest.mark.parametrize('H,W', [(20, 36), (37, 45), (32, 64)], ids=['20x36', '37x45', '32x64'])
def test_conv5_epilogue_variants_match_torch(bias, res2, H, W, tune64):
    """conv5 of a dense block: 0.2 * conv + x (+ the RRDB's second residual) -> fp32 stream + bf16 shadow, the compile-time epilogues
    232 / 233 / 248 / 249 of the LDS-DMA kernel (round 3: the fp32 residual loads and stores go through v_permlane16_swap so that every
    instruction covers whole 64-byte lines) -- on ragged sizes (partial tiles in both directions) and on both workgroup shapes"""
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, OpList, conv_op
    L = _lib.lib()
    cin, cout, N = 96, 64, 3
    w, b, P, pack, ref = make_conv(cout, cin, 3, 2, 1, dev, 17)
    g = torch.Generator().manual_seed(23)
    x = bf16r(torch.randn(N, cin, H, W, generator=g))
    r1 = torch.randn(N, cout, H, W, generator=g)
    r2 = torch.randn(N, cout, H, W, generator=g)
    xb, r1b, r2b = to_blocked(x, False, dev), to_blocked(r1, True, dev), to_blocked(r2, True, dev)
    of = BTensor(N, cout, H, W, True, dev)
    ob = BTensor(N, cout, H, W, False, dev)
    ops = OpList()
    kw = dict(alpha=0.04, res1=r1b.view(), beta1=0.2, out_f32=of.view(), out_bf16=ob.view())
    if bias:
        kw['bias'] = P.ptr('b')
    if res2:
        kw.update(res2=r2b.view(), beta2=1.0)
    ops.add(conv_op(pack, ref, xb.view(), False, cin, H, W, H, W, N, **kw))
    _lib.check(L.dasr_set_tuning(2, tune64))
    try:
        ops.run()
        torch.cuda.synchronize()
    finally:
        _lib.check(L.dasr_set_tuning(2, 13))
    y = F.conv2d(x.double(), bf16r(w).double(), b.double() if bias else None, padding=1)
    y = 0.04 * y + 0.2 * r1.double() + (r2.double() if res2 else 0.0)
    got = of.nchw().cpu()
    assert rel(got, y.float()) < 2e-5, rel(got, y.float())
    assert (got - y.float()).abs().max().item() < 1e-4   # a mis-addressed pixel would be O(1)
    assert rel(ob.nchw().cpu(), bf16r(y.float())) < 5e-3


def _to_split(x, dev, f16=True):
    """NCHW f32 -> split 16-bit blocked tensor [N][2K][H][W][16]: hi planes, then remainder planes"""
    from dasr_amd.engine import BTensor
    N, C_, H, W = x.shape
    K = (C_ + 15) // 16
    b = to_blocked(x, True, dev).t   # [N][K][H][W][16] f32
    dt = torch.float16 if f16 else torch.bfloat16
    hi = b.to(dt)
    lo = (b - hi.float()).to(dt)
    out = BTensor(N, 32 * K, H, W, False, dev, f16=f16)
    out.t[:, :K] = hi
    out.t[:, K:] = lo
    return out


def _from_split(bt, C_):
    K = bt.planes // 2
    v = bt.t[:, :K].float() + bt.t[:, K:].float()
    return v.permute(0, 1, 4, 2, 3).reshape(bt.N, K * 16, bt.H, bt.W)[:, :C_].cpu()


@pytest.mark.parametrize('f16', [True, False], ids=['f16', 'bf16'])
@pytest.mark.parametrize('cin,cout,mode', [(40, 64, 'fwd'), (64, 32, 'fwd'), (3, 64, 'fwd'), (64, 40, 'dgrad'), (128, 128, 'dgrad'), (64, 3, 'f32out'), (64, 64, 'res')])
def test_conv_on_split_tensors_is_fp32_grade(cin, cout, mode, f16):
    """split 16-bit tensors (dasr_conv_params::in_wrap / out16_lo, round 3): hi planes + remainder planes, the three-term product as ONE launch
    of the LDS-DMA kernel over 3K virtual chunks -- must be as accurate as the split precisions on f32 tensors (prec 3 / 4): 16 (bf16) or 22
    (f16) mantissa bits per operand.  fwd: bias + ReLU -> split output; dgrad: ReLU' mask from a 16-bit activation -> split output;
    f32out: scaled fp32 output (the hand-off at the first / last layer)"""
    dev = _gpu()
    from dasr_amd.engine import BTensor, OpList, conv_op
    N, H, W = 2, 20, 36
    mt = 2 if cout % 64 == 0 else 1
    c16 = lambda c: (c + 15) // 16 * 16
    w, b, P, pack, ref = make_conv(cout, cin, 3, mt, 1, dev, 29)   # (ref unused: a second pack in the virtual-chunk format)
    from dasr_amd.engine import PackRegistry
    pack = PackRegistry(P)
    ref = pack.add(cout, 3 * c16(cin), 9, mt, 5 if f16 else 6, [(0, cout, cin, 0, cin, 0, 0)])
    pack.finalize()
    pack.run()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, cin, H, W, generator=g)
    xs = _to_split(x, dev, f16)
    xv = _from_split(xs, cin).double()        # the values the kernel sees
    dt = torch.float16 if f16 else torch.bfloat16
    wv = (w.to(dt).float() + (w - w.to(dt).float()).to(dt).float()).double()
    y = F.conv2d(xv, wv, None, padding=1)
    kin = c16(cin) // 16
    ops = OpList()
    if mode == 'fwd':
        out = BTensor(N, 2 * c16(cout), H, W, False, dev, f16=f16)
        ops.add(conv_op(pack, ref, xs.view(), False, 3 * c16(cin), H, W, H, W, N, bias=P.ptr('b'), act=1, slope=0.0, out_bf16=out.view(),
                        out16_f16=int(f16), in_wrap=2 * kin, out16_lo=c16(cout) // 16))
        y = F.relu(y + b.double().view(1, -1, 1, 1))
    elif mode == 'dgrad':
        msk = torch.randn(N, cout, H, W, generator=g)
        mb = BTensor(N, cout, H, W, False, dev, f16=f16)
        mb.t.copy_(to_blocked(msk, True, dev).t.to(dt))
        out = BTensor(N, 2 * c16(cout), H, W, False, dev, f16=f16)
        ops.add(conv_op(pack, ref, xs.view(), False, 3 * c16(cin), H, W, H, W, N, mask=mb.view(), mask_f32=0, slope=0.0, out_bf16=out.view(),
                        out16_f16=int(f16), in_wrap=2 * kin, out16_lo=c16(cout) // 16))
        y = torch.where(msk.to(dt).double() > 0, y, torch.zeros_like(y))
    elif mode == 'res':   # bias + a SPLIT residual (res1_lo) -> split output: conv2 of a DSN residual block on split tensors
        r = torch.randn(N, cout, H, W, generator=g)
        rs = _to_split(r, dev, f16)
        out = BTensor(N, 2 * c16(cout), H, W, False, dev, f16=f16)
        ops.add(conv_op(pack, ref, xs.view(), False, 3 * c16(cin), H, W, H, W, N, bias=P.ptr('b'), res1=rs.view(), beta1=1.0, res1_lo=c16(cout) // 16,
                        out_bf16=out.view(), out16_f16=int(f16), in_wrap=2 * kin, out16_lo=c16(cout) // 16))
        y = y + b.double().view(1, -1, 1, 1) + _from_split(rs, cout).double()
    else:
        out = BTensor(N, cout, H, W, True, dev)
        ops.add(conv_op(pack, ref, xs.view(), False, 3 * c16(cin), H, W, H, W, N, alpha=0.125, out_f32=out.view(), in_wrap=2 * kin))
        y = 0.125 * y
    ops.run()
    torch.cuda.synchronize()
    got = out.nchw().cpu() if mode == 'f32out' else _from_split(out, cout)
    tol = 3e-6 if f16 else 1.5e-4   # operand bits: 22 / 16; the dropped lo*lo term and the split of the output are below that
    assert rel(got, y.float()) < tol, rel(got, y.float())
    if mode != 'f32out' and cout % 16:   # padded channels of the last plane pair stay zero
        K = c16(cout) // 16
        assert float(out.t[:, K - 1, :, :, cout % 16:].abs().max()) == 0.0 and float(out.t[:, 2 * K - 1, :, :, cout % 16:].abs().max()) == 0.0


def test_conv_prec3_is_fp32_grade():
    """split-bf16 must be ~fp32 accurate on un-rounded operands (this is what the residual stream relies on)."""
    dev = _gpu()
    from dasr_amd.engine import BTensor, OpList, conv_op
    N, cin, cout, H, W = 1, 64, 64, 32, 32
    w, b, P, pack, ref = make_conv(cout, cin, 3, 1, 3, dev, 3)
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(1))
    xb = to_blocked(x, True, dev)
    of = BTensor(N, cout, H, W, True, dev)
    ops = OpList()
    ops.add(conv_op(pack, ref, xb.view(), True, cin, H, W, H, W, N, bias=P.ptr('b'), out_f32=of.view()))
    ops.run()
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1).float()
    assert rel(of.nchw().cpu(), y) < 3e-5


def test_naive_conv_crosscheck():
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, OpList, conv_op, _stream
    N, cin, cout, H, W = 1, 32, 32, 16, 32
    w, b, P, pack, ref = make_conv(cout, cin, 3, 1, 3, dev, 4)
    x = torch.randn(N, cin, H, W, generator=torch.Generator().manual_seed(2))
    xb = to_blocked(x, True, dev)
    o1, o2 = BTensor(N, cout, H, W, True, dev), BTensor(N, cout, H, W, True, dev)
    op = conv_op(pack, ref, xb.view(), True, cin, H, W, H, W, N, bias=P.ptr('b'), act=1, out_f32=o1.view())
    _lib.check(_lib.lib().dasr_conv(C.byref(op.conv), _stream()))
    op2 = conv_op(pack, ref, xb.view(), True, cin, H, W, H, W, N, bias=P.ptr('b'), act=1, out_f32=o2.view())
    _lib.check(_lib.lib().dasr_conv_naive(C.byref(op2.conv), P.ptr('w'), _stream()))
    torch.cuda.synchronize()
    assert rel(o1.t.float().cpu(), o2.t.float().cpu()) < 3e-5


WG_CASES = [
    # name, kh, stride, cin, cout, H, W (input), ups, f32
    ('rdb', 3, 1, 96, 32, 24, 40, 0, False),
    ('rdb64', 3, 1, 192, 64, 16, 16, 0, False),
    ('stream_f32', 3, 1, 64, 64, 20, 28, 0, True),
    ('ups_f32', 3, 1, 64, 64, 9, 13, 1, True),
    ('cin3', 3, 1, 3, 64, 16, 24, 0, True),
    ('cout3', 3, 1, 64, 3, 16, 24, 0, True),
    ('k4s2', 4, 2, 16, 64, 32, 40, 0, True),
    ('k4s1', 4, 1, 64, 32, 15, 18, 0, True),
]


@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad_matches_torch(case):
    dev = _gpu()
    from dasr_amd.engine import ParamStore, WgradGroup, Workspace, OpList
    name, kh, stride, cin, cout, H, W, ups, f32 = case
    N = 2
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, cin, H, W, generator=g)
    HL, WL = (2 * H, 2 * W) if ups else (H, W)
    Ho = (HL + 2 - kh) // stride + 1
    Wo = (WL + 2 - kh) // stride + 1
    gy = torch.randn(N, cout, Ho, Wo, generator=g)
    xq, gq = bf16r(x), bf16r(gy)  # the kernel rounds both operands to bf16
    xb = to_blocked(x if f32 else xq, f32, dev)
    gb = to_blocked(gy if f32 else gq, f32, dev)
    P = ParamStore([('weight', (cout, cin, kh, kh)), ('bias', (cout,))], dev)
    ws = Workspace(dev)
    grp = WgradGroup(kh, stride)
    grp.add_conv(gb.view, f32, gb.planes, xb.view, f32, xb.planes, cout, cin, H, W, Ho, Wo, N, P.off('weight'), P.off('bias'), ups=ups)
    grp.finalize(ws, dev, target_wgs=24)
    ops = OpList()
    for o in grp.ops(P.grad.data_ptr()):
        ops.add(o)
    ws.finalize()
    ops.run()
    torch.cuda.synchronize()
    xx = xq.double().requires_grad_(False)
    if ups:
        xx = F.interpolate(xx, scale_factor=2, mode='nearest')
    wt = torch.zeros(cout, cin, kh, kh, dtype=torch.double, requires_grad=True)
    y = F.conv2d(xx, wt, None, stride=stride, padding=1)
    (y * gq.double()).sum().backward()
    dw = wt.grad.float()
    db = (gy if f32 else gq).double().sum(dim=(0, 2, 3)).float()  # f32 mode sums the unrounded gradients
    assert rel(P.view('weight', P.grad).cpu(), dw) < 2e-5, name
    assert rel(P.view('bias', P.grad).cpu(), db) < 2e-5, name


def test_pack_backward_is_transposed_flipped():
    """dgrad through the conv kernel with a transposed/flipped pack == autograd input gradient."""
    dev = _gpu()
    from dasr_amd.engine import ParamStore, PackRegistry, BTensor, OpList, conv_op
    N, cin, cout, H, W = 1, 64, 32, 16, 32
    g = torch.Generator().manual_seed(21)
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    P = ParamStore([('w', (cout, cin, 3, 3))], dev)
    P.load_state_dict({'w': w})
    pack = PackRegistry(P)
    refb = pack.add(cin, cout, 9, 1, 3, [(0, cout, cin, 0, cout, 0, 1)])
    pack.finalize()
    pack.run()
    gy = torch.randn(N, cout, H, W, generator=g)
    gb = to_blocked(gy, True, dev)
    of = BTensor(N, cin, H, W, True, dev)
    ops = OpList()
    ops.add(conv_op(pack, refb, gb.view(), True, cout, H, W, H, W, N, out_f32=of.view()))
    ops.run()
    x = torch.zeros(N, cin, H, W, dtype=torch.double, requires_grad=True)
    (F.conv2d(x, w.double(), padding=1) * gy.double()).sum().backward()
    assert rel(of.nchw().cpu(), x.grad.float()) < 3e-5


def test_elementwise_and_adam():
    dev = _gpu()
    from dasr_amd import _lib
    from dasr_amd.engine import BTensor, _stream, NULL_T
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    # nchw <-> blocked round trip
    x = torch.rand(2, 3, 10, 14, generator=g)
    xd = x.to(dev)
    b = BTensor(2, 16, 10, 14, True, dev)
    _lib.check(L.dasr_nchw_to_blocked(xd.data_ptr(), 2, 3, 10, 14, b.view(), NULL_T, _stream()))
    back = torch.zeros_like(xd)
    _lib.check(L.dasr_blocked_to_nchw(b.view(), 2, 3, 10, 14, back.data_ptr(), _stream()))
    assert torch.equal(back.cpu(), x)
    assert float(b.t[:, :, :, :, 3:].abs().max()) == 0.0
    # L1 loss + gradient
    hr = torch.rand(2, 3, 10, 14, generator=g)
    hrd = hr.to(dev)
    acc = torch.zeros(4, device=dev)
    gr = BTensor(2, 16, 10, 14, True, dev)
    coef = 1.0 / x.numel()
    _lib.check(L.dasr_l1_loss(b.view(), hrd.data_ptr(), None, 2, 3, 10, 14, coef, acc.data_ptr(), gr.view(), 0, 0.0, _stream()))
    assert abs(float(acc[0]) - float((x - hr).abs().mean())) < 1e-6
    assert torch.allclose(gr.nchw(3).cpu(), torch.sign(x - hr) * coef, atol=1e-9)
    # ... the same gradient straight into an f16 tensor, pre-scaled by a power of two (round 6: no padded fp32 image + conversion pass in front of the f16 HR tail)
    g16 = BTensor(2, 16, 10, 14, False, dev, f16=True)
    g16.t.fill_(0.0)
    _lib.check(L.dasr_l1_loss(b.view(), hrd.data_ptr(), None, 2, 3, 10, 14, coef, None, g16.view(), 4, 1024.0, _stream()))
    assert torch.equal(g16.nchw(3).cpu(), (torch.sign(x - hr) * coef * 1024.0).half().float()) and float(g16.t[:, :, :, :, 3:].abs().max()) == 0.0
    # 2x2 down-sum with LeakyReLU' mask
    s = torch.randn(1, 32, 8, 12, generator=g)
    m = torch.randn(1, 32, 4, 6, generator=g)
    sb, mb = to_blocked(s, True, dev), to_blocked(m, True, dev)
    db = BTensor(1, 32, 4, 6, True, dev)
    _lib.check(L.dasr_downsum2x(sb.view(), 1, 32, 4, 6, mb.view(), 1, 0.2, db.view(), NULL_T, _stream()))
    want = F.avg_pool2d(s, 2) * 4
    want = torch.where(m > 0, want, want * 0.2)
    assert torch.allclose(db.nchw().cpu(), want, atol=1e-5)
    # Adam vs torch.optim.Adam, 3 steps, with weight decay
    p0 = torch.randn(1000, generator=g)
    grads = [torch.randn(1000, generator=g) for _ in range(3)]
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=1e-3, betas=(0.9, 0.999), weight_decay=0.01)
    pd, md, vd = p0.to(dev), torch.zeros(1000, device=dev), torch.zeros(1000, device=dev)
    for i, gg in enumerate(grads):
        pt.grad = gg.clone()
        opt.step()
        gd = gg.to(dev)
        _lib.check(L.dasr_adam(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), 1000, 1e-3, 0.9, 0.999, 1e-8, 0.01, i + 1, None, None, _stream()))
    assert torch.allclose(pd.cpu(), pt.detach(), rtol=1e-5, atol=1e-7)
