"""Host-side plan builder for the MI355X kernels: parameter store (reference state_dict layout),
NC16HW16 activation buffers, packed-weight registry and recorded op lists that libdasr_hip.so executes.

PyTorch is used only as the device allocator / stream owner; all arithmetic runs in the HIP library.
"""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib
from ._lib import Tensor, ConvParams, WgradPart, WgradReducePart, PackDesc, Op

SLOPE = 0.2

# time buckets of a training step (OpList.tag / bench.py `buckets`)
BUCKETS = {1: 'G forward: trunk fea_conv .. LR_conv (DSN: whole generator)', 2: 'G HR tail forward', 3: 'G HR tail backward (data + weight gradients)',
           4: 'G trunk data gradient (DSN: generator backward)', 5: 'G trunk weight gradients (deferred phase + reduce)', 6: 'perceptual net forward, gradient images',
           7: 'perceptual net forward, target images (one pass)', 8: 'perceptual net data gradient', 9: 'discriminator forward',
           10: 'discriminator backward (data + weight gradients)', 11: 'G other backward (LR_conv, fea_conv)', 0: 'losses / filters / layout / optimiser'}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ceil_div(a, b):
    return (a + b - 1) // b


# --------------------------------------------------------------------------------------------------
class BTensor:
    """Blocked activation tensor T[n][cb][y][x][16] (bf16 or f32) living in a torch CUDA tensor."""

    def __init__(self, N, C_, H, W, f32, device, f16=False):
        """f16 (with f32 False): the 16-bit elements are IEEE half instead of bfloat16 (HR tail of the generator)"""
        self.N, self.C, self.H, self.W, self.f32 = N, C_, H, W, f32
        self.planes = ceil_div(C_, 16)
        self.t = torch.zeros((N, self.planes, H, W, 16), dtype=torch.float32 if f32 else (torch.float16 if f16 else torch.bfloat16), device=device)
        self.esz = 4 if f32 else 2

    @classmethod
    def wrap(cls, t, C_, f32):
        """blocked tensor over an existing torch tensor [N][planes][H][W][16] (e.g. images [n0, n1) of a larger allocation)"""
        assert t.dim() == 5 and t.shape[4] == 16 and t.is_contiguous() and t.shape[1] == ceil_div(C_, 16)
        b = cls.__new__(cls)
        b.N, b.C, b.H, b.W, b.f32 = t.shape[0], C_, t.shape[2], t.shape[3], f32
        b.planes, b.t, b.esz = t.shape[1], t, 4 if f32 else 2
        return b

    def view(self, c0=0):
        """dasr_tensor starting at channel c0 (multiple of 16)."""
        assert c0 % 16 == 0 and c0 // 16 < self.planes, (c0, self.planes)
        cbs = self.H * self.W * 16
        return Tensor(self.t.data_ptr() + (c0 // 16) * cbs * self.esz, self.planes * cbs, cbs)

    def nchw(self, C_=None):
        """Debug/IO helper: gather to a torch NCHW float tensor (torch indexing only, no arithmetic)."""
        C_ = C_ or self.C
        return self.t.permute(0, 1, 4, 2, 3).reshape(self.N, self.planes * 16, self.H, self.W)[:, :C_].float()


NULL_T = Tensor(None, 0, 0)


# --------------------------------------------------------------------------------------------------
class ParamStore:
    """Flat fp32 parameter / gradient / Adam-moment buffers with the reference's state_dict keys."""

    def __init__(self, spec, device):
        self.spec = OrderedDict()
        off = 0
        for k, shape in spec:
            n = 1
            for s in shape:
                n *= s
            self.spec[k] = (off, tuple(shape), n)
            off += n
        self.total = off
        self.device = device
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)

    def off(self, k):
        return self.spec[k][0]

    def ptr(self, k, buf=None):
        buf = self.flat if buf is None else buf
        return buf.data_ptr() + 4 * self.spec[k][0]

    def view(self, k, buf=None):
        o, shape, n = self.spec[k]
        return (self.flat if buf is None else buf)[o:o + n].view(shape)

    def state_dict(self):
        return OrderedDict((k, self.view(k).detach().clone().cpu()) for k in self.spec)

    def grad_dict(self):
        return OrderedDict((k, self.view(k, self.grad).detach().clone().cpu()) for k in self.spec)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.spec if k not in sd]
        unexpected = [k for k in sd if k not in self.spec]
        if strict and (missing or unexpected):
            raise RuntimeError('Error(s) in loading state_dict: missing %s unexpected %s' % (missing, unexpected))
        for k in self.spec:
            if k in sd:
                v = sd[k]
                if tuple(v.shape) != self.spec[k][1]:
                    raise RuntimeError('size mismatch for %s: %s vs %s' % (k, tuple(v.shape), self.spec[k][1]))
                self.view(k).copy_(v.to(self.device, torch.float32))


# --------------------------------------------------------------------------------------------------
class PackedRef:
    __slots__ = ('off', 'lo_off', 'cout', 'cin_pad', 'ntaps', 'mt', 'prec')


class PackRegistry:
    """bf16 MFMA-fragment-ordered weight copies, rebuilt from the fp32 masters by one kernel launch."""

    def __init__(self, params):
        self.params = params
        self.descs = []
        self.prefix = [0]
        self.size = 0  # bf16 elements
        self.buf = None

    def add(self, cout, cin_pad, ntaps, mt, prec, segs, tapmap=None, src_ntaps=None, tapmasks=None):
        """tapmap: packed tap -> source tap.  Default: identity for forward segments, reversed (tap flip) when the
        segments are transposed (stride-1 data-gradient)."""
        assert cin_pad % 16 == 0 and len(segs) <= 5 and ntaps <= 32
        mg = ceil_div(ceil_div(cout, 32), mt)
        pieces = mg * (cin_pad // 16) * ntaps * mt * 64
        r = PackedRef()
        r.off, r.cout, r.cin_pad, r.ntaps, r.mt, r.prec = self.size, cout, cin_pad, ntaps, mt, prec
        r.lo_off = pieces * 8 if prec in (3, 4) else 0
        d = PackDesc()
        # prec 2: f16 operands (one MFMA pass); 4: f16 hi + lo planes; 1 / 3: bf16 hi (+ lo) planes; 5 (f16) / 6 (bf16): split 16-bit TENSORS -- cin_pad
        # counts 3K virtual chunks [hi | hi | lo] for one launch of the LDS-DMA kernel (conv_op in_wrap); the conv itself runs as prec 2 / 1
        d.fmt = {2: 1, 4: 2, 5: 3, 6: 4}.get(prec, 0)
        if prec in (5, 6):
            assert cin_pad % 48 == 0
            r.prec = 2 if prec == 5 else 1
        d.dst_off, d.lo_off, d.cout, d.cin_pad, d.ntaps, d.mt, d.nseg = self.size, r.lo_off, cout, cin_pad, ntaps, mt, len(segs)
        d.src_ntaps = src_ntaps or ntaps
        if tapmap is None:
            tr = bool(segs and segs[0][6])
            tapmap = [ntaps - 1 - t for t in range(ntaps)] if tr else list(range(ntaps))
        for t in range(32):
            d.tapmap[t] = tapmap[t] if t < len(tapmap) else -1
        for t in range(16):  # packed tap = sum of the source taps in the bit mask (sub-pixel upconv)
            d.tapmask[t] = tapmasks[t] if (tapmasks is not None and t < len(tapmasks)) else 0
        for i, s in enumerate(segs):
            sg = d.seg[i]
            sg.src_off, sg.src_cout, sg.src_cin, sg.cin_start, sg.cin_len, sg.src_c0, sg.transpose = s
        self.descs.append(d)
        self.prefix.append(self.prefix[-1] + pieces)
        self.size += pieces * 8 * (2 if prec in (3, 4) else 1)
        return r

    def finalize(self):
        dev = self.params.device
        self.buf = torch.zeros(self.size, dtype=torch.bfloat16, device=dev)
        arr = (PackDesc * len(self.descs))(*self.descs)
        self.desc_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.prefix_dev = torch.tensor(self.prefix, dtype=torch.int64, device=dev)

    def ptr(self, ref):
        return self.buf.data_ptr() + 2 * ref.off

    def op(self):
        o = Op()
        o.op = _lib.OP_PACK
        o.p[0], o.i[0], o.l[0] = self.desc_dev.data_ptr(), len(self.descs), self.prefix[-1]
        o.p[1], o.p[2], o.p[3] = self.prefix_dev.data_ptr(), self.params.flat.data_ptr(), self.buf.data_ptr()
        return o

    def run(self):
        ops = (Op * 1)(self.op())
        _lib.check(_lib.lib().dasr_run_ops(C.cast(ops, C.c_void_p), 1, _stream()), 'pack')


# --------------------------------------------------------------------------------------------------
class OpList:
    def __init__(self):
        self.ops = []
        self.keep = []  # device tables that must outlive the list
        self._arr = None

    def add(self, o):
        self.ops.append(o)
        self._arr = None

    def extend(self, other):
        self.ops.extend(other.ops)
        self.keep.extend(other.keep)
        self._arr = None

    def tag(self, bucket, start=0, end=None):
        """time-bucket tag (Op.i[7], read by no kernel; bench.py's per-bucket table) on the not yet tagged ops [start, end)"""
        for o in self.ops[start:end]:
            if o.i[7] == 0:
                o.i[7] = bucket
        self._arr = None
        return self

    def set_f(self, idx, slot, value):
        """patch float argument `slot` of op `idx` in place (see set_i)"""
        self.ops[idx].f[slot] = value
        if self._arr is not None:
            self._arr[idx].f[slot] = value

    def set_i(self, idx, slot, value):
        """patch integer argument `slot` of op `idx` in place (recorded list AND its ctypes image): for the few arguments that change from call to
        call while the plan stays recorded (e.g. the random symmetry of DSN --lpips_rot_flip)"""
        self.ops[idx].i[slot] = value
        if self._arr is not None:
            self._arr[idx].i[slot] = value

    def safe_cuts(self, chunk):
        """[0, c1, c2, ..., len]: chunk boundaries of about `chunk` ops at which no stream redirect (OP_SET_STREAM) is open"""
        key = ('cuts', chunk, len(self.ops))
        if getattr(self, '_cuts_key', None) != key:
            cuts, redirected, last = [0], False, 0
            for i, o in enumerate(self.ops):
                if o.op == _lib.OP_SET_STREAM:
                    redirected = bool(o.p[0])
                if i + 1 - last >= chunk and not redirected:
                    cuts.append(i + 1)
                    last = i + 1
            if cuts[-1] != len(self.ops):
                cuts.append(len(self.ops))
            self._cuts, self._cuts_key = cuts, key
        return self._cuts

    def run(self, lo=0, hi=None):
        """enqueue ops[lo:hi] on the current stream"""
        if not self.ops:
            return
        if self._arr is None:
            self._arr = (Op * len(self.ops))(*self.ops)
        hi = len(self.ops) if hi is None else min(hi, len(self.ops))
        if hi <= lo:
            return
        L = _lib.lib()
        rc = L.dasr_run_ops(C.c_void_p(C.addressof(self._arr) + lo * C.sizeof(Op)), hi - lo, _stream())
        if rc != 0:
            k = L.dasr_last_failed_op() + lo
            raise _lib.DasrHipError('dasr_run_ops: op #%d (kind %d) failed with code %d' % (k, self.ops[k].op if 0 <= k < len(self.ops) else -1, rc))


def run_interleaved(lists, streams, chunk=None):
    """Enqueue several op lists on their own streams, alternating in chunks so that every hardware queue gets work
    early (the host enqueues ~250k launches/s; a whole 1200-op list first would leave the other stream idle for ms)."""
    chunk = chunk or 48
    # dasr_run_ops forgets a DASR_OP_SET_STREAM redirect when it returns, so a list is only cut where it is back on its own stream
    cuts = [l.safe_cuts(chunk) for l in lists]
    for k in range(max(len(c) for c in cuts) - 1):
        for l, st, c in zip(lists, streams, cuts):
            if k + 1 < len(c):
                with torch.cuda.stream(st):
                    l.run(c[k], c[k + 1])


def run_parallel(lists, streams):
    """Enqueue independent op lists on their own streams.  Default: one enqueue THREAD per list inside the library (dasr_run_ops_mt) -- with
    three or four sub-batch replicas a step is 5 000+ launches, more than one host thread feeds in the time the GPU needs for them.
    DASR_ENQ=chunk: the single-threaded chunk interleave of rounds 1-3 (run_interleaved).  While a stream capture is open (hipGraph) the
    lists are enqueued by the capturing thread."""
    import os
    lists = [l for l in lists]
    if len(lists) == 1 or os.environ.get('DASR_ENQ', 'mt') == 'chunk' or torch.cuda.is_current_stream_capturing():
        return run_interleaved(lists, streams)
    n = len(lists)
    for l in lists:
        if l._arr is None:
            l._arr = (Op * len(l.ops))(*l.ops)
    ptrs = (C.c_void_p * n)(*[C.addressof(l._arr) for l in lists])
    cnts = (C.c_int32 * n)(*[len(l.ops) for l in lists])
    sts = (C.c_void_p * n)(*[st.cuda_stream for st in streams])
    L = _lib.lib()
    rc = L.dasr_run_ops_mt(ptrs, cnts, sts, n)
    if rc != 0:
        k = L.dasr_last_failed_op()
        li, oi = (k >> 24) & 0x7f, k & 0xffffff
        kind = lists[li].ops[oi].op if (0 <= li < n and 0 <= oi < len(lists[li].ops)) else -1
        raise _lib.DasrHipError('dasr_run_ops_mt: list %d op #%d (kind %d) failed with code %d' % (li, oi, kind, rc))


def conv_op(pack, ref, inp, in_f32, cin, Hin, Win, Hout, Wout, N, bias=None, kh=3, stride=1, pad=1, ups=0, act=0, slope=SLOPE,
            mask=None, mask_f32=0, alpha=1.0, res1=None, beta1=0.0, res2=None, beta2=0.0, out_f32=None, out_bf16=None, gamma=1.0,
            pad_x=-1, out_stride=1, out_oy=0, out_ox=0, out_W=0, slope_ptr=None, in_stride=1, in_oy=0, in_ox=0, in_W=0, flops=None, in_scale=0.0, out16_f16=0,
            in_wrap=0, out16_lo=0, res1_lo=0):
    """flops: algorithmic FLOPs of the reference op this launch stands for (default: 2 * outputs * taps * cin * cout of the launch
    itself; the sub-pixel upconv launches pass a quarter of the reference's 3x3 conv on the up-sampled grid instead)."""
    assert cin == ref.cin_pad, (cin, ref.cin_pad)
    o = Op()
    o.op = _lib.OP_CONV
    o.flops = float(flops) if flops is not None else 2.0 * N * Hout * Wout * kh * kh * (cin // 3 if in_wrap else cin) * ref.cout
    p = o.conv
    p.inp, p.in_f32, p.Hin, p.Win, p.ups, p.cin = inp, int(in_f32), Hin, Win, ups, cin
    p.w, p.w_lo_off, p.bias = pack.ptr(ref), ref.lo_off, bias
    p.cout, p.Hout, p.Wout, p.N = ref.cout, Hout, Wout, N
    p.kh, p.stride, p.pad, p.prec, p.mt = kh, stride, pad, ref.prec, ref.mt
    p.act, p.slope = act, slope
    p.mask, p.mask_f32 = (mask if mask is not None else NULL_T), mask_f32
    p.alpha = alpha
    p.res1, p.beta1 = (res1 if res1 is not None else NULL_T), beta1
    p.res2, p.beta2 = (res2 if res2 is not None else NULL_T), beta2
    p.out_f32 = out_f32 if out_f32 is not None else NULL_T
    p.out_bf16 = out_bf16 if out_bf16 is not None else NULL_T
    p.gamma = gamma
    p.pad_x, p.out_stride, p.out_oy, p.out_ox, p.out_W = pad_x, out_stride, out_oy, out_ox, out_W
    p.slope_ptr = slope_ptr
    p.in_stride, p.in_oy, p.in_ox, p.in_W = in_stride, in_oy, in_ox, in_W
    p.in_scale = in_scale if (ref.prec in (2, 4) and in_f32) else 0.0   # power-of-two pre-scale of an f32 gradient input before its f16 rounding
    p.out16_f16 = int(out16_f16)
    p.res1_lo = int(res1_lo)   # res1 is a split 16-bit tensor: its remainder planes start res1_lo planes after the hi planes
    p.in_wrap, p.out16_lo = int(in_wrap), int(out16_lo)   # split 16-bit tensors (SplitTensor): 2K input planes before the hi planes repeat; K' output planes
    return o


class ConvChain:
    """One persistent launch for a chain of dense-block conv ops of one geometry (dasr_conv_chain, include/dasr_hip.h): the forward trunk of the
    generator without kernel boundaries between its layers.  `ops`: the conv Ops in execution order (their parameter blocks are copied);
    dep_chunks[i]: first 16-channel input chunk of layer i that holds layer i - 1's output (<= 0: all of them).  flags / err: shared per (N, h, w)
    plan -- the flag words count stages monotonically across launches, err stays zero unless a launch went wrong (check())."""

    def __init__(self, ops, dep_chunks, n_images, n_tiles, device, flags=None, err=None, form='layer'):
        assert len(ops) == len(dep_chunks) and ops
        assert form in ('layer', 'is')   # 'is': the input-stationary launch (dasr_rdb_chain): whole dense blocks, five layers each
        self.form = form
        n = len(ops)
        self.host = (_lib.ConvParams * n)(*[o.conv for o in ops])
        raw = torch.frombuffer(bytearray(bytes(self.host)), dtype=torch.uint8)
        self.dev = raw.to(device)
        self.dep = torch.tensor([int(d) for d in dep_chunks], dtype=torch.int32, device=device)
        self.flags = flags if flags is not None else torch.zeros(n_images * n_tiles + 8, dtype=torch.int32, device=device)   # + the per-XCD ticket counters
        self.err = err if err is not None else torch.zeros(1, dtype=torch.int32, device=device)
        self.n = n
        self.flops = sum(o.flops for o in ops)

    def op(self):
        o = Op()
        o.op = _lib.OP_RDB_CHAIN if self.form == 'is' else _lib.OP_CONV_CHAIN
        o.p[0], o.p[1], o.p[2], o.p[3] = self.dev.data_ptr(), C.cast(self.host, C.c_void_p).value, self.dep.data_ptr(), self.flags.data_ptr()
        o.l[0], o.i[0] = self.err.data_ptr(), self.n
        o.flops = self.flops
        return o

    def check(self):
        """host sync: raises if a launch of this chain flagged a broken neighbour wait / XCD placement"""
        e = int(self.err.item())
        if e:
            raise RuntimeError('conv chain: device error word %d (bit 1: a neighbour wait gave up)' % e)


class WgradGroup:
    """Parts of one wgrad launch (same kernel size / stride) + the matching reduce table."""

    def __init__(self, kh, stride):
        self.kh, self.stride = kh, stride
        self.parts = []  # (WgradPart, WgradReducePart)

    def add_conv(self, g, g_f32, g_planes_total, inp, in_f32, in_planes_total, cout, cin, Hin, Win, Hout, Wout, N,
                 dst_w_off, dst_b_off, pad=None, ups=0, f16=False, g_scale=0.0, split=None, more_pairs=()):
        """g / inp are BTensor-like callables c0 -> dasr_tensor view.  f16: the f32 tensors are rounded to f16 (g pre-scaled by the
        power of two g_scale) instead of bf16 while staging; the reduce op undoes the scale.
        split = (g_lo, inp_lo) (views like g / inp, filled by dasr_f16_residual): 22-bit operands on the f16 MFMA -- every part becomes THREE
        parts g.x, g.x_lo, g_lo.x whose partial sums lie behind one another as 3 * nsplit splits of ONE reduce part (the bias partials, which come
        from the unrounded gradient, only from the first).
        more_pairs = [(g2, inp2), ...]: further (gradient, input) pairs whose products are summed into the SAME weight gradient (the second-order pass
        of the DSN's --wgan penalty: dW = adj_z (x) a + adj_zdot (x) adot); same mechanism, the bias partials come from the first pair only."""
        assert not f16 or (g_f32 and in_f32)
        assert split is None or f16
        self.f16 = bool(f16) or getattr(self, 'f16', False)
        self.g_scale = float(g_scale) if f16 and g_scale else getattr(self, 'g_scale', 0.0)
        ntaps = self.kh * self.kh
        tpp = ntaps if ntaps <= 16 else 10  # WCfg::TAPS_PER_PART
        self.tpp = tpp
        self.flops = getattr(self, 'flops', 0.0) + 2.0 * N * Hout * Wout * ntaps * cin * cout
        pad = (self.kh - 1) // 2 if pad is None else pad
        cin_pad = ceil_div(cin, 16) * 16
        variants = ([(g, inp)] if split is None else [(g, inp), (g, split[1]), (split[0], inp)]) + list(more_pairs)
        for oc0 in range(0, cout, 32):
            for c0 in range(0, cin_pad, 64):
                for tap0 in range(0, ntaps, tpp):
                    rp = WgradReducePart()
                    first = dst_b_off is not None and c0 == 0 and tap0 == 0
                    wps = []
                    for vi, (gv, xv) in enumerate(variants):
                        wp = WgradPart()
                        wp.g, wp.g_f32 = gv(oc0), int(g_f32)
                        wp.inp, wp.in_f32 = xv(c0), int(in_f32)
                        wp.ups = ups
                        wp.n_ctiles = min(2, ceil_div(cin_pad - c0, 32))
                        wp.g_planes = min(2, g_planes_total - oc0 // 16)
                        wp.in_planes = min(4, in_planes_total - c0 // 16)
                        wp.Hin, wp.Win, wp.Hout, wp.Wout, wp.N = Hin, Win, Hout, Wout, N
                        wp.kh, wp.stride, wp.pad, wp.tap0 = self.kh, self.stride, pad, tap0
                        wp.want_bias = 1 if (first and vi == 0) else 0
                        wp.g_scale = float(g_scale) if f16 else 0.0
                        wps.append(wp)
                    rp.ntaps, rp.oc0, rp.c0, rp.cout, rp.cin, rp.n_ctiles = min(tpp, ntaps - tap0), oc0, c0, cout, cin, wps[0].n_ctiles
                    rp.tap0, rp.ntaps_total = tap0, ntaps
                    rp.dst_w_off = dst_w_off
                    rp.dst_b_off = dst_b_off if first else -1
                    for wp in wps:
                        self.parts.append((wp, rp if wp is wps[0] else None, len(wps) if wp is wps[0] else 0))

    def finalize(self, workspace, device, target_wgs=768):
        wp0 = self.parts[0][0]
        ph = 2 if self.stride == 2 else (8 if self.kh in (3, 1) else 4)  # WCfg::PH in wgrad.hip
        ntiles = wp0.N * ceil_div(wp0.Hout, ph) * ceil_div(wp0.Wout, 16)
        nparts = len(self.parts)
        self.nsplit = max(1, min(ntiles, target_wgs // nparts))
        if self.nsplit >= 16:
            self.nsplit -= self.nsplit % 8  # same pixel split -> same XCD (block id % 8) for every part: shared L2 lines
        off = 0
        reds = []
        for wp, rp, nvar in self.parts:   # a reduce part owns `nvar` consecutive wgrad parts (split operands: 3) = nvar * nsplit splits
            wp.ws_off = off
            if rp is not None:
                rp.ws_off = off
                rp.nsplit = self.nsplit * nvar
                rp.bias_nsplit = self.nsplit if nvar > 1 else 0
                rp.split_stride, rp.tap_stride, rp.bias_stride = self.tpp * 2048, 2048, 32
                reds.append(rp)
            off += self.nsplit * self.tpp * 2048
        for wp, rp, nvar in self.parts:   # bias partials behind all weight partials (only the first variant of a conv's first part writes them)
            wp.ws_bias_off = off
            if rp is not None:
                rp.ws_bias_off = off
            off += self.nsplit * 32
        self.ws_floats = off
        self.workspace = workspace
        workspace.reserve(off)
        wa = (WgradPart * nparts)(*[p[0] for p in self.parts])
        ra = (WgradReducePart * len(reds))(*reds)
        self.n_red = len(reds)
        self.w_dev = torch.frombuffer(bytearray(bytes(wa)), dtype=torch.uint8).to(device)
        self.r_dev = torch.frombuffer(bytearray(bytes(ra)), dtype=torch.uint8).to(device)

    def ops(self, grad_ptr, scale=1.0):
        """[wgrad, reduce] ops; the workspace pointer is patched in by Workspace.finalize()."""
        a, b = Op(), Op()
        a.op = _lib.OP_WGRAD
        a.p[0], a.i[0], a.i[1], a.i[2], a.i[3] = self.w_dev.data_ptr(), len(self.parts), self.nsplit, self.kh, self.stride
        f32s = set((p[0].g_f32, p[0].in_f32) for p in self.parts)
        assert f32s in ({(0, 0)}, {(1, 1)}), 'a wgrad group must be all-bf16 or all-f32'
        a.i[4] = self.parts[0][0].g_f32 | (2 if getattr(self, 'f16', False) else 0)
        a.flops = float(getattr(self, 'flops', 0.0))
        b.op = _lib.OP_WGRAD_REDUCE
        b.p[0], b.i[0], b.p[2], b.f[0] = self.r_dev.data_ptr(), self.n_red, grad_ptr, scale
        b.i[1] = 1 if (self.tpp <= 16 and all(rp is None or rp.nsplit <= 4 for _, rp, _ in self.parts)) else 0   # few_splits: one reduce workgroup per output channel
        gs = getattr(self, 'g_scale', 0.0)
        b.f[1] = 1.0 / gs if gs else 0.0   # second factor of the reduce scale (f[0] stays the data-parallel 1/world): undoes the f16 pre-scale
        self.workspace.register(a, b)
        return [a, b]


class WgradGroup3:
    """Parts of one 6-wave 3x3 wgrad launch (csrc/wgrad.hip, wgrad3_kernel): a part = one 64-channel input block x up
    to three consecutive 32-oc tiles of a gradient tensor; every (part, oc tile) has its own reduce descriptor."""

    kh, stride = 3, 1

    def __init__(self):
        self.parts = []   # (WgradPart, [(ot, dict)])

    def add_block(self, g_view, g_planes, in_view, in_planes, n_ctiles, Hin, Win, Hout, Wout, N, tiles, want_bias, f32=False, ups=0):
        """tiles: per oc tile of the part: dict(dst_w_off, dst_b_off|None, cout, cin, oc0, c0, n_ctiles) or None (unused tile)"""
        assert 1 <= len(tiles) <= 3 and g_planes <= 6
        wp = WgradPart()
        wp.g, wp.g_f32, wp.inp, wp.in_f32 = g_view, int(f32), in_view, int(f32)
        wp.ups, wp.n_ctiles, wp.g_planes, wp.in_planes = ups, n_ctiles, g_planes, in_planes
        wp.Hin, wp.Win, wp.Hout, wp.Wout, wp.N = Hin, Win, Hout, Wout, N
        wp.kh, wp.stride, wp.pad, wp.want_bias = 3, 1, 1, int(bool(want_bias))
        self.parts.append((wp, [(i, t) for i, t in enumerate(tiles) if t is not None]))

    def finalize(self, workspace, device, target_wgs=256, ppu=0):
        """ppu: the parts come in units of `ppu` consecutive parts that read the same tensors (one dense block); when the split count is not
        a multiple of 8 the kernel then places each (unit, split) on one XCD (csrc/wgrad.hip, w3_block_map)"""
        wp0 = self.parts[0][0]
        ntiles = wp0.N * ceil_div(wp0.Hout, 8) * ceil_div(wp0.Wout, 16)
        nparts = len(self.parts)
        self.nsplit = max(1, min(ntiles, target_wgs // nparts))
        if self.nsplit >= 8:
            # a multiple of 8: workgroup b runs on XCD b % 8, so with block = part * nsplit + split every part's workgroup of a given pixel split
            # lands on the same XCD and the G / X tiles the parts share are fetched once per XCD.  17 splits (255 workgroups instead of 240)
            # was 3 % faster on the launch but fetched 60 % more (351 vs 220 MB raw FETCH_SIZE per RRDB launch): not kept
            self.nsplit -= self.nsplit % 8
        self.ppu = 0
        if ppu and self.nsplit % 8 and nparts % ppu == 0 and ((nparts // ppu) * self.nsplit) % 8 == 0 and ppu < 256:
            self.ppu = ppu
        off, red = 0, []
        for wp, tiles in self.parts:
            wp.ws_off = off
            off += self.nsplit * 9 * 3 * 2048
            wp.ws_bias_off = off
            off += self.nsplit * 96
            for ot, t in tiles:
                rp = WgradReducePart()
                rp.ws_off, rp.ws_bias_off = wp.ws_off + ot * 2048, wp.ws_bias_off + ot * 32
                rp.nsplit, rp.ntaps, rp.oc0, rp.c0 = self.nsplit, 9, t['oc0'], t['c0']
                rp.cout, rp.cin, rp.n_ctiles = t['cout'], t['cin'], t['n_ctiles']
                rp.dst_w_off = t['dst_w_off']
                rp.dst_b_off = t['dst_b_off'] if t.get('dst_b_off') is not None else -1
                rp.split_stride, rp.tap_stride, rp.bias_stride = 9 * 3 * 2048, 3 * 2048, 96
                red.append(rp)
        self.n_red = len(red)
        self.ws_floats = off
        self.workspace = workspace
        workspace.reserve(off)
        wa = (WgradPart * nparts)(*[p[0] for p in self.parts])
        ra = (WgradReducePart * len(red))(*red)
        self.w_dev = torch.frombuffer(bytearray(bytes(wa)), dtype=torch.uint8).to(device)
        self.r_dev = torch.frombuffer(bytearray(bytes(ra)), dtype=torch.uint8).to(device)

    def ops(self, grad_ptr, scale=1.0):
        a, b = Op(), Op()
        a.op = _lib.OP_WGRAD
        f16 = getattr(self, 'f16', False)   # 16-bit f16 tensors (gradient pre-scaled by g_scale): f16 MFMA, reduce scale x 1 / g_scale
        a.p[0], a.i[0], a.i[1], a.i[2], a.i[3], a.i[4] = (self.w_dev.data_ptr(), len(self.parts), self.nsplit | (getattr(self, 'ppu', 0) << 16), 33, 1,
                                                          (2 if f16 else self.parts[0][0].g_f32))
        a.flops = float(getattr(self, 'flops', 0.0))
        b.op = _lib.OP_WGRAD_REDUCE
        b.p[0], b.i[0], b.p[2], b.f[0] = self.r_dev.data_ptr(), self.n_red, grad_ptr, scale
        b.i[1] = 1 if self.nsplit <= 4 else 0   # few_splits (9 taps): one reduce workgroup per output channel
        gs = getattr(self, 'g_scale', 0.0)
        b.f[1] = 1.0 / gs if (f16 and gs) else 0.0
        self.workspace.register(a, b)
        return [a, b]


class Workspace:
    """Grow-only float workspace shared by all wgrad groups of a plan (launches are stream ordered)."""

    def __init__(self, device):
        self.device = device
        self.need = 0
        self.buf = None
        self.pending = []

    def reserve(self, n):
        self.need = max(self.need, n)

    def register(self, *ops):
        self.pending.extend(ops)

    def finalize(self):
        self.buf = torch.zeros(max(self.need, 1), dtype=torch.float32, device=self.device)
        for o in self.pending:
            o.p[1] = self.buf.data_ptr()
        self.pending = []


def ensure_runtime_ready():
    """One-time device probe (selects the wgrad gather mode); must run outside graph capture."""
    L = _lib.lib()
    if not torch.cuda.is_available():
        raise _lib.DasrHipError('dasr_amd needs a ROCm GPU (MI355X); there is no CPU path')
    global _PROBED
    if not _PROBED:
        rc = L.dasr_probe_tr16(_stream())
        if rc < 0:
            raise _lib.DasrHipError('dasr_probe_tr16 failed: %d' % rc)
        _PROBED = True
        global TR16_OK
        TR16_OK = rc
        if rc != 1:   # every gfx950 has ds_read_b64_tr_b16; a device without it is not an MI355X
            raise _lib.DasrHipError('dasr_probe_tr16: this device has no transposing LDS reads (not gfx950); the weight-gradient kernels need them')
    return TR16_OK


_PROBED = False
TR16_OK = None
