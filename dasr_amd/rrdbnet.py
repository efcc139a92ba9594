"""RRDBNet generator (reference: codes/SRN/models/modules/architecture.py:174-205, block.py:254-309,854-861)
as recorded op lists over the MI355X kernels.

Numerics recipe ("recipe A"): the dense-block convs (92% of the FLOPs, damped by the
0.2 residual scales) run with bf16 operands / fp32 accumulate; the six convs that sit on the residual
stream (fea_conv, LR_conv, the two upconvs, HR_conv0/1) run in split-bf16 (prec 3, ~fp32) on fp32
activations.  The residual stream itself is kept in fp32; dense-slab channels are stored in bf16.
"""
import os
from collections import OrderedDict

import torch

from . import _lib
from .engine import (BTensor, ParamStore, PackRegistry, OpList, WgradGroup, WgradGroup3, Workspace, conv_op, ceil_div, SLOPE, NULL_T, ConvChain)
from ._lib import Op, Tensor

GC = 32  # growth channels are hard-wired to 32 in the reference (architecture.py:183)


def rrdbnet_param_spec(in_nc, out_nc, nf, nb, upsample_mode='upconv'):
    """state_dict keys / shapes of the reference RRDBNet.  upsample_mode 'upconv' (what define_G selects,
    networks.py:96-99): nearest x2 -> conv nf->nf -> LeakyReLU at model.{2,3,4} / {5,6,7}; 'pixelshuffle' (block.py:838-851): conv nf->4nf
    -> PixelShuffle(2) -> LeakyReLU, convs at model.2 / model.5 (checkpoints of the two modes are not interchangeable)."""
    spec = [('model.0.weight', (nf, in_nc, 3, 3)), ('model.0.bias', (nf,))]
    for i in range(nb):
        for r in (1, 2, 3):
            for j in range(1, 6):
                cin = nf + (j - 1) * GC
                cout = GC if j < 5 else nf
                p = 'model.1.sub.%d.RDB%d.conv%d.0.' % (i, r, j)
                spec += [(p + 'weight', (cout, cin, 3, 3)), (p + 'bias', (cout,))]
    spec += [('model.1.sub.%d.weight' % nb, (nf, nf, 3, 3)), ('model.1.sub.%d.bias' % nb, (nf,))]
    ups = ((3, nf), (6, nf)) if upsample_mode == 'upconv' else ((2, 4 * nf), (5, 4 * nf))
    for idx, co in ups + ((8, nf), (10, out_nc)):
        spec += [('model.%d.weight' % idx, (co, nf, 3, 3)), ('model.%d.bias' % idx, (co,))]
    return spec


class RRDBNetHIP:
    def __init__(self, in_nc=3, out_nc=3, nf=64, nb=23, upscale=4, device='cuda', rdb_prec=None, stream_prec=3, upsample_mode='upconv'):
        assert upscale == 4 and nf % 32 == 0 and in_nc <= 16 and out_nc <= 16
        if upsample_mode not in ('upconv', 'pixelshuffle'):
            raise NotImplementedError('upsample mode [{:s}] is not found'.format(str(upsample_mode)))   # architecture.py:190
        self.in_nc, self.out_nc, self.nf, self.nb, self.upsample_mode = in_nc, out_nc, nf, nb, upsample_mode
        self.device = torch.device(device)
        self.params = ParamStore(rrdbnet_param_spec(in_nc, out_nc, nf, nb, upsample_mode), self.device)
        # dense-block operand format: 1 = bf16 (the north_star's dtype, default), 2 = f16 STORAGE of the dense slabs and their gradients (11-bit operands,
        # gradients pre-scaled by a power of two: round 4, DASR_RDB_PREC=2 or rdb_prec=2).  Same MFMA rate; 8x finer operand rounding: on the
        # ill-conditioned BatchNorm source-discriminator step the G gradients go from 1.5e-2 to < 1e-2 of the fp32 reference (profiles/r04_bn_probe.txt),
        # which is why DASR_Model selects it for `which_model_pairD: discriminator_vgg_128`.  Range: activations beyond 65504 overflow (flagged by Adam).
        if rdb_prec is None:
            rdb_prec = int(os.environ.get('DASR_RDB_PREC', '1'))
        if rdb_prec not in (1, 2):
            raise ValueError('rdb_prec / DASR_RDB_PREC must be 1 (bf16) or 2 (f16 storage)')
        self.rdb_prec, self.stream_prec = rdb_prec, stream_prec
        self.rdb_f16 = rdb_prec == 2
        # DASR_CHAIN (default 1): the trunk's dense-block convs, forward and data gradient, as persistent chained launches where the batch fills the chip
        # exactly (chain_ok; _Plan._build_forward).  bf16 storage only: the f16 path re-patches scale factors of recorded ops (TrunkStore.set_gscale_from)
        self.chain = os.environ.get('DASR_CHAIN', '1') == '1' and not self.rdb_f16
        # DASR_CHAIN_FORM: 'is' (round 6) = the chained launches in their input-stationary form (dasr_rdb_chain: every slab chunk staged once per dense block,
        # one 8-wave workgroup per CU owning N * tiles / 256 tiles); 'layer' = round 4's layer-by-layer form (dasr_conv_chain, exactly 512 tiles per launch)
        self.chain_form = os.environ.get('DASR_CHAIN_FORM', 'auto')   # 'auto': the layer form where it fits, else the input-stationary form (chain_choice)
        assert self.chain_form in ('auto', 'is', 'layer')
        # ONE error word for every chained launch of this network (all plans): non-zero = a neighbour wait gave up, the step's results are not valid.
        # The optimisers that depend on this generator take it as their gate (AdamHIP(gate=...): such a step never reaches the weights); the trainers
        # read it where they synchronise anyway (log interval, checkpoints) and raise (check_chain).
        self.chain_err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.hr_prec = int(os.environ.get('DASR_HR_PREC', '2')) if stream_prec == 3 else stream_prec
        # hr_prec 2 (default): f16 STORAGE of the HR tail (u1, u2, h0 and their gradients live in HBM as f16): the consumers run on the LDS-DMA
        # dense-conv kernel / the grouped wgrad kernel with the f16 MFMA, and the HR tensors cost half the bytes.  DASR_HR_PREC=3 (numerics
        # switch): split-bf16 on f32 tensors (sub-pixel form of the upconvs), the fall-back when f16's range is not enough.
        if self.hr_prec not in (2, 3):
            raise ValueError('DASR_HR_PREC must be 2 (f16 storage) or 3 (split-bf16 on f32 tensors)')
        self.hr_f16s = self.hr_prec == 2
        self.ps = upsample_mode == 'pixelshuffle'
        if self.ps and not self.hr_f16s:
            raise NotImplementedError('the PixelShuffle upsampler is built on the f16-storage HR tail (unset DASR_HR_PREC)')
        self.pack = PackRegistry(self.params)
        self._register_packs()
        self.pack.finalize()
        self.plans = {}

    # ---- packed weights ----------------------------------------------------------------------------
    def _seg_fwd(self, key, cout, cin):
        return (self.params.off(key), cout, cin, 0, cin, 0, 0)

    def _register_packs(self):
        nf, P, sp = self.nf, self.params, self.stream_prec
        mt_nf = 2 if (nf >= 64 and self.rdb_prec in (1, 2)) else 1
        # HR tail (the two upconvs, HR_conv0, HR_conv1: 7.6 % of the FLOPs but a fifth of the step in split-bf16): f16 operands, ONE
        # MFMA pass.  Measured on the oracle with emulated operand rounding (oracle/precision_probe.py, nf64 nb23): SR output 2.4e-5,
        # HR-tail weight gradients 5e-4 normwise -- the tolerances are 1e-3 / 1e-2.  fea_conv and LR_conv feed the residual stream of
        # the whole trunk and stay split-bf16.  DASR_HR_PREC=3 restores split-bf16 everywhere.
        hp = self.hr_prec
        hmt = 1
        self.pk = {}
        # stream convs, forward (prec 3 -> mt 1)
        self.pk['fea'] = self.pack.add(nf, 16, 9, 1, sp, [(P.off('model.0.weight'), nf, self.in_nc, 0, self.in_nc, 0, 0)])
        lr = 'model.1.sub.%d.weight' % self.nb
        for name, key, cout in (('lr', lr, nf), ('up1', 'model.3.weight', nf), ('up2', 'model.6.weight', nf),
                                ('hr0', 'model.8.weight', nf), ('hr1', 'model.10.weight', self.out_nc)):
            if self.ps and name in ('up1', 'up2'):
                continue
            pr = sp if name == 'lr' else hp
            if self.hr_f16s and name != 'lr':
                continue   # f16 storage: the HR-tail packs are registered below (direct 3x3 forms on 16-bit tensors)
            self.pk[name] = self.pack.add(cout, nf, 9, hmt if (pr == 2 and cout % 64 == 0) else 1, pr, [self._seg_fwd(key, cout, nf)])
            # data-gradient: transposed + tap-flipped; packed cin = forward cout (padded to 16)
            cin_b = ceil_div(cout, 16) * 16
            self.pk[name + '_b'] = self.pack.add(nf, cin_b, 9, hmt if pr == 2 else 1, pr, [(P.off(key), cout, nf, 0, cout, 0, 1)])
        if self.hr_f16s:   # direct 3x3 forms on 16-bit tensors (nearest x2 folded into the DMA addresses of the dense-conv kernel)
            mtH = 2 if nf % 64 == 0 else 1
            ups = (('up1', 'model.2.weight', 4 * nf), ('up2', 'model.5.weight', 4 * nf)) if self.ps else (('up1', 'model.3.weight', nf), ('up2', 'model.6.weight', nf))
            for name, key, co in ups + (('hr0', 'model.8.weight', nf),):
                self.pk[name] = self.pack.add(co, nf, 9, mtH, 2, [self._seg_fwd(key, co, nf)])
                self.pk[name + '_b'] = self.pack.add(nf, co, 9, mtH, 2, [(P.off(key), co, nf, 0, co, 0, 1)])
            self.pk['hr1'] = self.pack.add(self.out_nc, nf, 9, 1, 2, [self._seg_fwd('model.10.weight', self.out_nc, nf)])
            self.pk['hr1_b'] = self.pack.add(nf, 16, 9, mtH, 2, [(P.off('model.10.weight'), self.out_nc, nf, 0, self.out_nc, 0, 1)])
        # sub-pixel form of nearest-x2 + 3x3 (upconv_blcok, block.py:854-861): output parity (py, px) is a 2x2 convolution of the
        # LOW-resolution input whose taps are sums of the 3x3 taps that land on the same source pixel -- 16 instead of 36 MACs per
        # input pixel, channel pair and 2x2 output block.  Row taps per parity: py=0 reads rows (i-1, i) with (w0, w1+w2), py=1 rows
        # (i, i+1) with (w0+w1, w2); same along x.  The data gradient is the transpose: per parity a 2x2 conv of that parity's
        # sub-grid of the output gradient with the tap order reversed, summed over the four parities.
        self.subpixel = not self.hr_f16s   # f32-tensor tail (DASR_HR_PREC=3): the upconvs run in their sub-pixel form
        rows = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}   # parity -> source taps of packed tap a = 0, 1
        for name, key in (() if self.hr_f16s else (('up1', 'model.3.weight'), ('up2', 'model.6.weight'))):
            for py in (0, 1):
                for px in (0, 1):
                    fw = [sum(1 << (ky * 3 + kx) for ky in rows[py][a] for kx in rows[px][b]) for a in (0, 1) for b in (0, 1)]
                    bw = [sum(1 << (ky * 3 + kx) for ky in rows[py][1 - a] for kx in rows[px][1 - b]) for a in (0, 1) for b in (0, 1)]
                    self.pk[(name, py, px)] = self.pack.add(nf, nf, 4, hmt, hp, [self._seg_fwd(key, nf, nf)], tapmap=[0, 0, 0, 0], src_ntaps=9, tapmasks=fw)
                    self.pk[(name + '_b', py, px)] = self.pack.add(nf, nf, 4, hmt, hp, [(P.off(key), nf, nf, 0, nf, 0, 1)], tapmap=[0, 0, 0, 0], src_ntaps=9,
                                                                   tapmasks=bw)
        # dense blocks
        for i in range(self.nb):
            for r in (1, 2, 3):
                pre = 'model.1.sub.%d.RDB%d.conv' % (i, r)
                for j in range(1, 6):
                    cin = nf + (j - 1) * GC
                    cout = GC if j < 5 else nf
                    self.pk[(i, r, j)] = self.pack.add(cout, cin, 9, mt_nf if j == 5 else 1, self.rdb_prec,
                                                       [self._seg_fwd('%s%d.0.weight' % (pre, j), cout, cin)])
                # backward convs: k = 4..1 produce g_k (32 ch), k = 0 produces g_x (nf ch).
                # gslab' channel order: [g5 (nf) | g4 | g3 | g2 | g1]
                for k in range(4, -1, -1):
                    segs = []
                    for j in range(5, k, -1):
                        cout_j = GC if j < 5 else nf
                        cin_j = nf + (j - 1) * GC
                        start = 0 if j == 5 else nf + (4 - j) * GC
                        src_c0 = 0 if k == 0 else nf + (k - 1) * GC
                        segs.append((P.off('%s%d.0.weight' % (pre, j)), cout_j, cin_j, start, cout_j, src_c0, 1))
                    cin_b = nf + (4 - k) * GC
                    cout_b = nf if k == 0 else GC
                    self.pk[(i, r, 'b', k)] = self.pack.add(cout_b, cin_b, 9, mt_nf if k == 0 else 1, self.rdb_prec, segs)

    def repack(self):
        self.pack.run()

    def chain_ok(self, N, h, w):
        """a training plan of this shape runs its trunk as chained launches (dasr_conv_chain / dasr_rdb_chain): chain_split(N, h, w) > 0"""
        return self.chain_split(N, h, w) > 0

    def chain_split(self, N, h, w):
        """number of chained launches per direction the trunk of a training plan of this shape runs as, 0 = one launch per conv (see chain_choice)"""
        return self.chain_choice(N, h, w)[1]

    @staticmethod
    def is_geometry(N, T):
        """(workgroups per XCD, tiles per workgroup) of an input-stationary chained launch over N images of T tiles, or None (mirrors dasr_rdb_chain, csrc/conv.hip)"""
        if N % 8 or T <= 0 or T > 32:
            return None
        per_xcd = N * T // 8
        for q in range(32 - 32 % T, T - 1, -T):
            if per_xcd % q == 0 and per_xcd // q <= 8:
                return q, per_xcd // q
        return None

    @staticmethod
    def is_launch(N, h, w):
        """(tile rows, workgroups per XCD, tiles per workgroup) of the input-stationary chained launch over N images of h x w, or None.  Tiles of 16, 8 or 4 rows of 32
        pixels: the height that minimises (tiles per workgroup) x (measured chain time of one tile at that height, 4.3 / 3.0 / 2.4 ms), ties to the taller --
        mirrors dasr_rdb_chain (csrc/conv.hip)"""
        best, cost = None, None
        for th, c in ((16, 43), (8, 30), (4, 24)):
            g = RRDBNetHIP.is_geometry(N, ceil_div(h, th) * ceil_div(w, 32))
            if g is not None and (cost is None or g[1] * c < cost):
                best, cost = (th,) + g, g[1] * c
        return best

    def chain_choice(self, N, h, w):
        """(form, k, why): how the trunk of a training plan of this shape runs.
        form 'layer', k >= 1: round 4's layer-by-layer chained launches (dasr_conv_chain).  A launch needs whole images per XCD and exactly 512 tiles: N images of T
            tiles qualify when N T = 512 k and N / k is a multiple of 8 -- the plan then runs k launches BACK TO BACK over image ranges of N / k (k = 1 at configs[1];
            k = 2 for configs[2]'s 32 crops of 128 x 128: at batch 32 the per-layer launches run a dense block in 274 + 287 us, two chained half-batches in 2 x (131 + 129) us).
            (Round 5 also built a form whose workgroups own several tiles -- bit-identical, but 88 ms per GAN step against 72 ms; -DDASR_BENCH library only.)
        form 'is', k = 1: round 6's input-stationary chained launch (dasr_rdb_chain, csrc/rdb_is.h): every slab chunk staged once per dense block, 8 q <= 256 workgroups of
            8 waves (one per CU) owning up to 8 tiles each; whole images per XCD (N % 8 == 0) and every tile of an image in flight at once (is_geometry).
            Measured (profiles/r06_is_chain.txt): level with the layer form where both apply (31.3 vs 30.6 ms at 16 x 128^2), 6-23 % faster than one launch per conv where
            only it applies (batch 8 / 24 of 128^2, 16 x 64 x 128, 32 x 64^2) -- so DASR_CHAIN_FORM=auto (default) takes 'layer' where it fits and 'is' otherwise.
        form None, k = 0: one launch per conv; `why` names the clause that refused both forms (logged once per plan)."""
        T = ceil_div(h, 16) * ceil_div(w, 32)
        ntiles = N * T
        if not self.chain:
            return None, 0, 'chained launches are off (DASR_CHAIN=0 or f16 dense blocks)'
        if getattr(self, 'debug_taps', ()):
            return None, 0, 'debug taps between the layers'
        from . import dist as _dist
        if _dist.SHARED_DEVICE:   # another rank of this job runs on the same GPU (gloo test set-up): the launch would not have the chip to itself
            return None, 0, 'another rank of this job shares the device'
        if not hasattr(self, '_cus'):   # a chained launch fills a whole MI355X (8 XCDs x 32 CUs) exactly; a partitioned device (CPX / DPX) has fewer CUs
            self._cus = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == 'cuda' else 0
        if self._cus != 256:
            return None, 0, 'the device does not expose 256 CUs (partitioned, or no GPU)'
        why = []
        if self.chain_form in ('auto', 'layer'):
            k = ntiles // 512
            kmax = int(os.environ.get('DASR_CHAIN_SPLIT', '4'))   # DASR_CHAIN_SPLIT=1: exact fit only (A/B)
            if k >= 1 and k <= kmax and ntiles == 512 * k and N % k == 0 and (N // k) % 8 == 0:
                return 'layer', k, ''
            why.append('layer form: %d tiles are not 512 k (k <= %d) with N / k a multiple of 8' % (ntiles, kmax))
        if self.chain_form in ('auto', 'is'):
            if RRDBNetHIP.is_launch(N, h, w) is not None and self.nf == 64:
                return 'is', 1, ''
            why.append('input-stationary form: needs nf 64, N %% 8 == 0 and q <= 32 workgroups per XCD with q a multiple of the %d tiles per image, q dividing the %s tiles per XCD, '
                       'at most 8 tiles per workgroup' % (T, ('%d' % (ntiles // 8)) if N % 8 == 0 else 'N x tiles / 8'))
        return None, 0, '; '.join(why)

    # ---- plan ---------------------------------------------------------------------------------------
    def plan(self, N, h, w, replica=0, store=None, n0=0):
        """replica > 0: an independent set of buffers for a sub-batch processed concurrently on another stream; its
        weight gradients go to a private flat buffer (plan.grad) that the trainer adds to params.grad.
        store / n0: the dense-block slabs of this plan are images [n0, n0 + N) of a shared TrunkStore (sub-batch replicas of one batch): the
        dense-block weight gradients are then NOT part of the plan's backward list but of store.phase (one launch over the whole batch)."""
        key = (N, h, w, replica, id(store) if store is not None else 0, n0)
        if key not in self.plans:
            self.plans[key] = _Plan(self, N, h, w, replica, store=store, n0=n0)
        return self.plans[key]

    # dense-block weight gradients run as a separate phase AFTER the data-gradient chain (TrunkStore.phase): grouped launches over the whole batch
    # with the chip to themselves, instead of one launch per RRDB interleaved with (and, under two sub-batch streams, competing with) the
    # data-gradient convs (round 2's schedule, -1.2 ms; removed in round 4).  Costs one gradient slab per RDB (3 nb x 50 MB at batch 8 x 128^2).
    defer_wgrad = True

    def trunk_store(self, N, h, w):
        key = (N, h, w)
        st = self.__dict__.setdefault('_stores', {})
        if key not in st:
            st[key] = TrunkStore(self, N, h, w)
        return st[key]

    # convenience API used by the trainers / tests -------------------------------------------------------
    def state_dict(self):
        return self.params.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.params.load_state_dict(sd, strict)
        self.repack()

    INFER_CACHE = 2   # forward-only plans kept (validation / test images come in many sizes; a plan of a big image is GBs)

    def forward(self, x):
        """Inference: x NCHW fp32 CUDA tensor -> SR NCHW fp32 CUDA tensor.  Uses a FORWARD-ONLY plan (two dense slabs reused by all RDBs,
        no gradient buffers, no wgrad workspace: the reference frees activations under no_grad too, SR_model.py:87-93) from a small LRU
        cache, so evaluating many differently sized images does not accumulate training-sized plans."""
        N, _, h, w = x.shape
        key = (N, h, w)
        cache = self.__dict__.setdefault('infer_plans', OrderedDict())
        p = cache.pop(key, None)
        if p is None:
            while len(cache) >= self.INFER_CACHE:
                cache.popitem(last=False)
            p = _Plan(self, N, h, w, inference=True)
        cache[key] = p
        p.set_input(x)
        p.fwd.run()
        return p.read_output()


def _sched(kind, handle):
    o = Op()
    o.op = kind
    o.p[0] = handle
    return o


def _sub_batch_conv(o, n0, n, N):
    """copy of a dense-block conv Op restricted to images [n0, n0 + n) of its N: every tensor view moves by n0 images (the chained launches of a batch that is a
    multiple of the 512-tile fit run one sub-batch after the other, RRDBNetHIP.chain_split)"""
    import ctypes as C
    q = Op()
    C.memmove(C.addressof(q), C.addressof(o), C.sizeof(Op))
    c = q.conv
    assert c.N == N and not c.res1_lo and not c.in_wrap and not c.out16_lo
    for name, esz in (('inp', 4 if c.in_f32 else 2), ('mask', 4 if c.mask_f32 else 2), ('res1', 4), ('res2', 4), ('out_f32', 4), ('out_bf16', 2)):
        t = getattr(c, name)
        if t.p:
            t.p = t.p + n0 * t.n_stride * esz
    c.N = n
    q.flops = o.flops * n / N
    return q


def rdb_wgrad_parts(grp, nf, pre, P, Gs, S, h, w, N):
    """parts of the 12-wave weight-gradient kernel for the 5 convs of one dense block: one part per 64-channel block of the forward slab S x up
    to three 32-oc tiles of the gradient slab Gs (= every conv that consumes those channels).  Returns (number of parts, algorithmic FLOPs)."""
    gt = []  # gslab' oc tiles in order: conv5 (nf/32 tiles), conv4, conv3, conv2, conv1
    for j in (5, 4, 3, 2, 1):
        cout_j = nf if j == 5 else GC
        for oc0 in range(0, cout_j, 32):
            gt.append(dict(j=j, oc0=oc0, cout=cout_j, cin=nf + (j - 1) * GC))
    n = 0
    for c0 in range(0, nf + 4 * GC, 64):
        need = [t for t in gt if t['cin'] > c0]
        blk_ch = min(64, nf + 4 * GC - c0)
        for k0 in range(0, len(need), 3):
            sub = need[k0:k0 + 3]
            tiles = []
            for t in sub:
                wkey = '%s%d.0.' % (pre, t['j'])
                tiles.append(dict(dst_w_off=P.off(wkey + 'weight'), dst_b_off=P.off(wkey + 'bias') if c0 == 0 else None,
                                  cout=t['cout'], cin=t['cin'], oc0=t['oc0'], c0=c0,
                                  n_ctiles=min(2, ceil_div(min(t['cin'] - c0, 64), 32))))
            grp.add_block(Gs.view(32 * k0), 2 * len(sub), S.view(c0), blk_ch // 16, ceil_div(blk_ch, 32),
                          h, w, h, w, N, tiles, want_bias=(c0 == 0))
            n += 1
    return n, 2.0 * N * h * w * 9 * sum((nf + (j - 1) * GC) * (GC if j < 5 else nf) for j in range(1, 6))


class TrunkStore:
    """Forward slabs and gradient slabs of ALL dense blocks for a whole batch of N images (one allocation per RDB; sub-batch replicas work on
    image ranges of it), and the deferred weight-gradient phase over them: after the data-gradient chain has filled every gradient slab, the
    weight gradients of the 3 nb dense blocks are computed by a few grouped launches (4 RRDBs per launch: every
    launch has ~240 workgroups of 12 waves, one per CU, nothing co-resident; four pixel splits instead of sixteen: a quarter of round 2's
    fp32 partial-sum traffic of 24.8 MB written and re-read per RRDB)."""

    def __init__(self, net, N, h, w):
        self.net, self.N, self.h, self.w = net, N, h, w
        dev, nf, nb = net.device, net.nf, net.nb
        sc = nf + 4 * GC
        mk = lambda: torch.zeros((N, ceil_div(sc, 16), h, w, 16), dtype=torch.float16 if net.rdb_f16 else torch.bfloat16, device=dev)
        # f16 dense blocks: every gradient slab holds gscale * dL/d(.); ONE power of two for the whole batch (the replicas of a step share the slabs and
        # the grouped weight-gradient launches), sized like the HR tail's: dL/d(trunk) of a mean loss is ~1 / (N 3 16 h w), brought to ~2^-3
        import math
        # The magnitude of dL/d(trunk) depends on the weights (kaiming x 0.1 at init: ~1e-5 of dL/dSR; a trained net: orders of magnitude more), so the
        # scale is CALIBRATED from data: the trainers measure max |dL/d(trunk output)| behind the HR tail (calibrate_due / set_gscale: on the first
        # step of a plan and every CALIB_EVERY steps after it, one host sync) and every op that carries the scale is patched in place.  f16 has
        # 30 normal binades; the calibration puts the largest slab entry at ~1, so a drift of 2^+-8 between two calibrations is harmless, and
        # an overflow reaches Adam's non-finite guard.
        self.gscale = 1.0
        self._scaled_ops, self._plans, self.steps_since_calib = [], [], None
        self.slab_t = [mk() for _ in range(3 * nb)]
        self.gslab_t = [mk() for _ in range(3 * nb)]
        self.sc = sc
        self.ws = Workspace(dev)
        self._build_phase()
        self.ws.finalize()

    def slab(self, ridx, n0=0, N=None):
        N = self.N if N is None else N
        return BTensor.wrap(self.slab_t[ridx][n0:n0 + N], self.sc, False)

    def gslab(self, ridx, n0=0, N=None):
        N = self.N if N is None else N
        return BTensor.wrap(self.gslab_t[ridx][n0:n0 + N], self.sc, False)

    def _build_phase(self):
        net, N, h, w = self.net, self.N, self.h, self.w
        nf, nb, P = net.nf, net.nb, net.params
        target = 256   # workgroups per launch: one 16-wave workgroup per CU
        gmax = 4       # RRDBs per launch: 4 measured best (16: -1 %, 1: -1.5 %; profiles/r03_wgrad_ablation.txt)
        self.phase = OpList()
        self.groups = []   # (first op, end op, lo, hi): ops [first, end) complete params.grad[lo:hi]; descending parameter order
        hi_rrdb = nb
        ppr = None
        while hi_rrdb > 0:
            if ppr is None:   # parts per RRDB: count on a scratch group
                tmp = WgradGroup3()
                ppu, _ = rdb_wgrad_parts(tmp, nf, 'model.1.sub.0.RDB1.conv', P, self.gslab(0), self.slab(0), h, w, N)
                ppr = 3 * ppu
            G = 1
            while G * 2 <= hi_rrdb and G * 2 <= gmax and G * 2 * ppr <= target:
                G *= 2
            lo_rrdb = hi_rrdb - G
            grp = WgradGroup3()
            grp.flops = 0.0
            for i in range(hi_rrdb - 1, lo_rrdb - 1, -1):
                for r in (3, 2, 1):
                    ridx = 3 * i + (r - 1)
                    _, fl = rdb_wgrad_parts(grp, nf, 'model.1.sub.%d.RDB%d.conv' % (i, r), P, self.gslab(ridx), self.slab(ridx), h, w, N)
                    grp.flops += fl
            if net.rdb_f16:
                grp.f16, grp.g_scale = True, self.gscale
            grp.finalize(self.ws, net.device, target_wgs=target, ppu=ppu)
            first = len(self.phase.ops)
            for o in grp.ops(P.grad.data_ptr()):
                self.phase.add(o)
            self.phase.keep.append(grp)
            lo = P.off('model.1.sub.%d.RDB1.conv1.0.weight' % lo_rrdb)
            hi = P.off('model.1.sub.%d.RDB1.conv1.0.weight' % hi_rrdb) if hi_rrdb < nb else P.off('model.1.sub.%d.weight' % nb)
            self.groups.append((first, len(self.phase.ops), lo, hi))
            hi_rrdb = lo_rrdb
        self.phase.tag(5)

    CALIB_EVERY = 256

    def register_scaled(self, op, base_gamma, has_alpha):
        """conv op whose 16-bit output carries the gradient scale (gamma = base_gamma * gscale) and, with has_alpha, whose accumulator is un-scaled
        for the fp32 gradient stream (alpha = 1 / gscale)"""
        self._scaled_ops.append((op, float(base_gamma), bool(has_alpha)))

    def calibrate_due(self):
        """f16 dense blocks: True when the gradient scale has to be (re-)measured before this step's dense-block backward"""
        if not self.net.rdb_f16:
            return False
        due = self.steps_since_calib is None or self.steps_since_calib >= self.CALIB_EVERY
        if not due:
            self.steps_since_calib += 1
        return due

    def set_gscale_from(self, g_t0_absmax):
        """g_t0_absmax: max |dL/d(trunk output)| (host float).  The first planes of the first gradient slab hold 0.04 * gscale * that: brought to ~1."""
        import math
        a = 0.04 * float(g_t0_absmax)
        s = 1.0 if not (a > 0.0 and math.isfinite(a)) else float(2.0 ** max(-60, min(60, -int(math.ceil(math.log2(a))))))
        self.steps_since_calib = 0
        if s == self.gscale:
            return s
        self.gscale = s
        for o, bg, has_alpha in self._scaled_ops:
            o.conv.gamma = bg * s
            if has_alpha:
                o.conv.alpha = 1.0 / s
        for o in self.phase.ops:
            if o.op == _lib.OP_WGRAD_REDUCE:
                o.f[1] = 1.0 / s
        self.phase._arr = None
        for pl in self._plans:   # every recorded list that holds copies of the patched ops
            pl.bwd._arr = None
            pl._segments = None
            if hasattr(pl, 'whole_step'):
                pl.whole_step._arr = None
        return s

    def set_grad_scale(self, scale):
        changed = False
        for o in self.phase.ops:
            if o.op == _lib.OP_WGRAD_REDUCE and o.f[0] != scale:
                o.f[0] = scale
                changed = True
        if changed:
            self.phase._arr = None
        return changed


class _Plan:
    """Buffers + recorded forward / backward op lists for one (N, h, w)."""

    def __init__(self, net, N, h, w, replica=0, inference=False, store=None, n0=0):
        self.net, self.N, self.h, self.w = net, N, h, w
        dev, nf, nb = net.device, net.nf, net.nb
        self.inference = inference
        self.replica = replica
        # deferred dense-block weight gradients (TrunkStore): shared = the store belongs to a group of sub-batch replicas and the trainer runs
        # its phase after all of them; otherwise the plan owns a store of its own batch and the phase is part of plan.bwd
        self.defer = not inference
        self.shared_store = store is not None
        self.store = store if store is not None else (TrunkStore(net, N, h, w) if self.defer else None)
        self.n0 = n0
        self.grad = net.params.grad if (replica == 0 or inference) else torch.zeros_like(net.params.grad)
        # power-of-two pre-scale of the HR-tail gradients before their f16 rounding (prec 2): dL/dSR of a mean loss is ~1 / (N 3 H W) ~ 1e-7,
        # far below f16's normal range; scaled to ~2^-3.  Exact (power of two), undone in the conv epilogue / the wgrad reduction.
        import math
        self.gscale = float(2.0 ** max(0, int(math.floor(math.log2(max(1, N * 3 * 16 * h * w)))) - 3))
        P, pack, pk = net.params, net.pack, net.pk
        H2, W2, H4, W4 = 2 * h, 2 * w, 4 * h, 4 * w
        sc = nf + 4 * GC  # dense-slab channels
        B = lambda C_, H, W, f32: BTensor(N, C_, H, W, f32, dev)
        self.x_nchw = torch.zeros((N, net.in_nc, h, w), dtype=torch.float32, device=dev)
        self.sr_nchw = torch.zeros((N, net.out_nc, H4, W4), dtype=torch.float32, device=dev)
        self.x_in = B(16, h, w, True)
        self.fea = B(nf, h, w, True)
        if inference:   # nothing is kept for a backward pass: two slabs alternate through the 3 nb dense blocks
            ab = [BTensor(N, sc, h, w, False, dev, f16=net.rdb_f16) for _ in range(2)]
            self.slabs = [ab[i & 1] for i in range(3 * nb)]
        else:
            self.slabs = [self.store.slab(r, n0, N) for r in range(3 * nb)]
        self.stream = [B(nf, h, w, True) for _ in range(4)]
        self.t0 = B(nf, h, w, True)
        hs = net.hr_f16s
        Bh = (lambda C_, H, W: BTensor(N, C_, H, W, False, dev, f16=True)) if hs else (lambda C_, H, W: B(C_, H, W, True))
        self.t0h = Bh(nf, h, w) if hs else None   # f16 shadow of the trunk output (input of upconv1)
        self.u1 = Bh(nf, H2, W2)
        self.u2 = Bh(nf, H4, W4)
        if net.ps:   # conv outputs before the shuffle (already activated) and their gradients
            self.ps1, self.ps2 = Bh(4 * nf, h, w), Bh(4 * nf, H2, W2)
            if not inference:
                self.g_ps1, self.g_ps2 = Bh(4 * nf, h, w), Bh(4 * nf, H2, W2)
        self.h0 = Bh(nf, H4, W4)
        self.sr = B(16, H4, W4, True)
        if inference:
            self._build_forward()
            return
        # backward
        self.g_sr = B(16, H4, W4, True)
        self.g_sr16 = Bh(16, H4, W4) if hs else None   # dL/dSR, pre-scaled by gscale, f16
        self.g4a = Bh(nf, H4, W4)
        self.g4b = Bh(nf, H4, W4)
        self.g2a = Bh(nf, H2, W2)
        self.g2b = Bh(nf, H2, W2)
        self.g_t0 = B(nf, h, w, True)
        self.gstream = [B(nf, h, w, True) for _ in range(4)]
        # one gradient slab per RDB (kept for the deferred weight-gradient phase), in the order the backward chain visits them (RDB 3 nb - 1 first)
        self.n_gslab = 3 * nb
        self.gslab = [self.store.gslab(3 * nb - 1 - k, n0, N) for k in range(3 * nb)]
        self.g_fea = B(nf, h, w, True)
        self.ws = Workspace(dev)
        self._build_forward()
        self._build_backward()
        self.ws.finalize()

    def take_f16_loss_gradient(self):
        """A trainer whose ONLY term in dL/dSR is one pixel loss (SRModel) may write it straight into the f16 tensor the HR tail's backward starts from -- pre-scaled by
        the returned power of two -- instead of filling the padded fp32 image plan.g_sr (64 B per pixel for 12 B of payload) that a conversion pass then re-reads
        (0.3 ms of the configs[1] step).  Returns (view of the f16 tensor, scale) and drops the conversion from plan.bwd, or None where the backward
        does not start from an f16 tensor (fp32 HR tail).  Call once, before the plan's step lists are recorded."""
        if self.g_sr16 is None or getattr(self, '_loss16_taken', False) or hasattr(self, 'whole_step'):
            return None
        o = self.bwd.ops[0]
        assert o.op == _lib.OP_CVT_F16 and o.t[1].p == self.g_sr16.view().p
        del self.bwd.ops[0]
        self.bwd._arr = None
        self.tail_end -= 1
        self._marks = [(idx - 1, lo, hi) for idx, lo, hi in self._marks]
        self._segments = None
        self._loss16_taken = True
        return self.g_sr16.view(), self.gscale

    def check_chain(self):
        """host sync: raise if a chained launch of this plan flagged a broken neighbour wait (the results of that step are not valid)"""
        if getattr(self, 'chain', None) is not None:
            try:
                self.chain.check()
            except RuntimeError as e:
                raise RuntimeError(str(e) + '.  The chained trunk launches (dasr_conv_chain) need the GPU to themselves -- another process sharing the '
                                   'device breaks their workgroup placement; DASR_CHAIN=0 restores one launch per conv.')

    # ---- IO -----------------------------------------------------------------------------------------
    def set_input(self, x):
        self.x_nchw.copy_(x)

    def read_output(self):
        """SR output as NCHW fp32.  Training plans convert on demand (one launch when somebody looks at the images -- visuals, tests -- instead of
        a 16 x 3 x 512 x 512 layout pass in every step); inference plans have the conversion at the end of their forward list."""
        if self._out_ops is not None:
            self._out_ops.run()
        return self.sr_nchw

    def _add_output_op(self, fwd, o):
        """the blocked -> NCHW conversion of the SR output: part of the forward list for inference plans, on demand (read_output) for training plans"""
        if self.inference:
            fwd.add(o)
            self._out_ops = None
        else:
            self._out_ops = OpList()
            self._out_ops.add(o)

    # ---- forward --------------------------------------------------------------------------------------
    def _build_forward(self):
        net, N, h, w = self.net, self.N, self.h, self.w
        nf, nb, P, pack, pk = net.nf, net.nb, net.params, net.pack, net.pk
        H2, W2, H4, W4 = 2 * h, 2 * w, 4 * h, 4 * w
        ops = OpList()
        o = Op()
        o.op = _lib.OP_NCHW2B
        o.p[0], o.i[0], o.i[1], o.i[2], o.i[3] = self.x_nchw.data_ptr(), N, net.in_nc, h, w
        o.t[0], o.t[1] = self.x_in.view(), NULL_T
        ops.add(o)
        # fea_conv: fp32 stream + bf16 shadow into the first dense slab
        f16 = int(net.rdb_f16)   # dense slabs in f16 storage: every 16-bit output of the trunk is f16
        ops.add(conv_op(pack, pk['fea'], self.x_in.view(), True, 16, h, w, h, w, N, bias=P.ptr('model.0.bias'),
                        out_f32=self.fea.view(), out_bf16=self.slabs[0].view(0), out16_f16=f16))
        X = self.fea  # fp32 input of the current RDB
        free = list(self.stream)
        self.rdb_in = []
        # DASR_CHAIN=1: the dense-block convs of the trunk as ONE persistent launch (dasr_conv_chain): no kernel boundary between layers, a tile waits for
        # its neighbour tiles only before the input chunks the previous layer produced.  Needs whole images per XCD (N % 8 == 0) and every workgroup
        # resident with the chip exactly full (N * tiles == 512); the taps of the tests sit between layers and keep the per-layer launches.
        tiles = ceil_div(h, 16) * ceil_div(w, 32)
        # (a plan built as one of several CONCURRENT sub-batch replicas never chains -- a chained launch needs every workgroup slot of the device, two of
        # them at once starve each other until the neighbour waits time out)
        form, nsub, why = (None, 0, 'inference plan') if self.inference else net.chain_choice(N, h, w)
        if nsub and (self.shared_store or self.replica > 0 or getattr(net, 'concurrent_replicas', 1) > 1):
            form, nsub, why = None, 0, 'the plan is one of several concurrent sub-batch replicas'
        self.chain_form, self.chain_why = form, why
        if not self.inference:
            import logging
            logging.getLogger('base').info('RRDBNet trunk at %d x %d x %d: %s' % (N, h, w, ('%d chained launch(es) per direction, %s form' % (nsub, form)) if nsub else ('one launch per conv (%s)' % why)))
        chain = [] if nsub else None
        trunk_ops = ops if chain is None else OpList()
        main_ops, ops = ops, trunk_ops
        for i in range(nb):
            Xrrdb = X
            for r in (1, 2, 3):
                ridx = 3 * i + (r - 1)
                S = self.slabs[ridx]
                pre = 'model.1.sub.%d.RDB%d.conv' % (i, r)
                for j in range(1, 5):
                    cin = nf + (j - 1) * GC
                    ops.add(conv_op(pack, pk[(i, r, j)], S.view(0), False, cin, h, w, h, w, N, bias=P.ptr('%s%d.0.bias' % (pre, j)),
                                    act=1, out_bf16=S.view(cin), out16_f16=f16))
                last = (ridx == 3 * nb - 1)
                Y = next(b for b in free if b is not X and b is not Xrrdb)
                nxt = None if last else self.slabs[ridx + 1].view(0)
                if r < 3:
                    ops.add(conv_op(pack, pk[(i, r, 5)], S.view(0), False, nf + 4 * GC, h, w, h, w, N, bias=P.ptr(pre + '5.0.bias'),
                                    alpha=0.2, res1=X.view(), beta1=1.0, out_f32=Y.view(), out_bf16=nxt, out16_f16=f16))
                else:  # RDB3 + RRDB residual fused: 0.2*(0.2*c + x3) + x_rrdb
                    ops.add(conv_op(pack, pk[(i, r, 5)], S.view(0), False, nf + 4 * GC, h, w, h, w, N, bias=P.ptr(pre + '5.0.bias'),
                                    alpha=0.04, res1=X.view(), beta1=0.2, res2=Xrrdb.view(), beta2=1.0, out_f32=Y.view(), out_bf16=nxt, out16_f16=f16))
                X = Y
            if i in getattr(net, 'debug_taps', ()):   # tests: fp32 copy of this RRDB's output (the stream buffers rotate)
                self.taps = getattr(self, 'taps', {})
                self.taps[i] = BTensor(N, nf, h, w, True, net.device)
                o = Op()
                o.op = _lib.OP_AXPBY
                o.t[0], o.f[0], o.t[1], o.f[1] = X.view(), 1.0, NULL_T, 0.0
                o.i[0], o.i[1], o.i[2], o.i[3] = N, nf, h, w
                o.t[2], o.t[3], o.f[2] = self.taps[i].view(), NULL_T, 1.0
                ops.add(o)
        ops = main_ops
        self.chain = None
        if chain is not None:
            # dep_chunk: conv1 of an RDB reads only what the previous conv5 (or fea_conv, a kernel earlier) wrote; conv j > 1 and conv5 read GC new channels
            # (the last two 16-channel chunks) from the layer in front of them
            body = [o for o in trunk_ops.ops if o.op == _lib.OP_CONV]
            assert len(body) == len(trunk_ops.ops) == 15 * nb
            deps = [0 if k % 5 == 0 else (o.conv.cin // 16) - GC // 16 for k, o in enumerate(body)]
            per = N // nsub
            self.chains = []   # (the last conv5 has no 16-bit shadow to write: its own launch, over the whole batch; the input-stationary form runs it too)
            if self.chain_form == 'is':
                ch = ConvChain(body, deps, N, ceil_div(h, 4) * ceil_div(w, 32), net.device, err=net.chain_err, form='is')   # (flag words for the finest tile height, 4 rows)
                self.chains.append(ch)
                ops.add(ch.op())
                ops.keep.append(ch)
            else:
                for sb in range(nsub):
                    sub = body[:-1] if nsub == 1 else [_sub_batch_conv(o, sb * per, per, N) for o in body[:-1]]
                    ch = ConvChain(sub, deps[:-1], per, tiles, net.device, err=net.chain_err)
                    self.chains.append(ch)
                    ops.add(ch.op())
                    ops.keep.append(ch)
                ops.add(body[-1])
            self.chain = self.chains[0]
        self.x_last = X
        lrb = 'model.1.sub.%d.bias' % nb
        if net.hr_f16s:
            ops.add(conv_op(pack, pk['lr'], X.view(), True, nf, h, w, h, w, N, bias=P.ptr(lrb), res1=self.fea.view(), beta1=1.0,
                            out_f32=self.t0.view(), out_bf16=self.t0h.view(), out16_f16=1))
            ops.tag(1)
            if net.ps:   # pixelshuffle_block (block.py:838-851): conv nf -> 4 nf, PixelShuffle(2), LeakyReLU (applied before the shuffle: it is elementwise)
                for name, bkey, src, pre, dst, hi, wi in (('up1', 'model.2.bias', self.t0h, self.ps1, self.u1, h, w), ('up2', 'model.5.bias', self.u1, self.ps2, self.u2, H2, W2)):
                    ops.add(conv_op(pack, pk[name], src.view(), False, nf, hi, wi, hi, wi, N, bias=P.ptr(bkey), act=1, out_bf16=pre.view(), out16_f16=1))
                    o = Op()
                    o.op = _lib.OP_PIXSHUF
                    o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[1] = pre.view(), N, 4 * nf, hi, wi, dst.view()
                    ops.add(o)
            for name, bkey, src, dst, hi, wi in (() if net.ps else (('up1', 'model.3.bias', self.t0h, self.u1, h, w), ('up2', 'model.6.bias', self.u1, self.u2, H2, W2))):
                ops.add(conv_op(pack, pk[name], src.view(), False, nf, hi, wi, 2 * hi, 2 * wi, N, bias=P.ptr(bkey), ups=1, act=1, out_bf16=dst.view(),
                                out16_f16=1))
            ops.add(conv_op(pack, pk['hr0'], self.u2.view(), False, nf, H4, W4, H4, W4, N, bias=P.ptr('model.8.bias'), act=1, out_bf16=self.h0.view(),
                            out16_f16=1))
            ops.add(conv_op(pack, pk['hr1'], self.h0.view(), False, nf, H4, W4, H4, W4, N, bias=P.ptr('model.10.bias'), out_f32=self.sr.view()))
            o = Op()
            o.op = _lib.OP_B2NCHW
            o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.p[0] = self.sr.view(), N, net.out_nc, H4, W4, self.sr_nchw.data_ptr()
            self._add_output_op(ops, o)
            self.fwd = ops.tag(2)
            return
        ops.add(conv_op(pack, pk['lr'], X.view(), True, nf, h, w, h, w, N, bias=P.ptr(lrb), res1=self.fea.view(), beta1=1.0,
                        out_f32=self.t0.view()))
        ops.tag(1)
        for name, bkey, src, dst, hi, wi in (('up1', 'model.3.bias', self.t0, self.u1, h, w), ('up2', 'model.6.bias', self.u1, self.u2, H2, W2)):
            if not net.subpixel:
                ops.add(conv_op(pack, pk[name], src.view(), True, nf, hi, wi, 2 * hi, 2 * wi, N, bias=P.ptr(bkey), ups=1, act=1, out_f32=dst.view()))
                continue
            for py in (0, 1):
                for px in (0, 1):
                    ops.add(conv_op(pack, pk[(name, py, px)], src.view(), True, nf, hi, wi, hi, wi, N, bias=P.ptr(bkey), kh=2, stride=1, pad=1 - py,
                                    pad_x=1 - px, act=1, out_f32=dst.view(), out_stride=2, out_oy=py, out_ox=px, out_W=2 * wi,
                                    flops=2.0 * N * hi * wi * 9 * nf * nf))
        ops.add(conv_op(pack, pk['hr0'], self.u2.view(), True, nf, H4, W4, H4, W4, N, bias=P.ptr('model.8.bias'), act=1,
                        out_f32=self.h0.view()))
        ops.add(conv_op(pack, pk['hr1'], self.h0.view(), True, nf, H4, W4, H4, W4, N, bias=P.ptr('model.10.bias'),
                        out_f32=self.sr.view()))
        o = Op()
        o.op = _lib.OP_B2NCHW
        o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.p[0] = self.sr.view(), N, net.out_nc, H4, W4, self.sr_nchw.data_ptr()
        self._add_output_op(ops, o)
        self.fwd = ops.tag(2)

    # ---- backward (input: self.g_sr filled by a loss kernel) -----------------------------------------------
    def _wg(self, ops, conv_key, g, g_f32, inp, in_f32, cout, cin, Hin, Win, Hout, Wout, ups=0, groups=None, f16=False):
        """single-conv wgrad group"""
        P = self.net.params
        grp = WgradGroup(3, 1)
        grp.add_conv(g.view, g_f32, g.planes, inp.view, in_f32, inp.planes, cout, cin, Hin, Win, Hout, Wout, self.N,
                     P.off(conv_key + 'weight'), P.off(conv_key + 'bias'), ups=ups, f16=f16, g_scale=self.gscale)
        grp.finalize(self.ws, self.net.device)
        for o in grp.ops(self.grad.data_ptr()):
            ops.add(o)
        ops.keep.append(grp)

    def _subpixel_dgrad(self, ops, name, g_hi, g_lo, hl, wl, mask):
        """g_lo[hl x wl] = (mask') * sum over output parities of the transposed 2x2 conv of g_hi's parity sub-grid (g_hi: 2hl x 2wl).
        Forward parity (py, px) read input rows i - (1 - py) + a; the transpose reads sub-grid rows m + (1 - py) - a = m - pad' + a' with
        a' = 1 - a (tap order reversed in the pack) and pad' = py."""
        net, pack, pk, N = self.net, self.net.pack, self.net.pk, self.N
        first = True
        for py in (0, 1):
            for px in (0, 1):
                ops.add(conv_op(pack, pk[(name, py, px)], g_hi.view(), True, net.nf, hl, wl, hl, wl, N, kh=2, stride=1, pad=py, pad_x=px,
                                mask=mask.view() if mask is not None else None, mask_f32=1, slope=SLOPE,
                                res1=None if first else g_lo.view(), beta1=0.0 if first else 1.0, out_f32=g_lo.view(),
                                in_stride=2, in_oy=py, in_ox=px, in_W=2 * wl, flops=2.0 * N * hl * wl * 9 * net.nf * net.nf, in_scale=self.gscale))
                first = False

    def _event(self):
        ev = _lib.lib().dasr_event_create()
        if not ev:
            raise _lib.DasrHipError('hipEventCreate failed')
        self._events = getattr(self, '_events', [])
        self._events.append(ev)
        return ev

    def _build_backward(self):
        net, N, h, w = self.net, self.N, self.h, self.w
        nf, nb, P, pack, pk = net.nf, net.nb, net.params, net.pack, net.pk
        H2, W2, H4, W4 = 2 * h, 2 * w, 4 * h, 4 * w
        ops = OpList()
        g_h0, g_u2, g_up2 = self.g4a, self.g4b, self.g4a
        f16 = net.hr_prec == 2
        gs = self.gscale
        if net.hr_f16s:
            self._build_backward_tail_f16(ops)
        else:
            self._build_backward_tail_f32(ops, f16, gs)
        ops.tag(3)
        self.tail_end = len(ops.ops)   # ops [0, tail_end): the HR tail's backward; self.g_t0 = dL/d(trunk output) is complete behind them
        self._build_backward_trunk(ops)
        self.store._plans.append(self)

    def _wg3_target(self, nparts):
        """workgroups of a 12-wave weight-gradient launch (one per CU, nothing co-resides with them): the whole chip for a single plan, an equal
        share for each of k concurrent sub-batch replicas"""
        return 256 // max(1, getattr(self.net, 'concurrent_replicas', 1))

    def _wg3(self, ops, conv_key, g, inp, cout, cin, Hin, Win, Hout, Wout, ups=0):
        """weight gradient of one 3x3 conv on f16 tensors (g pre-scaled by gscale) with the 12-wave kernel: one part per 64-channel block of
        the input x up to three 32-oc tiles"""
        P, N = self.net.params, self.N
        grp = WgradGroup3()
        octs = list(range(0, cout, 32))
        for c0 in range(0, cin, 64):
            blk = min(64, cin - c0)
            for k0 in range(0, len(octs), 3):
                sub = octs[k0:k0 + 3]
                tiles = [dict(dst_w_off=P.off(conv_key + 'weight'), dst_b_off=P.off(conv_key + 'bias') if c0 == 0 else None, cout=cout, cin=cin,
                              oc0=oc0, c0=c0, n_ctiles=min(2, ceil_div(blk, 32))) for oc0 in sub]
                grp.add_block(g.view(sub[0]), min(2 * len(sub), g.planes - sub[0] // 16), inp.view(c0), ceil_div(blk, 16), ceil_div(blk, 32),
                              Hin, Win, Hout, Wout, N, tiles, want_bias=(c0 == 0), ups=ups)
        grp.f16, grp.g_scale = True, self.gscale
        grp.flops = 2.0 * N * Hout * Wout * 9 * cin * cout
        grp.finalize(self.ws, self.net.device, target_wgs=self._wg3_target(len(grp.parts)))
        for o in grp.ops(self.grad.data_ptr()):
            ops.add(o)
        ops.keep.append(grp)

    def _build_backward_tail_f16(self, ops):
        """HR tail in f16 storage: every gradient tensor holds gscale * dL/d(.) in f16; the last 2x2 down-sum hands dL/d(trunk output) back in
        f32, un-scaled"""
        net, N, h, w = self.net, self.N, self.h, self.w
        nf, pack, pk, gs = net.nf, net.pack, net.pk, self.gscale
        H2, W2, H4, W4 = 2 * h, 2 * w, 4 * h, 4 * w
        g_h0, g_u2, g_up2, g_u1, g_up1 = self.g4a, self.g4b, self.g4a, self.g2a, self.g2b

        def downsum(src, hl, wl, mask, dst_f32, dst_f16, out_scale):
            o = Op()
            o.op = _lib.OP_DOWNSUM_F16
            o.t[0], o.i[0], o.i[1], o.i[2], o.i[3] = src.view(), N, nf, hl, wl
            o.t[1], o.f[0], o.f[1] = (mask.view() if mask is not None else NULL_T), SLOPE, out_scale
            o.t[2], o.t[3] = (dst_f32.view() if dst_f32 is not None else NULL_T), (dst_f16.view() if dst_f16 is not None else NULL_T)
            ops.add(o)

        o = Op()
        o.op = _lib.OP_CVT_F16
        o.t[0], o.i[0], o.i[1], o.i[2], o.i[3], o.f[0], o.t[1] = self.g_sr.view(), N, net.out_nc, H4, W4, gs, self.g_sr16.view()
        ops.add(o)
        # HR_conv1
        self._wg3(ops, 'model.10.', self.g_sr16, self.h0, net.out_nc, nf, H4, W4, H4, W4)
        ops.add(conv_op(pack, pk['hr1_b'], self.g_sr16.view(), False, 16, H4, W4, H4, W4, N, mask=self.h0.view(), mask_f32=0, out_bf16=g_h0.view(),
                        out16_f16=1))
        # HR_conv0
        self._wg3(ops, 'model.8.', g_h0, self.u2, nf, nf, H4, W4, H4, W4)
        ops.add(conv_op(pack, pk['hr0_b'], g_h0.view(), False, nf, H4, W4, H4, W4, N, mask=self.u2.view(), mask_f32=0, out_bf16=g_u2.view(), out16_f16=1))
        if net.ps:
            # PixelShuffle upsamplers: un-shuffle the gradient (+ LeakyReLU' of the activated conv output), then a plain conv nf -> 4 nf backward
            # (g_u2 already carries the LeakyReLU' of u2: HR_conv0's data-gradient applied it as its mask; g_u1 comes out of a plain conv)
            for key, name, g_hi, pre, g_pre, src, g_lo, hl, wl, last in (('model.5.', 'up2', g_u2, None, self.g_ps2, self.u1, g_u1, H2, W2, False),
                                                                          ('model.2.', 'up1', g_u1, self.ps1, self.g_ps1, self.t0h, None, h, w, True)):
                o = Op()
                o.op = _lib.OP_PIXUNSHUF
                o.t[0], o.t[1], o.f[0], o.i[0], o.i[1], o.i[2], o.i[3], o.t[2] = (g_hi.view(), pre.view() if pre is not None else NULL_T, SLOPE, N, 4 * nf,
                                                                                  hl, wl, g_pre.view())
                ops.add(o)
                self._wg3(ops, key, g_pre, src, 4 * nf, nf, hl, wl, hl, wl)
                if last:   # dL/d(trunk output) leaves the f16 / scaled domain
                    ops.add(conv_op(pack, pk[name + '_b'], g_pre.view(), False, 4 * nf, hl, wl, hl, wl, N, alpha=1.0 / gs, out_f32=self.g_t0.view()))
                else:
                    ops.add(conv_op(pack, pk[name + '_b'], g_pre.view(), False, 4 * nf, hl, wl, hl, wl, N, out_bf16=g_lo.view(), out16_f16=1))
            return
        # upconv2: weight gradient on the up-sampled u1; data gradient on the 4h x 4w grid, then the 2x2 sum with LeakyReLU'(u1)
        self._wg3(ops, 'model.6.', g_u2, self.u1, nf, nf, H2, W2, H4, W4, ups=1)
        ops.add(conv_op(pack, pk['up2_b'], g_u2.view(), False, nf, H4, W4, H4, W4, N, out_bf16=g_up2.view(), out16_f16=1))
        downsum(g_up2, H2, W2, self.u1, None, g_u1, 1.0)
        # upconv1: its input is the (linear) trunk output: no mask; the sum leaves the f16 / scaled domain
        self._wg3(ops, 'model.3.', g_u1, self.t0h, nf, nf, h, w, H2, W2, ups=1)
        ops.add(conv_op(pack, pk['up1_b'], g_u1.view(), False, nf, H2, W2, H2, W2, N, out_bf16=g_up1.view(), out16_f16=1))
        downsum(g_up1, h, w, None, self.g_t0, None, 1.0 / gs)

    def _build_backward_tail_f32(self, ops, f16, gs):
        net, N, h, w = self.net, self.N, self.h, self.w
        nf, nb, P, pack, pk = net.nf, net.nb, net.params, net.pack, net.pk
        H2, W2, H4, W4 = 2 * h, 2 * w, 4 * h, 4 * w
        g_h0, g_u2, g_up2 = self.g4a, self.g4b, self.g4a
        # HR_conv1
        self._wg(ops, 'model.10.', self.g_sr, True, self.h0, True, net.out_nc, nf, H4, W4, H4, W4, f16=f16)
        ops.add(conv_op(pack, pk['hr1_b'], self.g_sr.view(), True, 16, H4, W4, H4, W4, N, mask=self.h0.view(), mask_f32=1,
                        out_f32=g_h0.view(), in_scale=gs))
        # HR_conv0
        self._wg(ops, 'model.8.', g_h0, True, self.u2, True, nf, nf, H4, W4, H4, W4, f16=f16)
        ops.add(conv_op(pack, pk['hr0_b'], g_h0.view(), True, nf, H4, W4, H4, W4, N, mask=self.u2.view(), mask_f32=1,
                        out_f32=g_u2.view(), in_scale=gs))
        # upconv2 (model.6): wgrad on the upsampled u1; dgrad at 4h x 4w then 2x2 sum (+ LeakyReLU' of u1)
        self._wg(ops, 'model.6.', g_u2, True, self.u1, True, nf, nf, H2, W2, H4, W4, ups=1, f16=f16)
        g_u1 = self.g2a
        if net.subpixel:   # four parity sub-grids of g_u2 -> low-res gradient, LeakyReLU' of u1 applied to every (linear) partial
            self._subpixel_dgrad(ops, 'up2_b', g_u2, g_u1, H2, W2, mask=self.u1)
        else:
            ops.add(conv_op(pack, pk['up2_b'], g_u2.view(), True, nf, H4, W4, H4, W4, N, out_f32=g_up2.view(), in_scale=gs))
            o = Op()
            o.op = _lib.OP_DOWNSUM
            o.t[0], o.i[0], o.i[1], o.i[2], o.i[3] = g_up2.view(), N, nf, H2, W2
            o.t[1], o.i[4], o.f[0], o.t[2], o.t[3] = self.u1.view(), 1, SLOPE, g_u1.view(), NULL_T
            ops.add(o)
        # upconv1 (model.3)
        self._wg(ops, 'model.3.', g_u1, True, self.t0, True, nf, nf, h, w, H2, W2, ups=1, f16=f16)
        if net.subpixel:
            self._subpixel_dgrad(ops, 'up1_b', g_u1, self.g_t0, h, w, mask=None)
        else:
            g_up1 = self.g2b
            ops.add(conv_op(pack, pk['up1_b'], g_u1.view(), True, nf, H2, W2, H2, W2, N, out_f32=g_up1.view(), in_scale=gs))
            o = Op()
            o.op = _lib.OP_DOWNSUM
            o.t[0], o.i[0], o.i[1], o.i[2], o.i[3] = g_up1.view(), N, nf, h, w
            o.t[1], o.i[4], o.f[0], o.t[2], o.t[3] = NULL_T, 0, SLOPE, self.g_t0.view(), NULL_T
            ops.add(o)

    def _build_backward_trunk(self, ops):
        net, N, h, w = self.net, self.N, self.h, self.w
        nf, nb, P, pack, pk = net.nf, net.nb, net.params, net.pack, net.pk
        # LR_conv (model.1.sub.nb): t0 = fea + LR_conv(x_last)
        lrk = 'model.1.sub.%d.' % nb
        self._wg(ops, lrk, self.g_t0, True, self.x_last, True, nf, nf, h, w, h, w)
        ops.tag(11)
        # gradient buckets: (index into ops, lo, hi): params.grad[lo:hi] is complete once ops[:index] have run
        self._marks = [(len(ops.ops), P.off(lrk + 'weight'), P.total)]
        free = list(self.gstream)
        G = free[0]
        gs_cur = 0
        f16 = int(net.rdb_f16)
        gsc = self.store.gscale   # f16 dense blocks: the gradient slabs hold gsc * dL/d(.) (1.0 for bf16 storage); the fp32 gradient stream stays unscaled
        ops.add(conv_op(pack, pk['lr_b'], self.g_t0.view(), True, nf, h, w, h, w, N, out_f32=G.view(),
                        out_bf16=self.gslab[gs_cur].view(0), gamma=0.04 * gsc, out16_f16=f16))
        if f16:
            self.store.register_scaled(ops.ops[-1], 0.04, False)
        # RRDB chain, reversed.  No weight-gradient launch inside the chain: every RDB keeps its gradient slab and TrunkStore.phase computes all
        # of them afterwards (grouped launches over the whole batch).
        # DASR_CHAIN=1 (see _build_forward): the 15 nb data-gradient convs as one persistent chained launch (bf16 storage only: the f16 path patches
        # the scale factors of recorded ops when the gradient scale is re-calibrated, the chain holds copies)
        use_chain = self.chain is not None and not f16
        main_ops = ops
        if use_chain:
            ops = OpList()
        for i in range(nb - 1, -1, -1):
            Grr = G  # grad wrt the RRDB output
            Gout = None  # grad wrt the current RDB output (None: it is 0.2*Grr, folded into the epilogue)
            for r in (3, 2, 1):
                ridx = 3 * i + (r - 1)
                S, Gs = self.slabs[ridx], self.gslab[gs_cur]
                for k in range(4, 0, -1):
                    cin_b = nf + (4 - k) * GC
                    ops.add(conv_op(pack, pk[(i, r, 'b', k)], Gs.view(0), False, cin_b, h, w, h, w, N,
                                    mask=S.view(nf + (k - 1) * GC), mask_f32=0, out_bf16=Gs.view(cin_b), out16_f16=f16))
                # g_x conv with the residual bookkeeping fused
                Gin = next(b for b in free if b is not Grr and b is not Gout)
                first = (ridx == 0)
                nxt = None if first else self.gslab[gs_cur + 1].view(0)
                if r == 3:
                    ops.add(conv_op(pack, pk[(i, r, 'b', 0)], Gs.view(0), False, nf + 4 * GC, h, w, h, w, N,
                                    res1=Grr.view(), beta1=0.2, out_f32=Gin.view(), out_bf16=nxt, gamma=0.2 * gsc, alpha=1.0 / gsc, out16_f16=f16))
                elif r == 2:
                    ops.add(conv_op(pack, pk[(i, r, 'b', 0)], Gs.view(0), False, nf + 4 * GC, h, w, h, w, N,
                                    res1=Gout.view(), beta1=1.0, out_f32=Gin.view(), out_bf16=nxt, gamma=0.2 * gsc, alpha=1.0 / gsc, out16_f16=f16))
                else:
                    ops.add(conv_op(pack, pk[(i, r, 'b', 0)], Gs.view(0), False, nf + 4 * GC, h, w, h, w, N,
                                    res1=Gout.view(), beta1=1.0, res2=Grr.view(), beta2=1.0, out_f32=Gin.view(), out_bf16=nxt, gamma=0.04 * gsc, alpha=1.0 / gsc,
                                    out16_f16=f16))
                if f16:
                    self.store.register_scaled(ops.ops[-1], 0.04 if r == 1 else 0.2, True)
                Gout = Gin
                gs_cur += 1
            G = Gout
        if use_chain:
            body = list(ops.ops)
            assert len(body) == 15 * nb and all(o.op == _lib.OP_CONV for o in body)
            deps = [0 if k % 5 == 0 else (o.conv.cin // 16) - GC // 16 for k, o in enumerate(body)]
            tiles = ceil_div(h, 16) * ceil_div(w, 32)
            nsub = len(self.chains)
            per = N // nsub
            ops = main_ops
            self.chains_b = []   # (the last conv writes no 16-bit planes: its own launch)
            if self.chain_form == 'is':
                ch = ConvChain(body, deps, N, ceil_div(h, 4) * ceil_div(w, 32), net.device, err=self.chain.err, form='is')
                self.chains_b.append(ch)
                ops.add(ch.op())
                ops.keep.append(ch)
            else:
                for sb in range(nsub):
                    sub = body[:-1] if nsub == 1 else [_sub_batch_conv(o, sb * per, per, N) for o in body[:-1]]
                    ch = ConvChain(sub, deps[:-1], per, tiles, net.device, err=self.chain.err)
                    self.chains_b.append(ch)
                    ops.add(ch.op())
                    ops.keep.append(ch)
                ops.add(body[-1])
            self.chain_b = self.chains_b[0]
        ops.tag(4)
        rrdb0 = P.off('model.1.sub.0.RDB1.conv1.0.weight')
        if not self.shared_store:   # this plan owns the whole batch: the weight-gradient phase follows the chain in the same list
            st = self.store
            for first_op, end_op, lo, hi in st.groups:
                ops.ops.extend(st.phase.ops[first_op:end_op])
                self._marks.append((len(ops.ops), lo, hi))
            ops.keep.append(st)
            ops._arr = None
        # ShortcutBlock: g_fea = g_chain + g_t0
        o = Op()
        o.op = _lib.OP_AXPBY
        o.t[0], o.f[0], o.t[1], o.f[1] = G.view(), 1.0, self.g_t0.view(), 1.0
        o.i[0], o.i[1], o.i[2], o.i[3] = N, nf, h, w
        o.t[2], o.t[3], o.f[2] = self.g_fea.view(), NULL_T, 1.0
        ops.add(o)
        self._wg(ops, 'model.0.', self.g_fea, True, self.x_in, True, nf, net.in_nc, h, w, h, w)
        self._marks.append((len(ops.ops), 0, rrdb0))
        ops.tag(11)
        self.bwd = ops
        self._segments = None

    def set_grad_scale(self, scale):
        """fold 1/world_size into the deterministic wgrad reduction (data-parallel mean gradient); True if anything changed"""
        changed = False
        for o in self.bwd.ops:
            if o.op == _lib.OP_WGRAD_REDUCE and o.f[0] != scale:
                o.f[0] = scale
                changed = True
        if changed:
            self.bwd._arr = None
            self._segments = None
        return changed

    def run_backward_dp(self, dp, g):
        """the backward list in gradient-bucket segments, every finished bucket handed to the exchange (dp.reduce_async: SUM all-reduce on the
        communication stream, overlapped with the segments that follow) -- EXCEPT across a chained launch: dasr_conv_chain needs every workgroup slot
        of the device for its 512 resident workgroups, and a collective's kernels in flight on the communication stream would hold some of them
        (VERDICT / ADVICE r04).  A bucket that is complete in front of a segment with a chained launch is therefore held back and handed over together
        with that segment's own bucket: its event is recorded behind the chain, so the exchange overlaps the grouped weight-gradient launches only.
        (The forward chain of the next step is safe by the same rule: the trainers wait for the communication stream -- dp.wait() -- in front of the
        optimiser step.)"""
        segs = self.bwd_segments()
        has_chain = [any(o.op in (_lib.OP_CONV_CHAIN, _lib.OP_RDB_CHAIN) for o in seg.ops) for seg, _ in segs]
        last_chain = max([k for k, c in enumerate(has_chain) if c], default=-1)   # (EVERY bucket in front of the last chain-bearing segment is held, not only the one
        held = []                                                                  #  directly in front of it: a bucket mark ahead of the trunk must not bring the overlap back)
        for k, (seg, (lo, hi)) in enumerate(segs):
            seg.run()
            held.append((lo, hi))
            if k < last_chain:
                continue
            for a, b in held:
                dp.reduce_async(g[a:b])
            held = []
        dp.wait()

    def bwd_segments(self):
        """backward op list cut at gradient-bucket boundaries: [(OpList, (lo, hi) of the flat grad buffer that is
        complete once the segment has run)], in execution order (the buffer fills from its end)."""
        if self._segments is None:
            segs, prev_idx = [], 0
            for idx, lo, hi in self._marks:
                ol = OpList()
                ol.ops = self.bwd.ops[prev_idx:idx]
                segs.append((ol, (lo, hi)))
                prev_idx = idx
            self._segments = segs
        return self._segments
