"""Trainer objects with the reference's duck-typed interface (codes/SRN/models/base_model.py:6-85,
SR_model.py:18-173, DASR_model.py:24-460): update_learning_rate / feed_data / optimize_parameters /
get_current_log / get_current_learning_rate / test / get_current_visuals / save / save_training_state /
resume_training, and the same checkpoint files ({iter}_G.pth, {iter}_D_target.pth, {iter}.state).

All arithmetic of the step runs in libdasr_hip.so; this file only sequences recorded op lists, keeps the
learning-rate schedule, and (de)serialises checkpoints in the reference's torch formats.
"""
import bisect
import ctypes as C
import logging
import os
from collections import Counter, OrderedDict

import torch

from . import _lib
from .engine import Op, OpList, NULL_T, ensure_runtime_ready, _stream, run_parallel
from .init import kaiming_state_dict
from .rrdbnet import RRDBNetHIP, rrdbnet_param_spec

logger = logging.getLogger('base')


def create_model(opt):
    """codes/SRN/models/__init__.py:5-26; 'DASR_FS_ESRGAN_patchGAN' (shipped JSONs) aliases 'DASR'."""
    model = opt['model']
    if model == 'sr':
        m = SRModel(opt)
    elif model in ('DASR', 'DASR_FS_ESRGAN_patchGAN'):
        from .dasr_model import DASR_Model
        m = DASR_Model(opt)
    else:
        raise NotImplementedError('Model [{:s}] not recognized.'.format(model))
    logger.info('Model [{:s}] is created.'.format(m.__class__.__name__))
    return m


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR semantics as the reference uses it (scheduler stepped at the
    top of every iteration, DASR_model.py:146-149, base_model.py:35-37)."""

    def __init__(self, base_lr, milestones, gamma):
        self.base_lr, self.milestones, self.gamma = base_lr, sorted(int(m) for m in milestones), gamma
        self.last_epoch = 0

    def step(self):
        self.last_epoch += 1

    def get_lr(self):
        return self.base_lr * self.gamma ** bisect.bisect_right(self.milestones, self.last_epoch)

    def state_dict(self):
        return {'milestones': Counter(self.milestones), 'gamma': self.gamma, 'base_lrs': [self.base_lr],
                'last_epoch': self.last_epoch, '_step_count': self.last_epoch + 1, '_last_lr': [self.get_lr()]}

    def load_state_dict(self, sd):
        self.last_epoch = sd['last_epoch']
        self.gamma = sd.get('gamma', self.gamma)
        if 'milestones' in sd:
            self.milestones = sorted(sd['milestones'].elements()) if isinstance(sd['milestones'], Counter) else sorted(sd['milestones'])
        if 'base_lrs' in sd:
            self.base_lr = sd['base_lrs'][0]


class AdamHIP:
    """torch.optim.Adam (eps 1e-8, no amsgrad, L2 weight decay) over a ParamStore's flat buffers."""

    def __init__(self, params, lr, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8, gate=None):
        """gate: device int32 word; while it is non-zero a step changes nothing (dasr_adam gate_flag).  The trainers pass the error word of the
        generator's chained trunk launches (RRDBNetHIP.chain_err): a step whose trunk ran behind a broken neighbour wait never reaches the weights.
        (The host-side step_count still advances on a gated step: such a run is not resumed -- the trainers raise at the next logging interval, on every
        rank under data parallelism (DataParallelGroup.sync_error_words), and refuse to write a checkpoint while the word is set.)"""
        self.params, self.lr, self.betas, self.wd, self.eps = params, lr, betas, weight_decay, eps
        self.gate = gate
        self.step_count = 0
        self.nonfinite = torch.zeros(1, dtype=torch.int32, device=params.device)   # set by the kernel when a gradient element is inf / NaN

    def step(self, lr):
        self.step_count += 1
        P = self.params
        _lib.check(_lib.lib().dasr_adam(P.flat.data_ptr(), P.grad.data_ptr(), P.m.data_ptr(), P.v.data_ptr(), P.total, lr,
                                        self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, self.nonfinite.data_ptr(),
                                        self.gate.data_ptr() if self.gate is not None else None, _stream()), 'adam')

    def check_finite(self, what='generator', hint=None):
        """a non-finite gradient has reached the optimiser since the last check?  Costs one device->host sync: called where the trainer
        synchronises anyway (get_current_log) and by the drivers on EVERY rank at the logging interval (under data parallelism the flag is set on
        all ranks -- the gradients are all-reduced in front of Adam -- so all ranks raise together instead of rank 0 leaving the others inside
        the next collective).  DASR_ALLOW_NONFINITE=1: instrumented / ablation builds (scripts/) compute wrong results on purpose."""
        if int(self.nonfinite.item()):
            self.nonfinite.zero_()
            if os.environ.get('DASR_ALLOW_NONFINITE') == '1':
                return
            P = self.params
            if os.environ.get('DASR_DBG_DUMP'):   # debugging aid: which plan tensors hold non-finite values right now
                import gc
                for o in gc.get_objects():
                    if type(o).__name__.endswith('Plan'):
                        for k, t in list(vars(o).items()):
                            for i, x in enumerate(t if isinstance(t, (list, tuple)) else [t]):
                                tt = getattr(x, 't', None)
                                if torch.is_tensor(tt) and tt.is_floating_point() and not bool(torch.isfinite(tt.float()).all()):
                                    nf = (~torch.isfinite(tt.float())).nonzero()
                                    print('[dbg] %s.%s[%d] %s %s: %d non-finite, first at %s' % (type(o).__name__, k, i, tuple(tt.shape), tt.dtype, len(nf), nf[0].tolist()))
            bad = [k for k in P.spec if not bool(torch.isfinite(P.view(k, P.grad)).all())][:6]   # (the last step's gradients; error path only)
            hint = hint or ('Paths of this network that store activations / gradients in f16 with a power-of-two pre-scale sized for mean losses of weight ~1 '
                            '(very large loss weights or activations above 65504 overflow them): the HR tail of the RRDBNet generator (DASR_HR_PREC=3 keeps it '
                            'in split-bf16 on f32 tensors), the 16-bit backward of the DSN generator (DASR_DSN_BWD16=0), the one-pass f16 data gradient of the '
                            'perceptual VGG (DASR_VGG_BWD_PREC=5).')
            raise FloatingPointError('inf / NaN in the %s gradients (the weights have absorbed it)%s.  %s'
                                     % (what, (': non-finite now in ' + ', '.join(bad)) if bad else '', hint))

    def state_dict(self, lr):
        P = self.params
        state = {}
        for i, k in enumerate(P.spec):
            if self.step_count > 0:
                state[i] = {'step': torch.tensor(float(self.step_count)), 'exp_avg': P.view(k, P.m).detach().clone().cpu(),
                            'exp_avg_sq': P.view(k, P.v).detach().clone().cpu()}
        group = {'lr': lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.wd, 'amsgrad': False,
                 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
                 'initial_lr': self.lr, 'params': list(range(len(P.spec)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        P = self.params
        for i, k in enumerate(P.spec):
            st = sd['state'].get(i)
            if st is not None:
                P.view(k, P.m).copy_(st['exp_avg'].to(P.device))
                P.view(k, P.v).copy_(st['exp_avg_sq'].to(P.device))
                self.step_count = int(float(st['step']))
        g = sd['param_groups'][0]
        self.betas, self.eps, self.wd = tuple(g['betas']), g['eps'], g['weight_decay']


class BaseModel:
    def __init__(self, opt):
        self.opt = opt
        if opt['gpu_ids'] is None:
            raise _lib.DasrHipError('gpu_ids is null: the MI355X trainer has no CPU path (use the oracle for CPU runs)')
        ensure_runtime_ready()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []
        # data-parallel group (one process per GPU, see dasr_amd/dist.py); None = single GPU
        self.dp = None

    def update_learning_rate(self):
        for s in self.schedulers:
            s.step()

    @property
    def lpips_label(self):
        """label of the val_lpips column: 'LPIPS(random)' when the metric network is seeded (allow_random_perceptual), see lpips.load_lpips"""
        from .lpips import lpips_label
        return lpips_label(getattr(self, 'cri_fea_lpips', None))

    def get_current_learning_rate(self):
        return self.schedulers[0].get_lr()

    def get_network_description(self, network):
        n = network.params.total
        s = '%s(%s)' % (network.__class__.__name__, ', '.join('%s%s' % (k, list(v[1])) for k, v in list(network.params.spec.items())[:4]) + ', ...')
        return s, n

    def save_network(self, network, network_label, iter_step):
        """{iter}_{label}.pth = plain state_dict of CPU tensors (base_model.py:49-58)."""
        path = os.path.join(self.opt['path']['models'], '{}_{}.pth'.format(iter_step, network_label))
        if hasattr(self, 'check_finite'):
            self.check_finite()   # (host sync) never write weights behind a non-finite gradient or an invalid chained launch
        torch.save(network.state_dict(), path)

    def load_network(self, load_path, network, strict=True):
        network.load_state_dict(torch.load(load_path, map_location='cpu'), strict=strict)

    def save_training_state(self, epoch, iter_step):
        """training_state/{iter}.state (base_model.py:65-74); optimizer/scheduler entries are torch-format dicts."""
        state = {'epoch': epoch, 'iter': iter_step, 'schedulers': [s.state_dict() for s in self.schedulers],
                 'optimizers': [o.state_dict(s.get_lr()) for o, s in zip(self.optimizers, self.schedulers)]}
        torch.save(state, os.path.join(self.opt['path']['training_state'], '{}.state'.format(iter_step)))

    def resume_training(self, resume_state, opt=None):
        ro, rs = resume_state['optimizers'], resume_state['schedulers']
        assert len(ro) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(rs) == len(self.schedulers), 'Wrong lengths of schedulers'
        for i, o in enumerate(ro):
            self.optimizers[i].load_state_dict(o)
        for i, s in enumerate(rs):
            self.schedulers[i].load_state_dict(s)


def _define_G(opt, device, rdb_prec=None):
    """networks.py:83-147 restricted to the hot-path generator (RRDB_net, upconv)."""
    g = opt['network_G']
    which = g['which_model_G']
    if which not in ('RRDB_net', 'RRDB_mask'):
        raise NotImplementedError('Generator model [{:s}] not recognized'.format(which))
    # `upsample_mode` is an extension of the option surface: the reference's define_G hard-wires 'upconv' (networks.py:96-99) although
    # its RRDBNet also implements 'pixelshuffle' (architecture.py:186-191); absent / null = the reference's behaviour
    um = g['upsample_mode'] or 'upconv'
    # rdb_prec: operand format of the dense blocks (None: DASR_RDB_PREC or bf16; 2: f16 storage, see RRDBNetHIP)
    net = RRDBNetHIP(in_nc=g['in_nc'], out_nc=g['out_nc'], nf=g['nf'], nb=g['nb'], upscale=g['scale'], device=device, upsample_mode=um, rdb_prec=rdb_prec)
    if opt['is_train']:
        logger.info('Initialization method [kaiming]')
        net.load_state_dict(kaiming_state_dict(rrdbnet_param_spec(g['in_nc'], g['out_nc'], g['nf'], g['nb'], um), 0.1))
    else:
        net.repack()
    return net


class SRModel(BaseModel):
    """Generator-only trainer (SR_model.py:18-173): l_pix_w * L1|L2(G(LR), HR), Adam(default betas), MultiStepLR."""

    def __init__(self, opt):
        super().__init__(opt)
        t = opt['train']
        self.scale = opt['scale']
        self.netG = _define_G(opt, self.device)
        self.load()
        if self.is_train:
            if t['pixel_criterion'] not in ('l1', 'l2'):   # nn.L1Loss / nn.MSELoss (SR_model.py:31-37)
                raise NotImplementedError('Loss type [{:s}] is not recognized.'.format(str(t['pixel_criterion'])))
            self.pix_l2 = t['pixel_criterion'] == 'l2'
            self.l_pix_w = t['pixel_weight']
            self.netG.loss_weight = float(self.l_pix_w or 1.0)   # sizes the f16 pre-scale of the dense-block gradients when DASR_RDB_PREC=2 (TrunkStore.gscale)
            wd = t['weight_decay_G'] if t['weight_decay_G'] else 0
            self.optimizer_G = AdamHIP(self.netG.params, t['lr_G'], (0.9, 0.999), wd, gate=self.netG.chain_err)
            self.optimizers.append(self.optimizer_G)
            if t['lr_scheme'] != 'MultiStepLR':
                raise NotImplementedError('MultiStepLR learning rate scheme is enough.')
            self.schedulers.append(MultiStepLR(t['lr_G'], t['lr_steps'], t['lr_gamma']))
            self.log_dict = OrderedDict()
            self.loss_acc = torch.zeros(8, dtype=torch.float32, device=self.device)   # one slot per sub-batch replica stream (<= 8): the logged loss is their sum in slot order -- the same value run to run for any replica count
        self._step_ops = {}

    def feed_data(self, data, need_HR=True):
        self.var_L = data['LR'].to(self.device, non_blocking=True)
        if 'HR' in data:
            self.real_H = data['HR'].to(self.device, non_blocking=True)

    def _ops_for(self, plan, n_total):
        """loss op list for this plan: L1 forward+gradient into plan.g_sr (mean over the WHOLE batch of n_total crops)"""
        key = id(plan)
        if key not in self._step_ops:
            N, C_, H, W = plan.N, self.real_H.shape[1], self.real_H.shape[2], self.real_H.shape[3]
            hr_buf = torch.zeros((N, C_, H, W), dtype=torch.float32, device=self.device)
            ops = OpList()
            o = Op()
            o.op = _lib.OP_L1LOSS
            o.t[0], o.p[0], o.p[1] = plan.sr.view(), hr_buf.data_ptr(), None
            o.i[0], o.i[1], o.i[2], o.i[3], o.i[4] = N, C_, H, W, (2 if self.pix_l2 else 0)
            o.f[0] = float(self.l_pix_w) / float(n_total * C_ * H * W)
            o.p[2], o.t[1] = self.loss_acc.data_ptr() + 4 * (plan.replica % 8), plan.g_sr.view()
            fused = plan.take_f16_loss_gradient()   # (f16 HR tail: dL/dSR goes straight into its f16 input, pre-scaled; no fp32 image, no conversion pass)
            if fused is not None:
                o.t[1], o.f[1] = fused
                o.i[4] |= 4
            ops.add(o)
            self._step_ops[key] = (ops, hr_buf)
        return self._step_ops[key]

    def _sub_plans(self, N, h, w):
        """Sub-batches processed concurrently on separate HIP streams: at batch 16 one launch has only 512 workgroups
        (2 per CU) and ~35 % of every kernel is ramp / first-load / epilogue latency; two independent half-batches
        fill those gaps with each other's main loops.  DASR_STREAMS=1 disables it."""
        k = max(1, int(os.environ.get('DASR_STREAMS', '2')))
        k = min(k, N // 4) if N >= 8 else 1   # a replica keeps at least 4 crops (128 workgroups per dense-block launch at 128 x 128)
        if self.netG.chain_ok(N, h, w):
            k = 1   # the whole batch fills the chip exactly: the trunk runs as persistent chained launches (rrdbnet._Plan), which beat the two-stream schedule
        if k <= 1:
            if (N, h, w, 0, 0, 0) not in self.netG.plans:
                self.netG.concurrent_replicas = 1   # a plan built now has the chip to itself (wgrad split count)
            return [self.netG.plan(N, h, w)]
        self.netG.concurrent_replicas = k  # the 1-WG/CU wgrad launches of the replicas must fit on the chip together
        # deferred dense-block weight gradients: the replicas work on image ranges of ONE set of slabs (TrunkStore) and the weight-gradient
        # phase runs once over the whole batch after both data-gradient chains (rrdbnet.TrunkStore)
        store = self.netG.trunk_store(N, h, w)
        sizes = [N // k + (1 if i < N % k else 0) for i in range(k)]   # uneven splits allowed (16 -> 6 + 5 + 5)
        offs = [sum(sizes[:i]) for i in range(k)]
        return [self.netG.plan(sizes[i], h, w, replica=i, store=store, n0=offs[i]) for i in range(k)]

    @staticmethod
    def _whole_step(plan, loss_ops):
        """forward + loss + backward of one sub-batch replica as ONE recorded list; an event is recorded on the replica's stream at
        every gradient-bucket boundary of the backward list (plan.buckets: [(event, lo, hi)] in completion order)"""
        if not hasattr(plan, 'whole_step'):
            from .rrdbnet import _sched
            ws = OpList()
            ws.extend(plan.fwd)
            ws.extend(loss_ops)
            plan.buckets, prev = [], 0
            for idx, lo, hi in plan._marks:
                ws.ops.extend(plan.bwd.ops[prev:idx])
                ev = plan._event()
                ws.ops.append(_sched(_lib.OP_EVENT_RECORD, ev))
                plan.buckets.append((ev, lo, hi))
                prev = idx
            ws.ops.extend(plan.bwd.ops[prev:])
            ws.keep.extend(plan.bwd.keep)
            ws._arr = None
            plan.whole_step = ws
            plan.whole_step_gen = getattr(plan, 'whole_step_gen', 0) + 1   # identifies THIS recording (id() of a rebuilt list can repeat)
        return plan.whole_step

    def _bucket_ops(self, plans):
        """per gradient bucket: op list for the communication stream = wait for every replica's boundary event, then add the private
        gradient slices of replicas 1.. into params.grad[lo:hi]"""
        key = tuple((id(p), p.whole_step_gen) for p in plans)
        if getattr(self, '_bucket_key', None) != key:
            from .rrdbnet import _sched
            g, out = self.netG.params.grad, []
            for k in range(len(plans[0].buckets)):
                _, lo, hi = plans[0].buckets[k]
                ol = OpList()
                for p in plans:
                    ol.add(_sched(_lib.OP_STREAM_WAIT, p.buckets[k][0]))
                for p in plans[1:]:
                    o = Op()
                    o.op = _lib.OP_ADD_FLAT
                    o.p[0], o.p[1], o.l[0] = g.data_ptr() + 4 * lo, p.grad.data_ptr() + 4 * lo, hi - lo
                    if hi > lo:
                        ol.add(o)
                out.append((ol, lo, hi))
            self._bucket_list, self._bucket_key = out, key
        return self._bucket_list

    def optimize_parameters(self, step):
        N, _, h, w = self.var_L.shape
        plans = self._sub_plans(N, h, w)
        L = _lib.lib()
        _lib.check(L.dasr_fill_f32(self.loss_acc.data_ptr(), 8, 0.0, _stream()), 'fill')
        dp_on = self.dp is not None and self.dp.active
        if len(plans) == 1:
            plan = plans[0]
            loss_ops, hr_buf = self._ops_for(plan, N)
            hr_buf.copy_(self.real_H)
            plan.set_input(self.var_L)
            plan.fwd.run()
            loss_ops.run()
            if plan.store.calibrate_due():   # f16 dense blocks: measure dL/d(trunk output) behind the HR tail, then run the backward from the start
                plan.bwd.run(0, plan.tail_end)
                plan.store.set_gscale_from(float(plan.g_t0.t.abs().max().item()))
            if not dp_on:
                plan.bwd.run()
            else:
                plan.set_grad_scale(self.dp.grad_scale)
                plan.run_backward_dp(self.dp, self.netG.params.grad)   # bucket-wise exchange overlapped with the backward, never across a chained launch
        else:
            if len(getattr(self, '_streams', ())) != len(plans):
                self._streams = [torch.cuda.Stream() for _ in plans]
            cur = torch.cuda.current_stream()
            if plans[0].store.calibrate_due():
                # f16 dense blocks: the replicas share ONE gradient scale (their slabs feed the same grouped weight-gradient launches), measured on a
                # dry run of replica 0 -- forward, loss, HR-tail backward: no optimiser / gradient-buffer side effects (loss_acc is cleared below)
                p0 = plans[0]
                l0, hb0 = self._ops_for(p0, N)
                hb0.copy_(self.real_H[p0.n0:p0.n0 + p0.N])
                p0.set_input(self.var_L[p0.n0:p0.n0 + p0.N])
                p0.fwd.run()
                l0.run()
                p0.bwd.run(0, p0.tail_end)
                p0.store.set_gscale_from(float(p0.g_t0.t.abs().max().item()))
                _lib.check(L.dasr_fill_f32(self.loss_acc.data_ptr(), 8, 0.0, _stream()), 'fill')
            steps = []
            for i, (plan, st) in enumerate(zip(plans, self._streams)):
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    loss_ops, hr_buf = self._ops_for(plan, N)
                    hr_buf.copy_(self.real_H[plan.n0:plan.n0 + plan.N])
                    plan.set_input(self.var_L[plan.n0:plan.n0 + plan.N])
                    if dp_on and plan.set_grad_scale(self.dp.grad_scale) and hasattr(plan, 'whole_step'):
                        del plan.whole_step  # the recorded list holds copies of the patched reduce ops
                steps.append(self._whole_step(plan, loss_ops))
            run_parallel(steps, self._streams)
            g = self.netG.params.grad
            comm = self.dp.comm_stream if dp_on else None
            store = plans[0].store if plans[0].shared_store else None
            if store is not None and dp_on:
                store.set_grad_scale(self.dp.grad_scale)
            if comm is not None:
                # bucket k of the flat gradient buffer is summed over the sub-batch replicas and all-reduced over the ranks on the
                # communication stream as soon as BOTH replica streams have passed its boundary event: the exchange overlaps the
                # remaining data-gradient / weight-gradient kernels (reference mechanism replaced: networks.py:144-146)
                for k, (bops, lo, hi) in enumerate(self._bucket_ops(plans)):
                    with torch.cuda.stream(comm):
                        bops.run()
                        self.dp.all_reduce_here(g[lo:hi])
                for st in self._streams:
                    cur.wait_stream(st)
                if store is not None:
                    # deferred dense-block weight gradients: one grouped launch sequence over the whole batch on the main stream; every group's
                    # slice of the flat gradient buffer is all-reduced while the next group is computed
                    for first_op, end_op, lo, hi in store.groups:
                        store.phase.run(first_op, end_op)
                        comm.wait_stream(cur)
                        with torch.cuda.stream(comm):
                            self.dp.all_reduce_here(g[lo:hi])
                cur.wait_stream(comm)
            else:
                for st in self._streams:
                    cur.wait_stream(st)
                if store is not None:
                    # only the buckets outside the dense blocks hold per-replica weight gradients (HR tail + LR_conv, fea_conv)
                    for _, lo, hi in plans[0].buckets:
                        for plan in plans[1:]:
                            _lib.check(L.dasr_add_flat(g.data_ptr() + 4 * lo, plan.grad.data_ptr() + 4 * lo, hi - lo, _stream()), 'add_flat')
                    store.phase.run()
                else:
                    for plan in plans[1:]:
                        _lib.check(L.dasr_add_flat(g.data_ptr(), plan.grad.data_ptr(), g.numel(), _stream()), 'add_flat')
                if dp_on:
                    self.dp.allreduce_mean(g)
        self.optimizer_G.step(self.schedulers[0].get_lr())
        self.netG.repack()
        self._out_plans = plans
        self._fake_H = None
        self._l_pix_dev = self.loss_acc
        self.log_dict['l_pix'] = None  # materialised lazily by get_current_log (no per-step host sync)

    @property
    def fake_H(self):
        if getattr(self, '_fake_H', None) is not None:
            return self._fake_H
        ps = self._out_plans
        return ps[0].read_output() if len(ps) == 1 else torch.cat([p.read_output() for p in ps], 0)

    @fake_H.setter
    def fake_H(self, v):
        self._fake_H = v

    def check_finite(self):
        """raise FloatingPointError if a non-finite gradient reached an optimiser since the last check (every rank; see AdamHIP.check_finite)"""
        self.optimizer_G.check_finite('generator')
        for plan in (getattr(self, '_out_plans', None) or []):
            plan.check_chain()

    def sync_error_words(self):
        """data parallelism: called by the drivers on EVERY rank at the logging interval, in front of check_finite (a collective: never from a rank-0-only path such as a
        checkpoint write): every rank then holds the MAX of every rank's error words and all gate / raise together (DataParallelGroup.sync_error_words)"""
        if self.dp is not None and self.dp.active:
            self.dp.sync_error_words([self.optimizer_G.nonfinite, self.netG.chain_err])

    def get_current_log(self):
        if 'l_pix' in self.log_dict:
            self.log_dict['l_pix'] = float(sum(self._l_pix_dev.tolist()))   # (replica slots added in index order, in double)
            self.check_finite()
        return self.log_dict

    def test(self):
        """inference on var_L (SR_model.py:87-93; `chop`: quadrant inference of DASR_model.py:333-339 / util.py:87-147)"""
        if self.opt['chop']:
            from .util import forward_chop
            self.fake_H = forward_chop(self.var_L, self.opt['scale'], lambda x: self.netG.forward(x).clone(), min_size=320000)
        else:
            self.fake_H = self.netG.forward(self.var_L).clone()
        if self.opt['val_lpips']:    # SR_model.py:95-99
            from .lpips import load_lpips, lpips_metric
            if getattr(self, 'cri_fea_lpips', None) is None:
                self.cri_fea_lpips = load_lpips(self.opt, self.device)
            self.LPIPS = lpips_metric(self.cri_fea_lpips, self.fake_H, self.real_H)

    def get_current_visuals(self, need_HR=True):
        out = OrderedDict()
        out['LR'] = self.var_L.detach()[0].float().cpu()
        out['SR'] = self.fake_H.detach()[0].float().cpu()
        if self.opt['val_lpips']:
            out['LPIPS'] = self.LPIPS.detach().float().cpu()
        if need_HR:
            out['HR'] = self.real_H.detach()[0].float().cpu()
        return out

    def networks(self):
        return [self.netG]

    def print_network(self):
        s, n = self.get_network_description(self.netG)
        logger.info('Network G structure: {}, with parameters: {:,d}'.format(self.netG.__class__.__name__, n))

    def load(self):
        p = self.opt['path']['pretrain_model_G']
        if p is not None:
            logger.info('Loading pretrained model for G [{:s}] ...'.format(p))
            self.load_network(p, self.netG)

    def save(self, iter_step):
        self.save_network(self.netG, 'G', iter_step)
