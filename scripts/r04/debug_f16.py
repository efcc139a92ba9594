"""debug: per-tensor gradient errors of the SR step with f16 dense blocks (DASR_RDB_PREC=2) against the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import fixtures, nets, trainers
from dasr_amd import options
from dasr_amd.models import create_model

case = sys.argv[1] if len(sys.argv) > 1 else 'sr_nf64_nb2_b2_32'
c = fixtures.CASES[case]
opt = fixtures.make_opt(case)
netG = nets.RRDBNet(3, 3, c['nf'], c['nb'], 4)
sd0 = fixtures.seeded_state_dict(netG.state_dict(), 1, 0.1)
netG.load_state_dict(sd0)
t = trainers.SRTrainer(opt, netG=netG)
batch = fixtures.make_batch(case)
t.update_learning_rate(); t.feed_data(batch); t.optimize_parameters(1)
ref = {k: p.grad.detach().clone() for k, p in netG.named_parameters()}
opt['gpu_ids'] = [0]
m = create_model(options.dict_to_nonedict(opt))
m.netG.load_state_dict(sd0)
m.update_learning_rate(); m.feed_data(batch); m.optimize_parameters(1)
torch.cuda.synchronize()
gd = m.netG.params.grad_dict()
rel = lambda a, b: float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
P = m._out_plans[0]
print('gscale trunk', P.store.gscale, 'hr', P.gscale, 'rdb_f16', m.netG.rdb_f16)
for k in reversed(list(gd)):
    if k.endswith('weight'):
        r = float(gd[k].double().norm() / (ref[k].double().norm() + 1e-30))
        print('%-42s rel %.3e  norm ratio %.4f   bias rel %.3e' % (k, rel(gd[k], ref[k]), r, rel(gd[k.replace('weight', 'bias')], ref[k.replace('weight', 'bias')])))
# gradient slabs: fraction of non-finite / zero values, max
for i, g in enumerate(P.gslab[:6]):
    tt = g.t.float()
    print('gslab', i, 'max', float(tt.abs().max()), 'zero frac', float((tt == 0).float().mean()), 'finite', bool(torch.isfinite(tt).all()))
