"""Data-parallel step on ONE GPU box: two processes (both on cuda:0) exchange gradients through torch.distributed
(gloo on CUDA tensors stands in for RCCL, which refuses two ranks on one device).  Everything else is the production
path: per-rank shard of the batch, 1/world folded into the wgrad reduction, bucketed all-reduce, replicated Adam.
The result must equal the single-process full-batch step (shard-mean gradient == full-batch gradient)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, case, out, streams):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      DASR_STREAMS=str(streams))
    import torch
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.dist import DataParallelGroup, shard_minibatch
    from dasr_amd.models import create_model
    torch.cuda.set_device(0)
    dp = DataParallelGroup(backend='gloo') if world > 1 else None
    opt = fixtures.make_opt(case)
    opt['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(opt))
    kind = fixtures.CASES[case]['kind']
    from oracle import nets
    sd = fixtures.seeded_state_dict(m.netG.state_dict(), 1, 0.1)
    m.netG.load_state_dict(sd)
    if kind == 'dasr':
        m.netD_target.load_state_dict(fixtures.seeded_state_dict(m.netD_target.state_dict(), 2, 1.0))
    if dp:
        m.dp = dp
        for net in m.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    batch = fixtures.make_batch(case)
    if dp:
        batch = shard_minibatch(batch, rank, world)
    for step in (1, 2):
        m.update_learning_rate()
        m.feed_data(batch, True) if kind == 'dasr' else m.feed_data(batch)
        m.optimize_parameters(step)
    torch.cuda.synchronize()
    res = {'G': m.netG.state_dict(), 'log': dict(m.get_current_log())}
    if kind == 'dasr':
        res['D'] = m.netD_target.state_dict()
    torch.save(res, out % (world, rank))
    if dp:
        dp.barrier()


@pytest.mark.parametrize('case', ['sr_nf64_nb2_b2_32', 'dasr_wavelet_nf32_nb2_n2_32'])
def test_two_rank_step_equals_full_batch_step(case, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import torch.multiprocessing as mp
    out = str(tmp_path / 'w%d_r%d.pt')
    port = 29611 + (os.getpid() % 300)
    mp.spawn(_worker, args=(1, port, case, out, 1), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, port + 1, case, out, 1), nprocs=2, join=True)
    full = torch.load(out % (1, 0))
    r0, r1 = torch.load(out % (2, 0)), torch.load(out % (2, 1))
    for net in [k for k in ('G', 'D') if k in full]:
        for k, v in full[net].items():
            assert torch.equal(r0[net][k], r1[net][k]), (net, k)  # replicas stay bit-identical
            d = (r0[net][k] - v).abs().max().item()
            assert d <= 3.2e-4, (net, k, d)                      # Adam: sign flips of ~0 gradients move a weight by 2*lr
            assert ((r0[net][k] - v).abs() > 2e-5).float().mean().item() < 0.02, (net, k)


def _dsn_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    from dasr_amd.dist import DataParallelGroup
    from dasr_amd.dsn_model import DSNModel
    from oracle.gen_golden_dsn import dsn_state, dsn_batch
    torch.cuda.set_device(0)
    dp = DataParallelGroup(backend='gloo') if world > 1 else None
    torch.manual_seed(0)
    m = DSNModel(dict(filter='wavelet', w_per=0.01, vgg_seed=78))
    m.netG.load_state_dict(dsn_state(m.netG.state_dict(), 21, 0.5))
    m.netD.load_state_dict(dsn_state(m.netD.state_dict(), 22, 1.0))
    if dp:
        m.dp = dp
        for net in m.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    hr, bic, real = dsn_batch(dict(n=2, crop=128))  # VGG16's five pools need >= 32 px LR
    if dp:
        hr, bic, real = (t[rank:rank + 1] for t in (hr, bic, real))
    for _ in range(2):
        m.iteration(hr.cuda(), bic.cuda(), real.cuda())
    torch.cuda.synchronize()
    torch.save({'G': m.netG.state_dict(), 'D': m.netD.state_dict()}, out % (world, rank))
    if dp:
        dp.barrier()


def test_dsn_two_rank_iteration_equals_full_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    import torch.multiprocessing as mp
    out = str(tmp_path / 'dsn_w%d_r%d.pt')
    port = 29911 + (os.getpid() % 300)
    mp.spawn(_dsn_worker, args=(1, port, out), nprocs=1, join=True)
    mp.spawn(_dsn_worker, args=(2, port + 1, out), nprocs=2, join=True)
    full = torch.load(out % (1, 0))
    r0, r1 = torch.load(out % (2, 0)), torch.load(out % (2, 1))
    for net in ('G', 'D'):
        for k, v in full[net].items():
            assert torch.equal(r0[net][k], r1[net][k]), (net, k)
            if k in ('net.net.2.bias', 'net.net.5.bias'):
                continue  # bias in front of an InstanceNorm: true gradient 0, Adam turns the rounding noise into +-lr steps (no effect on D)
            d = (r0[net][k] - v).abs().max().item()
            assert d <= 4.2e-4, (net, k, d)   # two Adam steps at lr 1e-4: a flipped ~0 gradient moves a weight by <= 2*lr per step
            if v.numel() > 64:
                assert ((r0[net][k] - v).abs() > 2e-5).float().mean().item() < 0.03, (net, k)
