"""CPU tests: C-ABI library loads and exports every declared symbol; host logic (options, schedule, init,
sharding, gloo all-reduce of gradient buckets)."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_all_declared_symbols():
    from dasr_amd import build, _lib
    lib = build.build()
    assert os.path.exists(lib)
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'dasr_hip.h')).read()
    declared = set(re.findall(r'^\s*(?:int|void\*)\s+(dasr_\w+)\s*\(', hdr, flags=re.M))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    assert L.dasr_abi_version() == _lib.ABI_VERSION


def test_struct_sizes_match_header_layout():
    """compile a tiny C program against include/dasr_hip.h and compare sizeof with the ctypes mirrors"""
    import ctypes
    from dasr_amd import _lib
    src = r'''#include <stdio.h>
#include "dasr_hip.h"
int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(dasr_tensor), sizeof(dasr_conv_params), sizeof(dasr_wgrad_part),
 sizeof(dasr_wgrad_reduce_part), sizeof(dasr_pack_seg), sizeof(dasr_pack_desc), sizeof(dasr_op), sizeof(dasr_crop_desc)); return 0;}'''
    import tempfile
    d = tempfile.mkdtemp()
    open(os.path.join(d, 't.c'), 'w').write(src)
    subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 't.c'), '-o', os.path.join(d, 't')])
    sizes = [int(x) for x in subprocess.check_output([os.path.join(d, 't')]).split()]
    mine = [ctypes.sizeof(c) for c in (_lib.Tensor, _lib.ConvParams, _lib.WgradPart, _lib.WgradReducePart, _lib.PackSeg,
                                       _lib.PackDesc, _lib.Op, _lib.CropDesc)]
    assert sizes == mine, (sizes, mine)


def test_options_parse_and_nonedict(tmp_path):
    from dasr_amd import options
    p = tmp_path / 'o.json'
    p.write_text('''{
  "name": "debug_x" // comment
  , "model": "DASR_FS_ESRGAN_patchGAN", "scale": 4, "gpu_ids": [0]
  , "datasets": {"train": {"name": "a", "mode": "LRHR", "dataroot_HR": "~/hr", "dataroot_LR": "~/x.lmdb", "batch_size": 2}}
  , "path": {"root": "/tmp/dasr_opt_test", "pretrain_model_G": null}
  , "network_G": {"which_model_G": "RRDB_net", "nf": 64, "nb": 23, "in_nc": 3, "out_nc": 3}
  , "train": {"lr_G": 1e-4, "val_freq": 5000}, "logger": {"print_freq": 200, "save_checkpoint_freq": 5000}
}''')
    opt = options.dict_to_nonedict(options.parse(str(p)))
    assert opt['is_train'] and opt['network_G']['scale'] == 4
    assert opt['datasets']['train']['data_type'] == 'lmdb' and opt['datasets']['train']['phase'] == 'train'
    assert opt['path']['models'].endswith('experiments/debug_x/models')
    assert opt['train']['val_freq'] == 8 and opt['logger']['print_freq'] == 2  # debug mode
    assert opt['nonexistent'] is None and opt['train']['nope'] is None
    assert os.environ['CUDA_VISIBLE_DEVICES'] == '0'
    assert 'lr_G' in options.dict2str(opt)


def test_options_parse_leaves_visible_devices_alone_under_a_launcher(tmp_path, monkeypatch):
    """one process per GPU: rank r uses device LOCAL_RANK; gpu_ids [0] of the shipped JSONs must not hide the other devices"""
    from dasr_amd import options
    p = tmp_path / 'o.json'
    p.write_text('{"name": "x", "model": "sr", "scale": 4, "gpu_ids": [0], "datasets": {}, "path": {"root": "/tmp/dasr_opt_test"}, '
                 '"network_G": {"which_model_G": "RRDB_net"}, "train": {}, "logger": {}}')
    monkeypatch.setenv('WORLD_SIZE', '2')
    monkeypatch.setenv('LOCAL_RANK', '1')
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', '0,1,2,3')
    options.parse(str(p))
    assert os.environ['CUDA_VISIBLE_DEVICES'] == '0,1,2,3'
    monkeypatch.delenv('WORLD_SIZE')
    monkeypatch.delenv('LOCAL_RANK')
    options.parse(str(p))
    assert os.environ['CUDA_VISIBLE_DEVICES'] == '0'


def test_multistep_lr_matches_torch():
    from dasr_amd.models import MultiStepLR
    p = torch.nn.Parameter(torch.zeros(1))
    o = torch.optim.Adam([p], lr=1e-3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ts = torch.optim.lr_scheduler.MultiStepLR(o, [2, 4, 7], 0.5)
        mine = MultiStepLR(1e-3, [2, 4, 7], 0.5)
        for _ in range(10):
            ts.step()
            mine.step()
            assert abs(o.param_groups[0]['lr'] - mine.get_lr()) < 1e-15
    sd = mine.state_dict()
    m2 = MultiStepLR(1e-3, [2, 4, 7], 0.5)
    m2.load_state_dict(sd)
    assert m2.get_lr() == mine.get_lr()


def test_reference_init_is_reproduced(golden_dir):
    from dasr_amd.init import kaiming_state_dict
    from dasr_amd.rrdbnet import rrdbnet_param_spec
    from oracle import nets
    ref = np.load(os.path.join(golden_dir, 'misc_modules.npz'))
    torch.manual_seed(5)
    sd = kaiming_state_dict(rrdbnet_param_spec(3, 3, 32, 1), 0.1)
    d = np.array([nets.tensor_digest(v) for v in sd.values()])
    np.testing.assert_array_equal(d, ref['init_digest'])


def test_param_spec_matches_reference_keys(golden_dir):
    from dasr_amd.rrdbnet import rrdbnet_param_spec
    ref = np.load(os.path.join(golden_dir, 'cfg1_sr_nf32_nb4_b2_64.npz'))
    assert [k for k, _ in rrdbnet_param_spec(3, 3, 32, 4)] == list(ref['state_keys'])
    n = sum(int(np.prod(s)) for _, s in rrdbnet_param_spec(3, 3, 64, 23))
    assert n == 16697987


def test_shard_minibatch():
    from dasr_amd.dist import shard_minibatch
    b = {'LR': torch.arange(8).view(8, 1), 'p': ['x']}
    s = shard_minibatch(b, 1, 4)
    assert s['LR'].flatten().tolist() == [2, 3] and s['p'] == ['x']


def _dp_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from dasr_amd.dist import DataParallelGroup
    dp = DataParallelGroup(backend='gloo')
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(1000, generator=g) * dp.grad_scale  # the wgrad reduction pre-scales by 1/world
    for lo, hi in ((600, 1000), (250, 600), (0, 250)):     # buckets complete from the end of the buffer
        dp.reduce_async(grad[lo:hi])
    dp.wait()
    m = dp.max_over_ranks(float(rank))
    torch.save({'grad': grad, 'max': m}, out % rank)
    dp.barrier()


def test_gloo_world2_bucketed_allreduce(tmp_path):
    import torch.multiprocessing as mp
    port = 29511 + (os.getpid() % 200)
    out = str(tmp_path / 'r%d.pt')
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out % 0), torch.load(out % 1)
    want = (torch.randn(1000, generator=torch.Generator().manual_seed(100)) + torch.randn(1000, generator=torch.Generator().manual_seed(101))) / 2
    assert torch.allclose(r0['grad'], want, atol=1e-6) and torch.equal(r0['grad'], r1['grad'])
    assert r0['max'] == 1.0 and r1['max'] == 1.0


def _env_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    if rank == 1:
        os.environ['DASR_VGG_PREC'] = '2'   # one rank with a different numerics switch
    from dasr_amd.dist import DataParallelGroup
    try:
        DataParallelGroup(backend='gloo')
        msg = 'no error'
    except RuntimeError as e:
        msg = str(e)
    open(out % rank, 'w').write(msg)


def test_ranks_must_agree_on_numerics_switches(tmp_path):
    """defaults-by-environment-variable is how an 8-rank run silently diverges between ranks; the group hashes the
    DASR_* numerics / schedule switches at start-up and every rank fails loudly on a mismatch"""
    import torch.multiprocessing as mp
    port = 29811 + (os.getpid() % 200)
    out = str(tmp_path / 'env_r%d.txt')
    mp.spawn(_env_worker, args=(2, port, out), nprocs=2, join=True)
    for r in (0, 1):
        msg = open(out % r).read()
        assert 'disagree on DASR_VGG_PREC' in msg, msg


def test_product_library_carries_one_kernel_family_per_op():
    """the measured-slower alternatives (wgrad3_kernel, wgrad3_glds_kernel, wgrad4_kernel, conv_ring3_kernel, the
    ring / loader-wave forms of the dense conv) are compiled into libdasr_hip_ablate.so (-DDASR_BENCH) only"""
    from dasr_amd import _lib
    data = open(_lib.LIB_PATH, "rb").read()
    fams = set(m.decode() for m in re.findall(rb'(conv_ring3_kernel|wgrad4_kernel|wgrad3_glds_kernel|wgrad3_kernel|wgrad3_ld_kernel|conv_glds_kernel|conv_kernel|wgrad_kernel)', data))
    assert fams == {'conv_glds_kernel', 'conv_kernel', 'wgrad3_ld_kernel', 'wgrad_kernel'}, fams
    # no ring / loader-wave instantiation of the LDS-DMA conv (template argument RING != 0): mangled names end in ...Lb<F16>ELi<RING>EEEv...
    rings = set(re.findall(rb'conv_glds_kernelILi\d+ELi\d+ELi\d+ELi(\d+)ELb[01]ELi(\d+)EE', data))
    assert rings and all(abl == b'0' and ring == b'0' for abl, ring in rings), rings
    # the environment switches that selected them are gone from the host code
    src = ''.join(open(os.path.join(ROOT, 'dasr_amd', f)).read() for f in os.listdir(os.path.join(ROOT, 'dasr_amd')) if f.endswith('.py'))
    for gone in ('DASR_WGRAD4', 'DASR_WGRAD_GLDS', 'DASR_WGRAD_LD', 'DASR_WGRAD_ABL', 'DASR_WG_DEFER', 'DASR_WG_BATCH', 'DASR_WG_GROUP', 'DASR_WG3_TARGET',
                 'DASR_HR_STORE', 'DASR_HR_MT', 'DASR_SUBPIXEL', 'DASR_ENQ_CHUNK'):
        assert gone not in src, gone


def test_product_path_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, 'dasr_amd')):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), f


# ---- DSN host side (codes/DSN) -------------------------------------------------------------------------------------
def test_dsn_default_init_replays_reference_rng_and_key_order(golden_dir):
    """nn.Conv2d / nn.PReLU default init in construction order under torch.manual_seed(0) (codes/DSN/train.py:76,127-135);
    key order against the fixtures made from the reference modules"""
    import numpy as np
    from dasr_amd import dsn_model as M
    from dasr_amd.gan_nets import fsd_spec
    from oracle import dsn
    for filt, case in (('gau', 'dsn_gau5_inst_b2_128'), ('wavelet', 'dsn_wavelet_inst_b2_128'), ('avg_pool', 'dsn_avg5_inst_b1_160')):
        gold = np.load(os.path.join(golden_dir, case + '.npz'))
        torch.manual_seed(0)
        G, D = dsn.DeResnet(), dsn.Discriminator(5, 'Instance', filt)
        torch.manual_seed(0)
        nc = 9 if filt == 'wavelet' else 3
        g_spec, d_spec = M.deresnet_spec(8), fsd_spec(nc, 5 if filt == 'gau' else None)[0]
        a, b = M.default_init_state(g_spec), M.default_init_state(d_spec)
        assert [k for k, _ in g_spec] == list(gold['G_keys']) and [k for k, _ in d_spec] == list(gold['D_keys'])
        assert all(torch.equal(a[k], v) for k, v in G.state_dict().items())
        assert all(torch.equal(b[k], v) for k, v in D.state_dict().items())
    assert sum(v.numel() for v in a.values()) == 668238
    # --norm_layer Batch: BatchNorm2d draws nothing from the RNG (weight 1, bias 0), the convs around it keep their draws
    torch.manual_seed(0)
    G, D = dsn.DeResnet(), dsn.Discriminator(5, 'Batch', 'gau')
    torch.manual_seed(0)
    d_spec, d_layers = fsd_spec(3, 5, norm='Batch')
    M.default_init_state(M.deresnet_spec(8))
    b = M.default_init_state(d_spec, bn_prefixes=[L['bn'] for L in d_layers if L['norm'] == 'batch'])
    ref_sd = D.state_dict()
    assert [k for k in ref_sd if 'running' not in k and 'num_batches' not in k] == [k for k, _ in d_spec]
    assert all(torch.equal(b[k], ref_sd[k]) for k in b)


def test_dsn_cli_flags_and_lr_rule():
    from dasr_amd import dsn_train, dsn_model
    o = dsn_train.build_parser().parse_args([])
    assert (o.batch_size, o.num_epochs, o.num_decay_epochs, o.learning_rate, o.adam_beta_1) == (4, 400, 150, 1e-4, 0.5)
    assert (o.w_col, o.w_tex, o.w_per, o.kernel_size, o.filter, o.discriminator, o.generator) == (1, 0.005, 0.01, 5, 'gau', 'FSD', 'DeResnet')
    assert (o.dataset, o.per_type) == ('df2k', 'LPIPS')   # the reference's defaults (codes/DSN/train.py:38,54)
    dsn_train.check_supported(o)
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--ragan']))
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--norm_layer', 'Batch']))
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--wgan', '--ragan']))   # (round 5)
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--norm_layer', 'Batch', '--discriminator', 'nld_s1']))   # (round 6: model.py:136-160)
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--wgan', '--norm_layer', 'Batch', '--discriminator', 'nld_s2']))       # (round 6: second-order pass through BatchNorm)
    for bad in (['--generator', 'SRGAN'], ['--discriminator', 'nld_s3'], ['--norm_layer', 'Group']):
        with pytest.raises(NotImplementedError):
            dsn_train.check_supported(dsn_train.build_parser().parse_args(bad))
    # every flag the model acts on reaches its option dict (--disc_freq / --gen_freq were parsed, accepted and then dropped)
    o2 = dsn_train.build_parser().parse_args(['--disc_freq', '2', '--gen_freq', '3', '--ragan', '--cat_or_sum', 'sum', '--filter', 'wavelet', '--w_col', '0.5'])
    mo = dsn_train.model_options(o2)
    assert (mo['disc_freq'], mo['gen_freq'], mo['ragan'], mo['cat_or_sum'], mo['filter'], mo['w_col']) == (2, 3, True, 'sum', 'wavelet', 0.5)
    import inspect, re
    dflt = re.search(r"o = dict\((.*?)\)\n", inspect.getsource(dsn_model.DSNModel.__init__), re.S).group(1)
    model_keys = set(re.findall(r"(\w+)=", dflt)) - {'vgg_seed'}   # (vgg_seed: no CLI flag)
    assert model_keys <= set(mo), model_keys - set(mo)   # every default of the model that a CLI flag backs is forwarded
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--lpips_rot_flip']))
    dsn_train.check_supported(dsn_train.build_parser().parse_args(['--wgan']))
    # the 12 (k_rot, flip, flip) draws of loss.py:155-168 map onto the 8 symmetries of the square, identity for (0, F, F)
    codes = {(k, a, b): dsn_model.symmetry_code(k, a, b) for k in (-1, 0, 1) for a in (False, True) for b in (False, True)}
    assert codes[(0, False, False)] == 0 and set(codes.values()) == set(range(8))
    x = torch.rand(1, 1, 5, 5)
    for (k, a, b), code in codes.items():
        t = torch.rot90(x, k, [2, 3])
        t = torch.flip(t, (2,)) if a else t
        t = torch.flip(t, (3,)) if b else t
        i, j = torch.meshgrid(torch.arange(5), torch.arange(5), indexing='ij')
        u, v = (j, i) if code & 1 else (i, j)
        u = 4 - u if code & 2 else u
        v = 4 - v if code & 4 else v
        assert torch.equal(x[0, 0][u, v], t[0, 0]), (k, a, b, code)
    # LambdaLR rule of train.py:154-157 against torch's scheduler
    m = dsn_model.DSNModel.__new__(dsn_model.DSNModel)
    m.opt = dict(num_epochs=10, num_decay_epochs=4, learning_rate=2e-4)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=2e-4)
    rule = lambda e: 1.0 if e < 6 else 1.0 - max(0.0, float(e - 6) / 4)
    sch = torch.optim.lr_scheduler.LambdaLR(opt, rule)
    for e in range(10):
        m.epoch = e
        assert abs(m.lr() - opt.param_groups[0]['lr']) < 1e-12
        opt.step()
        sch.step()


def test_fold_batchnorm_fsd_equals_eval_mode_network(golden_dir):
    """gan_nets.fold_batchnorm_fsd (what DSNModel.translate / ddm_of run for a BatchNorm discriminator): the conv-only network with the folded
    weights equals the FSD-Batch network in eval() mode -- on the reference's own test.tar weights and on random statistics"""
    import torch.nn as nn
    from dasr_amd.gan_nets import fold_batchnorm_fsd
    from oracle import dsn
    fx = np.load(os.path.join(golden_dir, 'dsn_fsd_batch_test_tar.npz'))
    sd_tar = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith('w/')}
    g = torch.Generator().manual_seed(3)
    D = dsn.Discriminator(5, 'Batch', 'gau')
    sd_rand = {k: v.clone() for k, v in D.state_dict().items()}
    for k in sd_rand:
        if k.endswith('running_mean'):
            sd_rand[k] = torch.randn(sd_rand[k].shape, generator=g) * 0.3
        elif k.endswith('running_var'):
            sd_rand[k] = torch.rand(sd_rand[k].shape, generator=g) + 0.2
        elif k.startswith(('net.net.3.', 'net.net.6.')) and k.endswith(('weight', 'bias')):
            sd_rand[k] = torch.randn(sd_rand[k].shape, generator=g) * 0.5 + (1.0 if k.endswith('weight') else 0.0)
    x = torch.rand(2, 3, 40, 36, generator=g)
    for sd in (sd_tar, sd_rand):
        D.load_state_dict(sd, strict=False)
        D.eval()
        plain = dsn.Discriminator(5, 'Instance', 'gau')
        plain.net.net[3], plain.net.net[6] = nn.Identity(), nn.Identity()
        plain.load_state_dict(fold_batchnorm_fsd({k: v for k, v in D.state_dict().items()}), strict=False)
        with torch.no_grad():
            want, got = D(x), plain(x)
        assert float((got - want).abs().max()) < 2e-6 * max(1.0, float(want.abs().max()))


def test_product_library_has_no_probes_and_no_wrong_result_kernels():
    """micro-benchmark probes live in libdasr_bench.so (include/dasr_hip_bench.h), the ablation instantiations of the dense conv
    kernel (results wrong on purpose) only exist under -DDASR_BENCH (libdasr_hip_ablate.so); neither is in the product library or its header"""
    from dasr_amd import build, _lib
    blob = open(build.build(), 'rb').read()
    assert blob.count(b'conv_glds_kernelILi1ELi67ELi4ELi0E') > 0          # the production instantiation is there ...
    for abl in (1, 2, 3, 4, 7, 8, 12, 15):                                   # ... none of the ablated ones
        assert blob.count(b'conv_glds_kernelILi1ELi67ELi4ELi%dE' % abl) == 0, abl
    for sym in (b'dasr_probe_mfma', b'dasr_probe_tile_sync', b'mfma_peak_kernel', b'tile_sync_kernel'):
        assert blob.count(sym) == 0, sym
    hdr = open(os.path.join(ROOT, 'include', 'dasr_hip.h')).read()
    assert 'dasr_probe_mfma' not in hdr and 'dasr_probe_tile_sync' not in hdr
    bench = build.build_bench()
    bhdr = open(os.path.join(ROOT, 'include', 'dasr_hip_bench.h')).read()
    declared = set(re.findall(r'^\s*int\s+(dasr_\w+)\s*\(', bhdr, flags=re.M))
    assert declared == set(_lib._BENCH_SIGS), declared ^ set(_lib._BENCH_SIGS)
    bb = open(bench, 'rb').read()
    for name in declared:
        assert bb.count(name.encode()) > 0, name


def test_perceptual_networks_refuse_to_run_seeded_without_opt_in():
    """the reference always runs pretrained LPIPS / VGG weights; a missing weight file must not silently become a random
    network.  load_lpips raises before touching the device; `allow_random_perceptual` is the explicit opt-in and labels the metric."""
    from dasr_amd import lpips, dsn_train
    with pytest.raises(FileNotFoundError):
        lpips.load_lpips({'path': {'lpips_alexnet': None, 'lpips_lin': None}}, 'cpu')
    with pytest.raises(FileNotFoundError):
        lpips.load_lpips({'path': {'lpips_alexnet': None, 'lpips_lin': '/x/alex.pth'}, 'allow_random_perceptual': False}, 'cpu')

    class _N:
        seeded = True
    assert lpips.lpips_label(_N()) == 'LPIPS(random)' and lpips.lpips_label(object()) == 'LPIPS'
    # DSN CLI: the dataset and the opt-in are checked before any model is built
    p = dsn_train.build_parser()
    with pytest.raises(NotImplementedError):
        dsn_train.check_supported(p.parse_args([]), have_loader=False)          # default --dataset df2k has no built-in loader
    dsn_train.check_supported(p.parse_args(['--dataset', 'synthetic']), have_loader=False)
    assert p.parse_args(['--allow_random_perceptual']).allow_random_perceptual is True and p.parse_args([]).allow_random_perceptual is False


def test_tensorboard_event_writer_roundtrip(tmp_path):
    """dasr_amd.tb_writer: the scalars / sample images of the reference drivers (codes/SRN/train.py:112-121,168,231-233; codes/DSN/train.py:245-270)
    as TensorBoard event files without the tensorboardX package: TFRecord framing with masked CRC32C, hand-encoded Event / Summary protobufs"""
    from dasr_amd import tb_writer as tw
    assert tw.crc32c(b'123456789') == 0xE3069283          # the CRC-32C check value (RFC 3720 B.4)
    assert tw.masked_crc(b'') == 0xA282EAD8
    w = tw.SummaryWriter(str(tmp_path / 'tb'))
    w.add_scalar('loss/l_g_pix', 0.125, 7)
    w.add_scalar('psnr', 27.5, 1000)
    img = torch.rand(3, 6, 10)
    w.add_image('train/train_samples_0', img, 1000)
    w.close()
    ev = tw.read_events(w.path)
    assert ev[0] == (7, 'loss/l_g_pix', 0.125) and ev[1] == (1000, 'psnr', 27.5)
    step, tag, (kind, h, wd, png) = ev[2]
    assert (step, tag, kind, h, wd) == (1000, 'train/train_samples_0', 'image', 6, 10) and png[:8] == b'\x89PNG\r\n\x1a\n'
    try:
        from PIL import Image
        import io
        import numpy as np
        back = np.asarray(Image.open(io.BytesIO(png)))
        want = (img.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
        assert back.shape == (6, 10, 3) and np.array_equal(back, want)
    except ImportError:
        pass


def test_dp_backward_never_hands_a_bucket_to_the_exchange_in_front_of_a_chained_launch():
    """rrdbnet._Plan.run_backward_dp: a chained launch needs every workgroup slot of the device, so no collective may be in flight while
    it runs.  The gradient bucket that is complete in front of a segment with a chained launch is held back and handed over together with that segment's own
    bucket; every other bucket goes out right behind its segment (overlap with the weight-gradient launches); the call ends with dp.wait().  Pure host
    logic: stub segments / stub group, no device."""
    from types import SimpleNamespace as NS
    from dasr_amd import _lib
    from dasr_amd.rrdbnet import _Plan
    events = []

    def seg(name, kinds):
        return NS(ops=[NS(op=k) for k in kinds], run=lambda name=name: events.append(('run', name)))

    CONV, CHAIN, WG = _lib.OP_CONV, _lib.OP_CONV_CHAIN, _lib.OP_WGRAD
    segs = [(seg('hr_tail', [CONV, WG]), (900, 1000)), (seg('chain+wgrad0', [CONV, CHAIN, CONV, WG]), (600, 900)), (seg('wgrad1', [WG]), (300, 600)),
            (seg('fea', [CONV, WG]), (0, 300))]
    plan = NS(bwd_segments=lambda: segs)
    dp = NS(reduce_async=lambda sl: events.append(('reduce', (sl.start, sl.stop))), wait=lambda: events.append(('wait',)))

    class G:
        def __getitem__(self, sl):
            return sl
    _Plan.run_backward_dp(plan, dp, G())
    assert events == [('run', 'hr_tail'), ('run', 'chain+wgrad0'), ('reduce', (900, 1000)), ('reduce', (600, 900)), ('run', 'wgrad1'), ('reduce', (300, 600)),
                      ('run', 'fea'), ('reduce', (0, 300)), ('wait',)], events
    # without a chained launch in the list: every bucket right behind its segment
    events.clear()
    segs[1] = (seg('per_layer', [CONV, CONV, WG]), (600, 900))
    _Plan.run_backward_dp(plan, dp, G())
    assert events[:4] == [('run', 'hr_tail'), ('reduce', (900, 1000)), ('run', 'per_layer'), ('reduce', (600, 900))] and events[-1] == ('wait',)


def test_chain_split_rule():
    """RRDBNetHIP.chain_split: which batches run their trunk as chained launches, and as how many image sub-batches (host rule; a whole 256-CU device assumed)"""
    from types import SimpleNamespace as NS
    from dasr_amd.rrdbnet import RRDBNetHIP
    net = NS(chain=True, _cus=256, chain_form='layer', nf=64)
    f = lambda N, h, w: RRDBNetHIP.chain_choice(net, N, h, w)[1]
    assert (f(16, 128, 128), f(8, 256, 128), f(32, 64, 128), f(32, 128, 128), f(24, 128, 256), f(64, 128, 128)) == (1, 1, 1, 2, 3, 4)
    assert (f(16, 64, 64), f(12, 128, 128), f(20, 128, 128), f(24, 128, 128), f(80, 128, 128), f(4, 256, 256)) == (0, 0, 0, 0, 0, 0)   # too few / not 512 k / images per XCD / > 4
    g = lambda ns: RRDBNetHIP.chain_choice(ns, 16, 128, 128)[1]
    assert g(NS(chain=False, _cus=256, chain_form='layer')) == 0 and g(NS(chain=True, _cus=64, chain_form='layer')) == 0
    # the input-stationary form (round 6, dasr_rdb_chain): one launch; whole images per XCD, N * tiles a multiple of 256 (<= 8 tiles per workgroup), tiles per image divides 32
    net = NS(chain=True, _cus=256, chain_form='is', nf=64)
    assert (f(16, 128, 128), f(8, 128, 128), f(32, 128, 128), f(64, 128, 128), f(16, 64, 128), f(64, 64, 64), f(8, 128, 112), f(16, 64, 64), f(16, 32, 32)) == (1,) * 9
    assert (f(8, 256, 128), f(24, 128, 256), f(12, 128, 128), f(4, 256, 256), f(128, 128, 128), f(16, 192, 192)) == (0, 0, 0, 0, 0, 0)   # 64 tiles per image / ... / N % 8 / > 8 per workgroup / 72 tiles per image
    # launch geometry (workgroups per XCD, tiles per workgroup): mirrors dasr_rdb_chain
    G = RRDBNetHIP.is_geometry
    assert (G(16, 32), G(8, 32), G(24, 32), G(64, 32), G(16, 2), G(32, 2), G(16, 8), G(64, 8), G(16, 16)) == ((32, 2), (32, 1), (32, 3), (32, 8), (4, 1), (8, 1), (16, 1), (32, 2), (32, 1))
    assert G(128, 32) is None and G(8, 64) is None and G(12, 32) is None
    # ... and the tile height (16 / 8 / 4 rows): tiles per workgroup x measured chain time of one tile (the shipped 16 crops of 32 x 32: 128 workgroups of 4-row tiles;
    # 24 x 64 x 64 keeps 16-row tiles: the finer ones would give every workgroup three of them)
    Lc = RRDBNetHIP.is_launch
    assert (Lc(16, 128, 128), Lc(8, 128, 128), Lc(16, 32, 32), Lc(32, 32, 32), Lc(16, 64, 64), Lc(8, 64, 64), Lc(16, 64, 128), Lc(8, 32, 32), Lc(24, 64, 64), Lc(16, 48, 48), Lc(32, 64, 64)) == \
        ((16, 32, 2), (16, 32, 1), (4, 16, 1), (4, 32, 1), (8, 32, 1), (4, 32, 1), (16, 32, 1), (4, 8, 1), (16, 24, 1), (8, 24, 1), (16, 32, 1))
    assert Lc(8, 256, 128) is None and Lc(16, 192, 192) is None and Lc(12, 128, 128) is None
    assert g(NS(chain=True, _cus=256, chain_form='is', nf=32)) == 0
    # the default: the layer form where it fits, else the input-stationary form; the refusal names its clause (the decisions at batch 12 / 20 / 24 x 128^2, 16 x 192^2)
    net = NS(chain=True, _cus=256, chain_form='auto', nf=64)
    c = lambda N, h, w: RRDBNetHIP.chain_choice(net, N, h, w)[:2]
    assert (c(16, 128, 128), c(32, 128, 128), c(8, 256, 128), c(24, 128, 256)) == (('layer', 1), ('layer', 2), ('layer', 1), ('layer', 3))
    assert (c(8, 128, 128), c(24, 128, 128), c(40, 128, 128), c(16, 64, 128), c(32, 64, 64), c(16, 32, 32)) == (('is', 1),) * 6   # (16 crops of 32 x 32: the reference's shipped shape)
    assert (c(12, 128, 128), c(20, 128, 128), c(16, 192, 192)) == ((None, 0),) * 3
    why = RRDBNetHIP.chain_choice(net, 16, 192, 192)[2]
    assert 'layer form' in why and 'input-stationary form' in why and '72' in why   # 12 x 6 tiles per image: more than the 32 workgroups of an XCD, 1152 tiles not 512 k
    assert 'DASR_CHAIN=0' in RRDBNetHIP.chain_choice(NS(chain=False), 16, 128, 128)[2]


def test_crc32c_fast_path_equals_the_byte_loop():
    """tb_writer.crc32c: records above 16 KB go through the chunk-parallel numpy path (the per-byte loop cost seconds per image); same value as
    the byte loop at every size class, and the CRC-32C check value of '123456789'"""
    from dasr_amd import tb_writer as t
    assert t.crc32c(b'123456789') == 0xE3069283
    rng = random.Random(7) if 'random' in globals() else __import__('random').Random(7)
    for n in (0, 1, 4095, 16383, 16384, 16385, 100003, 300000):
        d = bytes(rng.getrandbits(8) for _ in range(n))
        assert t.crc32c(d) == (t._crc_bytes(0xFFFFFFFF, d) ^ 0xFFFFFFFF), n
    assert t.crc32c(bytearray(b'x' * 20000)) == t.crc32c(b'x' * 20000)   # (any bytes-like object)
