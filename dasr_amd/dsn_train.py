"""`python -m dasr_amd.dsn_train [flags]` -- the DSN training driver (reference: codes/DSN/train.py:24-376).

Same flags and defaults as the reference's argparse block (train.py:24-72) and the same loop: per iteration one
DSNModel.iteration(hr, bicubic_lr, real_lr); schedulers step once per epoch (train.py:287-288); every
`save_model_interval` epochs the `.tar` checkpoint dict of train.py:357-376 is written to
<save_path>/checkpoints/iteration_<n>.tar and last_iteration.tar.  Accepted-but-unsupported choices fail the way the
reference does for unknown strings (NotImplementedError).  --norm_layer Batch runs with every discriminator and with --wgan (round 6; --wgan with --ragan: round 5).  --ragan is supported, also under data parallelism (the batch means are all-reduced between the loss stages).  --per_type LPIPS (the reference
default) runs LPIPS(alex) with weights from --lpips_alexnet / --lpips_lin, --per_type VGG with --vgg_path; the pretrained files cannot be
downloaded offline, a missing file is an error unless --allow_random_perceptual opts into a seeded random network.  Data: `--dataset aim2019 |
ntire2020 | realsr | camerasr` read the image folders `--paths` (the reference's codes/paths.yml, train.py:82-83) names for `--artifacts`, through
dasr_amd.dsn_data (the reference's Train_/Val_Deresnet_Dataset on PIL tensors, no torchvision); any iterable of (hr, bicubic_lr, real_lr) tuples
passed to main(loader=, val_loader=) works too, and `--dataset synthetic` ships fixed-seed random crops for benchmarks and tests.  Every
`val_interval` epochs the validation pass of train.py:293-355 runs over the paired validation folders (mse, psnr, rgb / mean / perceptual / colour
errors as `val/*` scalars), every `val_img_interval` epochs it also writes the `val/target_fake_crop_low_high_<i>` image strips.
"""
import argparse
import json
import logging
import os

import torch

from .dist import DataParallelGroup
from .dsn_model import DSNModel


def build_parser():
    p = argparse.ArgumentParser(description='Train Downscaling Models')
    p.add_argument('--upscale_factor', default=4, type=int, choices=[4])
    p.add_argument('--crop_size', default=256, type=int)
    p.add_argument('--crop_size_val', default=256, type=int)
    p.add_argument('--batch_size', default=4, type=int)
    p.add_argument('--num_workers', default=6, type=int)
    p.add_argument('--num_epochs', default=400, type=int)
    p.add_argument('--num_decay_epochs', default=150, type=int)
    p.add_argument('--learning_rate', default=0.0001, type=float)
    p.add_argument('--adam_beta_1', default=0.5, type=float)
    p.add_argument('--val_interval', default=5, type=int)
    p.add_argument('--val_img_interval', default=5, type=int)
    p.add_argument('--save_model_interval', default=5, type=int)
    p.add_argument('--artifacts', default='tdsr', type=str)
    p.add_argument('--dataset', default='df2k', type=str)   # reference default (train.py:38), which its own loop cannot unpack; built in: the four
    #                                                          Train_Deresnet_Dataset branches (aim2019, ntire2020, realsr, camerasr) and 'synthetic'
    p.add_argument('--flips', dest='flips', action='store_true')
    p.add_argument('--rotations', dest='rotations', action='store_true')
    p.add_argument('--num_res_blocks', default=8, type=int)
    p.add_argument('--ragan', dest='ragan', action='store_true')
    p.add_argument('--wgan', dest='wgan', action='store_true')
    p.add_argument('--no_highpass', dest='highpass', action='store_false')
    p.add_argument('--kernel_size', default=5, type=int)
    p.add_argument('--no_per_loss', dest='use_per_loss', action='store_false')
    p.add_argument('--lpips_rot_flip', dest='lpips_rot_flip', action='store_true')
    p.add_argument('--per_type', default='LPIPS', type=str)   # reference default (train.py:54)
    p.add_argument('--disc_freq', default=1, type=int)
    p.add_argument('--gen_freq', default=1, type=int)
    p.add_argument('--w_col', default=1, type=float)
    p.add_argument('--w_tex', default=0.005, type=float)
    p.add_argument('--w_per', default=0.01, type=float)
    p.add_argument('--checkpoint', default=None, type=str)
    p.add_argument('--save_path', default=None, type=str)
    p.add_argument('--generator', default='DeResnet', type=str)
    p.add_argument('--discriminator', default='FSD', type=str)
    p.add_argument('--filter', default='gau', type=str)
    p.add_argument('--cat_or_sum', default='cat', type=str)
    p.add_argument('--norm_layer', default='Instance', type=str)
    p.add_argument('--no_saving', dest='saving', action='store_false')
    p.add_argument('--debug', dest='debug', action='store_true')
    # additions of this build
    p.add_argument('--iters_per_epoch', default=100, type=int, help='synthetic dataset: iterations per epoch')
    p.add_argument('--paths', default='../paths.yml', type=str, help="yaml file with the dataset folders (the reference reads '../paths.yml', train.py:82)")
    p.add_argument('--vgg_path', default=None, type=str, help='torchvision vgg16 state_dict for --per_type VGG')
    p.add_argument('--lpips_alexnet', default=None, type=str, help='torchvision alexnet state_dict for --per_type LPIPS')
    p.add_argument('--lpips_lin', default=None, type=str, help="the reference's codes/PerceptualSimilarity/models/weights/v0.1/alex.pth")
    p.add_argument('--allow_random_perceptual', action='store_true',
                   help='run the perceptual term on a SEEDED RANDOM network when the pretrained weight files are not supplied (the reference always uses pretrained weights)')
    return p


def check_supported(o, have_loader=True):
    """everything that cannot run is refused HERE, before any model is built or any parameter is broadcast"""
    from .dsn_data import DERESNET_DATASETS
    if not have_loader and o.dataset != 'synthetic' and o.dataset not in DERESNET_DATASETS:
        raise NotImplementedError("dataset [%s]: built in are aim2019 / ntire2020 / realsr / camerasr (image folders from --paths, the reference's "
                                  "Train_Deresnet_Dataset branches) and 'synthetic'; or pass a loader of (hr, bicubic_lr, real_lr) batches to main()" % o.dataset)
    if o.generator not in ('DeResnet', 'DSGAN'):
        raise NotImplementedError('Generator model [{:s}] not recognized'.format(o.generator))
    if o.discriminator.lower() not in ('fsd', 'nld_s1', 'nld_s2'):
        raise NotImplementedError('Discriminator architecture [{:s}] not recognized'.format(o.discriminator))
    if not o.highpass or o.cat_or_sum not in ('cat', 'sum') or o.norm_layer not in ('Instance', 'Batch'):
        raise NotImplementedError('DSN on MI355X covers: high-pass front end, wavelet bands cat / sum, Instance or Batch norm')
    if o.disc_freq < 1 or o.gen_freq < 1:
        raise ValueError('--disc_freq / --gen_freq must be >= 1')


class SyntheticCrops:
    """(hr [b,3,c,c], bicubic_lr [b,3,c/4,c/4], real_lr [b,3,c/4,c/4]) in [0,1] (data_loader.py:50-54)"""

    def __init__(self, batch, crop, n, seed=1234):
        self.b, self.c, self.n, self.seed = batch, crop, n, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        b, c = self.b, self.c
        for _ in range(self.n):
            yield torch.rand(b, 3, c, c, generator=g), torch.rand(b, 3, c // 4, c // 4, generator=g), torch.rand(b, 3, c // 4, c // 4, generator=g)


def model_options(o):
    """argparse namespace -> DSNModel option dict: every flag the model acts on (a flag missing here silently keeps the model default)"""
    return dict(n_res_blocks=o.num_res_blocks, kernel_size=o.kernel_size, filter=o.filter, norm_layer=o.norm_layer, discriminator=o.discriminator,
                learning_rate=o.learning_rate, adam_beta_1=o.adam_beta_1, w_col=o.w_col, w_tex=o.w_tex, w_per=o.w_per if o.use_per_loss else 0.0,
                per_type=o.per_type, generator=o.generator, vgg_path=o.vgg_path, lpips_alexnet=o.lpips_alexnet, lpips_lin=o.lpips_lin, num_epochs=o.num_epochs,
                num_decay_epochs=o.num_decay_epochs, upscale_factor=o.upscale_factor, ragan=o.ragan, allow_random_perceptual=o.allow_random_perceptual,
                cat_or_sum=o.cat_or_sum, disc_freq=o.disc_freq, gen_freq=o.gen_freq, lpips_rot_flip=o.lpips_rot_flip, wgan=o.wgan)   # train.py:55-56, 229, 251


def validate(model, val_loader, o, tb, epoch, log):
    """train.py:293-355: every validation pair through the generator (clamped to [0, 1]); mse / psnr / rgb / mean / perceptual / colour errors
    averaged over the set as `val/*` scalars; on `val_img_interval` epochs strips of (target, fake, random LR crop, low-pass, high-pass of the fake)
    in the reference's 400 x 400 display transform, five validation images per grid"""
    from .dsn_data import display_transform
    from .util import make_grid
    dev = model.device
    with_images = epoch % o.val_img_interval == 0 and epoch != 0
    sums, count, strips = None, 0, []
    for hr, bic, disc, target in val_loader:
        inp = bic if o.generator == 'DSGAN' else hr
        fake = model.generate(inp.to(dev)).clamp(0, 1)
        target = target.to(dev)
        m = model.validation_metrics(fake, target)
        vals = torch.stack([v.float() for v in m.values()])
        sums = vals if sums is None else sums + vals
        keys = list(m.keys())
        count += 1
        if with_images:
            strips += [display_transform(t[0]) for t in (target, fake, disc, model.filter_low(fake), model.filter_high(fake))]
    if not count:
        return {}
    avg = dict(zip(keys, (sums / count).tolist()))   # one host sync per validation pass
    log.info('[validation] epoch %d iter %d ' % (epoch, model.iteration_count) + ' '.join('%s: %.4e' % kv for kv in avg.items()))
    if tb is not None:
        for k, v in avg.items():
            tb.add_scalar('val/' + k, v, model.iteration_count)
        if with_images:
            per_grid = 5 * 5   # n_val_images (the five views) x 5 validation images per grid: torch.chunk(all, count // 25) as train.py:347-353 has it
            for index, chunk in enumerate(torch.chunk(torch.stack(strips), max(1, len(strips) // per_grid))):   # (fewer than five images: one grid)
                tb.add_image('val/target_fake_crop_low_high_' + str(index), make_grid(chunk, nrow=5, padding=5), model.iteration_count)
    return avg


def main(argv=None, loader=None, val_loader=None):
    o = build_parser().parse_args(argv)
    check_supported(o, have_loader=loader is not None)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(message)s')
    log = logging.getLogger('base')
    torch.manual_seed(0)  # train.py:76
    dp = DataParallelGroup() if int(os.environ.get('WORLD_SIZE', '1')) > 1 else None
    rank = dp.rank if dp else 0
    world = dp.world if dp else 1
    if dp:
        torch.cuda.set_device(dp.device_index)
    if o.debug:
        o.num_epochs, o.iters_per_epoch = min(o.num_epochs, 2), min(o.iters_per_epoch, 3)
    opt = model_options(o)
    model = DSNModel(opt)
    if dp:
        model.dp = dp
        for net in model.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    start_epoch = 1
    if o.checkpoint:
        model.load(o.checkpoint)
        start_epoch = model.epoch + 1
        log.info('Continuing training at epoch %d' % start_epoch)
    if loader is None and o.dataset == 'synthetic':
        loader = SyntheticCrops(o.batch_size // world, o.crop_size, o.iters_per_epoch, seed=1234 + rank)
    elif loader is None:   # train.py:81-115: image folders from paths.yml
        from . import dsn_data
        train_set, val_set = dsn_data.make_datasets(o, dsn_data.load_paths(o.paths))
        loader = dsn_data.make_loader(train_set, o.batch_size, True, o.num_workers, seed=0, rank=rank, world=world)
        if val_loader is None:
            val_loader = dsn_data.make_loader(val_set, 1, False, min(1, o.num_workers))
    save_path = o.save_path or os.path.join('experiments', 'dsn_' + o.filter)
    if o.saving and rank == 0:
        os.makedirs(os.path.join(save_path, 'checkpoints'), exist_ok=True)
        with open(os.path.join(save_path, 'commandline_args.txt'), 'w') as f:
            json.dump(o.__dict__, f, indent=2)
    tb = None
    if o.saving and rank == 0:   # train.py:138,245-270: tensorboardX scalars under <save_path>/logs -- same tags, written by dasr_amd.tb_writer, once per
        from .tb_writer import SummaryWriter   # epoch (the trainer keeps its scalars on the device between log points instead of 7 .item() syncs per iteration)
        tb = SummaryWriter(os.path.join(save_path, 'logs'))
    dev = model.device
    for epoch in range(start_epoch, o.num_epochs + 1):
        for hr, bic, real in loader:
            model.iteration(hr.to(dev, non_blocking=True), bic.to(dev, non_blocking=True), real.to(dev, non_blocking=True))
        model.end_epoch()
        if rank != 0:
            model.check_finite()   # rank 0 checks inside get_current_log below: all ranks raise together
        if rank == 0:
            lg = model.get_current_log()
            log.info('[%d/%d] iter %d lr %.3e ' % (epoch, o.num_epochs, model.iteration_count, model.lr()) +
                     ' '.join('%s: %.4e' % kv for kv in lg.items()))
            if tb is not None:
                for k, v in lg.items():
                    tb.add_scalar(k, v, model.iteration_count)
                tb.add_scalar('param/learning_rate', model.lr(), epoch)   # train.py:289-290
            if val_loader is not None and (epoch % o.val_interval == 0 or epoch % o.val_img_interval == 0):
                validate(model, val_loader, o, tb, epoch, log)
            if o.saving and epoch % o.save_model_interval == 0:
                model.save(os.path.join(save_path, 'checkpoints', 'iteration_{}.tar'.format(model.iteration_count)))
                model.save(os.path.join(save_path, 'checkpoints', 'last_iteration.tar'))
    if tb is not None:
        tb.close()
    return model


if __name__ == '__main__':
    main()
