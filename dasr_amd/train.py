"""`python -m dasr_amd.train -opt X.json` -- the SRN training driver (reference: codes/SRN/train.py:20-245).

Same loop order as the reference: update_learning_rate -> feed_data -> optimize_parameters, log every
logger.print_freq, checkpoint every logger.save_checkpoint_freq (model.save + save_training_state), final
model.save('latest').  Data: the reference's cv2/lmdb datasets are CPU-side IO and out of the hot-path scope
(SURVEY.md section 8, row 10); this driver accepts any iterable of batch dicts with the reference's keys and ships a
synthetic dataset (`datasets.train.mode: "synthetic"`) used by the benchmark and the tests.  Under
torch.distributed.run every rank takes its shard of each batch (dasr_amd.dist.shard_minibatch).
"""
import argparse
import logging
import math
import os
import random
import time

import numpy as np
import torch

from . import options as option
from .dist import DataParallelGroup, shard_minibatch
from .models import create_model


class SyntheticValDataset:
    """LR/HR validation pairs (batch 1, reference keys incl. the *_path strings): HR = smooth random image, LR = 4x4 box average"""

    def __init__(self, ds_opt, scale):
        self.n = int(ds_opt.get('n_images') or 4)
        self.h = int(ds_opt.get('LR_size') or 32)
        self.scale = scale
        self.seed = int(ds_opt.get('seed') or 4321)
        self.opt = ds_opt

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        s, h = self.scale, self.h
        for i in range(self.n):
            base = torch.rand(1, 3, h // 4 + 2, h // 4 + 2, generator=g)
            hr = torch.nn.functional.interpolate(base, size=(h * s, h * s), mode='bilinear', align_corners=False).clamp(0, 1)
            lr = torch.nn.functional.avg_pool2d(hr, s)
            yield {'LR': lr, 'HR': hr, 'LR_path': ['synthetic/val_%03d.png' % i], 'HR_path': ['synthetic/val_%03d.png' % i]}


class SyntheticDataset:
    """Fixed-seed random crops with the reference batch-dict keys."""

    def __init__(self, ds_opt, scale, model):
        self.n = int(ds_opt['batch_size'])
        hr = int(ds_opt['HR_size'] or 128)
        self.h = hr // scale
        self.scale = scale
        self.len = int(ds_opt.get('n_batches') or 1000)
        self.dasr = model not in ('sr',)
        self.seed = int(ds_opt.get('seed') or 1234)

    def __len__(self):
        return self.len

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        n, h, H = self.n, self.h, self.h * self.scale
        for _ in range(self.len):
            if self.dasr:
                yield {'LR_fake': torch.rand(n, 3, h, h, generator=g), 'LR_real': torch.rand(n, 3, h, h, generator=g),
                       'HR': torch.rand(n, 3, H, H, generator=g), 'HR_unpair': torch.rand(n, 3, H, H, generator=g),
                       'fake_w': torch.rand(n, 1, h, h, generator=g)}
            else:
                yield {'LR': torch.rand(n, 3, h, h, generator=g), 'HR': torch.rand(n, 3, H, H, generator=g)}


def create_dataset(ds_opt, opt):
    mode = ds_opt['mode']
    if mode == 'synthetic':
        if ds_opt.get('phase') in ('val', 'test'):
            return SyntheticValDataset(ds_opt, opt['scale'])
        return SyntheticDataset(ds_opt, opt['scale'], opt['model'])
    if mode == 'LRHR_wavelet_unpair_fake_weights_EQ' and ds_opt.get('phase', 'train') == 'train':
        # the DASR training set (data/__init__.py:35-36): images resident in HBM, batches assembled by dasr_gather_crops
        from .data import DeviceUnpairedDataset
        return DeviceUnpairedDataset(ds_opt, opt['scale'])
    if mode == 'LRHR' and ds_opt.get('phase', 'train') == 'train' and ds_opt.get('dataroot_LR'):
        from .data import DevicePairedDataset
        return DevicePairedDataset(ds_opt, opt['scale'])
    raise NotImplementedError('Dataset [{:s}] is not recognized (the cv2/lmdb loaders of the reference stay on its side of '
                              'the boundary; feed their batch dicts to the trainer object).'.format(str(mode)))


def validate(model, val_set, opt, current_step, logger):
    """validation pass of codes/SRN/train.py:174-235: test() per image, SR image saved, PSNR on the `scale`-pixel-cropped uint8 images"""
    from . import util
    avg_psnr, avg_lpips, idx = 0.0, 0.0, 0
    for val_data in val_set:
        idx += 1
        img_name = os.path.splitext(os.path.basename(val_data['LR_path'][0]))[0]
        img_dir = os.path.join(opt['path']['val_images'], img_name)
        util.mkdir(img_dir)
        model.feed_data(val_data, False)
        model.test()
        visuals = model.get_current_visuals()
        sr_img = util.tensor2img(visuals['SR'])
        log_info = '{}'.format(val_data['HR_path'][0].split('/')[-1])
        if opt['val_lpips']:      # train.py:194-197
            avg_lpips += float(visuals['LPIPS'])
            log_info += '         {}:{:.3f}'.format(model.lpips_label, float(visuals['LPIPS']))
        logger.info(log_info)
        util.save_img(sr_img, os.path.join(img_dir, '{:s}_{:d}.png'.format(img_name, current_step)))
        if 'HR' in visuals:
            gt_img = util.tensor2img(visuals['HR'])
            c = opt['scale']
            avg_psnr += util.calculate_psnr((sr_img / 255.)[c:-c, c:-c, :] * 255, (gt_img / 255.)[c:-c, c:-c, :] * 255)
    avg_psnr = avg_psnr / max(idx, 1)
    logger.info('# Validation # PSNR: {:.4e}'.format(avg_psnr))
    if opt['val_lpips']:
        return avg_psnr, avg_lpips / max(idx, 1)
    return avg_psnr


def setup_logger(name, root, phase, level=logging.INFO, screen=False):
    lg = logging.getLogger(name)
    fmt = logging.Formatter('%(asctime)s.%(msecs)03d - %(levelname)s: %(message)s', datefmt='%y-%m-%d %H:%M:%S')
    os.makedirs(root, exist_ok=True)
    fh = logging.FileHandler(os.path.join(root, phase + '_{}.log'.format(time.strftime('%y%m%d-%H%M%S'))), mode='w')
    fh.setFormatter(fmt)
    lg.setLevel(level)
    lg.addHandler(fh)
    if screen:
        sh = logging.StreamHandler()
        sh.setFormatter(fmt)
        lg.addHandler(sh)


def training_samples(model, opt, current_step, tb_logger, logger, n_samples=5):
    """the `save_tsamples` branch of the reference driver (codes/SRN/train.py:123-172): five random (fake LR, real LR, HR) triples, centre-cropped to
    128 x 128 LR / 512 x 512 HR, go through model.test(tsamples=True); per triple one image [fake SR | HR | real SR] over [their high-frequency
    views] is written to TensorBoard (`train/train_samples_{i}`) -- and, here, also as a PNG under <experiments_root>/tsamples.  The file lists
    come from datasets.train.dataroot_{fake_LR, real_LR, HR} like the reference's (PIL); a 'synthetic' train set draws seeded random crops."""
    ds = opt['datasets']['train']
    dirs = [ds.get('dataroot_fake_LR'), ds.get('dataroot_real_LR'), ds.get('dataroot_HR')]
    from .tb_writer import png_encode
    out_dir = os.path.join(opt['path']['experiments_root'] or '.', 'tsamples')
    os.makedirs(out_dir, exist_ok=True)
    lists = None
    if all(d and os.path.isdir(d) for d in dirs):
        lists = [os.listdir(d) for d in dirs]   # (unsorted, like the reference)
    for i in range(n_samples):
        if lists is not None:
            from PIL import Image
            idx = np.random.choice(range(len(lists[0])))
            fake_LR, real_LR, HR = (np.array(Image.open(os.path.join(d, l[idx]))) for d, l in zip(dirs, lists))
        else:
            g = torch.Generator().manual_seed(int(np.random.randint(0, 2 ** 31 - 1)))
            fake_LR, real_LR, HR = ((torch.rand(s, s, 3, generator=g) * 255).to(torch.uint8).numpy() for s in (128, 128, 512))
        crop = lambda a, r: a[a.shape[0] // 2 - r:a.shape[0] // 2 + r, a.shape[1] // 2 - r:a.shape[1] // 2 + r, :]
        fake_LR, real_LR, HR = crop(fake_LR, 64), crop(real_LR, 64), crop(HR, 256)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(np.transpose(a[:, :, :3], (2, 0, 1)))).float().unsqueeze(0) / 255
        model.feed_data({'LR': torch.cat([t(fake_LR), t(real_LR)], 0), 'HR': t(HR)}, False)
        model.test(tsamples=True)
        v = model.get_current_visuals(tsamples=True)
        image_1 = torch.cat([v['SR'][0], v['HR'], v['SR'][1]], dim=2).clamp(0, 1)
        image_2 = torch.cat([v['hf'][0], v['HR_hf'][0], v['hf'][1]], dim=2).clamp(0, 1)
        image = torch.cat([image_1, image_2], dim=1)
        if tb_logger is not None:
            tb_logger.add_image('train/train_samples_{}'.format(i), image, current_step)
        with open(os.path.join(out_dir, '{:d}_{:d}.png'.format(current_step, i)), 'wb') as f:
            f.write(png_encode(image)[0])
    logger.info('Saved training Samples')


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True, help='Path to option JSON file.')
    opt = option.dict_to_nonedict(option.parse(ap.parse_args(argv).opt, is_train=True))
    dp = DataParallelGroup() if int(os.environ.get('WORLD_SIZE', '1')) > 1 else None
    rank = dp.rank if dp else 0
    if dp:
        torch.cuda.set_device(dp.device_index)
    resume_state = None
    if opt['path']['resume_state']:
        resume_state = torch.load(opt['path']['resume_state'], weights_only=False)
    elif rank == 0:
        root = opt['path']['experiments_root']
        if os.path.exists(root):
            os.rename(root, root + '_archived_' + time.strftime('%y%m%d-%H%M%S'))
        for k, p in opt['path'].items():
            if k != 'experiments_root' and 'pretrain_model' not in k and 'resume' not in k and p:
                os.makedirs(p, exist_ok=True)
    if dp:
        dp.barrier()
    setup_logger('base', opt['path']['log'], 'train_rank%d' % rank if dp else 'train', screen=(rank == 0))
    logger = logging.getLogger('base')
    if resume_state:
        logger.info('Resuming training from epoch: {}, iter: {}.'.format(resume_state['epoch'], resume_state['iter']))
        option.check_resume(opt)
    logger.info(option.dict2str(opt))
    seed = opt['train']['manual_seed']
    if seed is None:
        seed = random.randint(1, 10000)
    logger.info('Random seed: {}'.format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)

    # tensorboard scalars / images (train.py:57-59): written by dasr_amd.tb_writer (no tensorboardX in an air-gapped image); same tags as the reference
    tb_logger = None
    if opt['use_tb_logger'] and 'debug' not in opt['name'] and rank == 0:
        from .tb_writer import SummaryWriter
        tb_logger = SummaryWriter(os.path.join(opt['path']['tb_logger'] or os.path.join(opt['path']['root'] or '.', 'SRN_tb_logger'), opt['name']))
    train_set = create_dataset(opt['datasets']['train'], opt)
    val_set = create_dataset(opt['datasets']['val'], opt) if opt['datasets'].get('val') else None
    if rank == 0:
        setup_logger('val', opt['path']['log'], 'val')
    total_iters = int(opt['train']['niter'])
    total_epochs = int(math.ceil(total_iters / max(1, len(train_set))))
    model = create_model(opt)
    if dp:
        model.dp = dp
        for net in model.networks():
            dp.broadcast_params(net.params.flat)
            net.repack()
    current_step, start_epoch = 0, 0
    if resume_state:
        start_epoch, current_step = resume_state['epoch'], resume_state['iter']
        model.resume_training(resume_state, opt['train'])
    logger.info('Start training from epoch: {:d}, iter: {:d}'.format(start_epoch, current_step))
    for epoch in range(start_epoch, total_epochs):
        for batch in train_set:
            current_step += 1
            if current_step > total_iters:
                break
            model.update_learning_rate()
            if dp:
                batch = shard_minibatch(batch, dp.rank, dp.world)
            model.feed_data(batch, True)
            model.optimize_parameters(current_step)
            if current_step % opt['logger']['print_freq'] == 0 and dp and hasattr(model, 'sync_error_words'):
                model.sync_error_words()   # (collective, every rank) every rank holds the MAX of every rank's device error words from here on: all gate / raise together
            if current_step % opt['logger']['print_freq'] == 0 and rank != 0:
                model.check_finite()   # rank 0 checks inside get_current_log: all ranks raise together (the flag is set behind the all-reduce)
            if current_step % opt['logger']['print_freq'] == 0 and rank == 0:
                msg = '<epoch:{:3d}, iter:{:8,d}, lr:{:.3e}> '.format(epoch, current_step, model.get_current_learning_rate())
                for k, v in model.get_current_log().items():
                    msg += '{:s}: {:.4e} '.format(k, v)
                    if tb_logger is not None:
                        tb_logger.add_scalar(k, v, current_step)
                logger.info(msg)
            if opt['train']['save_tsamples'] and current_step % opt['train']['save_tsamples'] == 0 and rank == 0:
                if not hasattr(model, 'filter_high'):
                    raise NotImplementedError('save_tsamples needs the DASR trainer (get_current_visuals(tsamples=True), DASR_model.py:349-366)')
                training_samples(model, opt, current_step, tb_logger, logger)
            if val_set is not None and opt['train']['val_freq'] and current_step % opt['train']['val_freq'] == 0 and rank == 0:
                res = validate(model, val_set, opt, current_step, logger)
                if opt['val_lpips']:   # train.py:226-228
                    logging.getLogger('val').info('<epoch:{:3d}, iter:{:8,d}> psnr: {:.4e}, {}: {:.4f}'.format(epoch, current_step, res[0], model.lpips_label, res[1]))
                else:
                    logging.getLogger('val').info('<epoch:{:3d}, iter:{:8,d}> psnr: {:.4e}'.format(epoch, current_step, res))
                if tb_logger is not None:   # train.py:231-233
                    tb_logger.add_scalar('psnr', res[0] if opt['val_lpips'] else res, current_step)
                    tb_logger.add_scalar('LPIPS', res[1] if opt['val_lpips'] else 0.0, current_step)
            if current_step % opt['logger']['save_checkpoint_freq'] == 0 and rank == 0:
                logger.info('Saving models and training states.')
                model.save(current_step)
                model.save_training_state(epoch, current_step)
    if rank == 0:
        logger.info('Saving the final model.')
        model.save('latest')
        logger.info('End of training.')
        if tb_logger is not None:
            tb_logger.close()


if __name__ == '__main__':
    main()
