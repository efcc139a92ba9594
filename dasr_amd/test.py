"""`python -m dasr_amd.test -opt X.json` -- SRN inference / evaluation driver (reference: codes/SRN/test.py:17-138).

For every dataset of the option file: feed_data -> test() -> get_current_visuals -> SR image saved under
<results_root>/<dataset name>/imgs, and, when HR is present, PSNR / SSIM on the `scale`-pixel-cropped uint8 images in RGB and on the
Y channel, per image and averaged (same log lines as the reference).  `chop: true` runs the quadrant inference.  `val_lpips: true` adds the
LPIPS(alex) distance of the 8-bit images (test.py:88-128; weights from `path.lpips_alexnet` / `path.lpips_lin`, seeded when absent).
`save_RealorFake` needs the discriminator visual, not available here: NotImplementedError.  Datasets: `mode: "synthetic"` ships seeded LR/HR pairs; any iterable of the reference's batch dicts works
through `main(loaders=...)`.
"""
import argparse
import logging
import os
import time
from collections import OrderedDict

from . import options as option
from . import util
from .models import create_model
from .train import create_dataset, setup_logger


def evaluate(model, loader, opt, dataset_dir, logger, scale):
    res = OrderedDict((k, []) for k in ('psnr', 'ssim', 'psnr_y', 'ssim_y', 'lpips'))
    for data in loader:
        need_HR = 'HR' in data
        model.feed_data(data, False)
        img_name = os.path.splitext(os.path.basename(data['LR_path'][0]))[0]
        model.test()
        visuals = model.get_current_visuals(need_HR=need_HR)
        sr_img = util.tensor2img(visuals['SR'])
        suffix = opt['suffix']
        util.save_img(sr_img, os.path.join(dataset_dir, img_name + (suffix or '') + '.png'))
        if not need_HR:
            logger.info(img_name)
            continue
        gt_img = util.tensor2img(visuals['HR']) / 255.
        sr_img = sr_img / 255.
        c = scale
        csr, cgt = sr_img[c:-c, c:-c, :], gt_img[c:-c, c:-c, :]
        psnr, ssim = util.calculate_psnr(csr * 255, cgt * 255), util.calculate_ssim(csr * 255, cgt * 255)
        res['psnr'].append(psnr)
        res['ssim'].append(ssim)
        lpips = float(visuals['LPIPS']) if opt['val_lpips'] else None      # test.py:88-100
        if lpips is not None:
            res['lpips'].append(lpips)
        if gt_img.shape[2] == 3:
            sr_y, gt_y = util.bgr2ycbcr(sr_img, only_y=True), util.bgr2ycbcr(gt_img, only_y=True)
            psnr_y = util.calculate_psnr(sr_y[c:-c, c:-c] * 255, gt_y[c:-c, c:-c] * 255)
            ssim_y = util.calculate_ssim(sr_y[c:-c, c:-c] * 255, gt_y[c:-c, c:-c] * 255)
            res['psnr_y'].append(psnr_y)
            res['ssim_y'].append(ssim_y)
            if lpips is not None:
                logger.info('{:20s} - PSNR: {:.6f} dB; SSIM: {:.6f}; PSNR_Y: {:.6f} dB; SSIM_Y: {:.6f}; {}: {:.3f}.'.format(img_name, psnr, ssim, psnr_y, ssim_y, model.lpips_label, lpips))
            else:
                logger.info('{:20s} - PSNR: {:.6f} dB; SSIM: {:.6f}; PSNR_Y: {:.6f} dB; SSIM_Y: {:.6f};.'.format(img_name, psnr, ssim, psnr_y, ssim_y))
        else:
            logger.info('{:20s} - PSNR: {:.6f} dB; SSIM: {:.6f}.'.format(img_name, psnr, ssim))
    return res


def main(argv=None, loaders=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-opt', type=str, required=True, help='Path to options JSON file.')
    opt = option.parse(ap.parse_args(argv).opt, is_train=False)
    for key, path in opt['path'].items():
        if key != 'pretrain_model_G' and path:
            util.mkdir(path)
    opt = option.dict_to_nonedict(opt)
    if opt['save_RealorFake']:
        # codes/SRN/test.py:44,65-66,77: the patch map comes from visuals['realorfake'], which only DePatchGAN_wavelet_model.get_current_visuals
        # (DePatchGAN_wavelet_model.py:287-309, model 'De_patch_wavelet_GAN': the SRN tree's own copy of the DSN, outside SURVEY 8) produces; with the
        # trainers this path serves ('sr', 'DASR') the reference's test.py itself stops with KeyError at test.py:66.  The DSN's patch maps are written by
        # `python -m dasr_amd.dsn_create_dataset` (fake LR + domain-distance map, codes/DSN/create_dataset_modified.py).
        raise NotImplementedError("save_RealorFake needs model 'De_patch_wavelet_GAN' (its get_current_visuals is the only one with a 'realorfake' entry); "
                                  "use dasr_amd.dsn_create_dataset for the DSN's discriminator maps")
    setup_logger('base', opt['path']['log'], 'test', screen=True)
    logger = logging.getLogger('base')
    logger.info(option.dict2str(opt))
    if loaders is None:
        loaders = []
        for phase, ds in sorted(opt['datasets'].items()):
            ds['phase'] = 'test'
            loaders.append((ds['name'], create_dataset(ds, opt)))
    model = create_model(opt)
    summary = OrderedDict()
    for name, loader in loaders:
        logger.info('\nTesting [{:s}]...'.format(name))
        t0 = time.time()
        dataset_dir = os.path.join(opt['path']['results_root'], name, 'imgs')
        util.mkdir(dataset_dir)
        res = evaluate(model, loader, opt, dataset_dir, logger, opt['scale'])
        if res['psnr']:
            ave_psnr, ave_ssim = sum(res['psnr']) / len(res['psnr']), sum(res['ssim']) / len(res['ssim'])
            summary[name] = {'psnr': ave_psnr, 'ssim': ave_ssim}
            if res['lpips']:
                summary[name]['lpips'] = sum(res['lpips']) / len(res['lpips'])
                logger.info('----Average PSNR/SSIM/{} results for {}----\n\tPSNR: {:.6f} dB; SSIM: {:.6f}; {}: {:.3f}\n'.format(
                    model.lpips_label, name, ave_psnr, ave_ssim, model.lpips_label, summary[name]['lpips']))
            else:
                logger.info('----Average PSNR/SSIM/LPIPS results for {}----\n\tPSNR: {:.6f} dB; SSIM: {:.6f}\n'.format(name, ave_psnr, ave_ssim))
            if res['psnr_y']:
                ay, asy = sum(res['psnr_y']) / len(res['psnr_y']), sum(res['ssim_y']) / len(res['ssim_y'])
                logger.info('----Y channel, average PSNR/SSIM----\n\tPSNR_Y: {:.6f} dB; SSIM_Y: {:.6f}\n'.format(ay, asy))
                summary[name].update(psnr_y=ay, ssim_y=asy)
        logger.info('[{:s}] done in {:.1f} s'.format(name, time.time() - t0))
    return summary


if __name__ == '__main__':
    main()
