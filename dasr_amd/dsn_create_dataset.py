"""`python -m dasr_amd.dsn_create_dataset --checkpoint X.tar [flags]` -- apply a trained DSN to build the SRN training set
(reference: codes/DSN/create_dataset_modified.py:14-176, receptive_cal.py:34-60).

For every target-domain HR image: fake LR = G(img) saved as PNG under <out>/imgs_from_target, the domain-distance map of the fake LR
(float64 .npy, shape [1,1,h,w]; h/2 x w/2 for the wavelet filter) under <out>/ddm_target; with --including_source_ddm the map of
every source-domain LR image under <out>/ddm_source.  Same flags as the reference; its `../paths.yml` lookup is replaced by
--target_dir / --source_dir (or --dataset synthetic for seeded random images).  Generator / discriminator run on the HIP kernels
(dasr_amd.dsn_model.DSNModel.translate / .ddm_of); image IO is PIL on the host.
"""
import argparse
import os
import shutil

import numpy as np
import torch

from .dsn_model import DSNModel

IMG_EXT = ('.png', '.jpg', '.jpeg', '.JPG', '.JPEG', '.PNG')


def build_parser():
    p = argparse.ArgumentParser(description='Apply the trained model to create a dataset')
    p.add_argument('--checkpoint', default=None, type=str)
    p.add_argument('--generator', default='DeResnet', type=str)
    p.add_argument('--num_res_blocks', default=8, type=int)
    p.add_argument('--discriminator', default='FSD', type=str)
    p.add_argument('--kernel_size', default=5, type=int)
    p.add_argument('--wgan', dest='wgan', action='store_true')
    p.add_argument('--no_highpass', dest='highpass', action='store_false')
    p.add_argument('--filter', default='gau', type=str)
    p.add_argument('--cat_or_sum', default='cat', type=str)
    p.add_argument('--norm_layer', default='Instance', type=str)
    p.add_argument('--artifacts', default='tdsr', type=str)
    p.add_argument('--name', default='0603_DSN_LRs', type=str)
    p.add_argument('--dataset', default='synthetic', type=str)
    p.add_argument('--including_source_ddm', dest='including_source_ddm', action='store_true')
    p.add_argument('--upscale_factor', default=4, type=int, choices=[4])
    # additions of this build
    p.add_argument('--target_dir', default=None, type=str, help='folder of target-domain HR images (replaces ../paths.yml)')
    p.add_argument('--source_dir', default=None, type=str, help='folder of source-domain LR images (for --including_source_ddm)')
    p.add_argument('--paths', default='../paths.yml', type=str, help="yaml file with the dataset folders (the reference reads '../paths.yml', create_dataset_modified.py:49-50)")
    p.add_argument('--out_root', default='DSN_results', type=str)
    p.add_argument('--n_synthetic', default=4, type=int)
    return p


def _load(path):
    from PIL import Image
    a = np.asarray(Image.open(path).convert('RGB'), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).unsqueeze(0).contiguous()


def _save_png(t, path):
    from PIL import Image
    a = (t.clamp(0, 1) * 255.0).round().byte().permute(1, 2, 0).cpu().numpy()  # TF.to_pil_image: mul(255).byte() after clamp
    Image.fromarray(a).save(path)


# --dataset name -> (first key, second key) of paths.yml (codes/DSN/create_dataset_modified.py:52-81)
DATASETS = {'aim2019': ('aim2019', 'tdsr'), 'ntire2020': ('ntire2020', 'tdsr'), 'realsr_tddiv2k': ('realsr', 'tddiv2k'), 'realsr_tdrealsr': ('realsr', 'tdrealsr'),
            'realsr_tdrealsr_2x': ('realsr', 'tdrealsr_x2'), 'camerasr': ('camerasr', 'tdsr')}


def _named_dir(o, which):
    """folder of `which` ('source' / 'target') images of --dataset through the paths file, as the reference resolves it"""
    if o.dataset not in DATASETS:
        return None
    from .dsn_data import load_paths
    a, b = DATASETS[o.dataset]
    try:
        return load_paths(o.paths)[a][b][which]
    except (KeyError, TypeError):
        raise KeyError("%s has no entry ['%s']['%s']['%s'] (codes/paths.yml layout)" % (o.paths, a, b, which))


def _images(o, which):
    d = (o.target_dir if which == 'target' else o.source_dir) or _named_dir(o, which)
    if d:
        for f in sorted(os.listdir(d)):
            if f.endswith(IMG_EXT):
                yield f, _load(os.path.join(d, f))
    elif o.dataset == 'synthetic':
        g = torch.Generator().manual_seed(7 if which == 'target' else 8)
        for i in range(o.n_synthetic):
            hw = (160, 192) if which == 'target' else (40, 48)
            yield '%s_%03d.png' % (which, i), torch.rand(1, 3, *hw, generator=g)
    else:
        raise NotImplementedError('dataset [%s]: known names are %s (folders from --paths) and synthetic; or pass --target_dir / --source_dir' % (o.dataset, ' / '.join(sorted(DATASETS))))


def main(argv=None):
    o = build_parser().parse_args(argv)
    if o.generator not in ('DeResnet', 'DSGAN'):
        raise NotImplementedError('Generator model [{:s}] not recognized'.format(o.generator))
    if o.discriminator not in ('FSD', 'nld_s1', 'nld_s2'):
        raise NotImplementedError('Please specified conv_net of discriminator.')
    if not o.highpass or o.cat_or_sum not in ('cat', 'sum') or o.norm_layer not in ('Instance', 'Batch'):
        raise NotImplementedError('DSN on MI355X covers: high-pass front end, wavelet bands cat / sum, Instance or Batch norm')
    if o.checkpoint is None:
        print('Use --checkpoint to define the model parameters used')
        return None
    out = os.path.join(o.out_root, o.name)
    dirs = {k: os.path.join(out, k) for k in ('imgs_from_target', 'ddm_target', 'ddm_source')}
    for d in dirs.values():
        os.makedirs(d, exist_ok=True)
    m = DSNModel(dict(n_res_blocks=o.num_res_blocks, kernel_size=o.kernel_size, filter=o.filter, norm_layer=o.norm_layer, w_per=0.0,
                      discriminator=o.discriminator, generator=o.generator, cat_or_sum=o.cat_or_sum, wgan=o.wgan))   # (--wgan: raw logit maps, model.py:104-105)
    m.load(o.checkpoint)
    print('Using model at epoch %d' % m.epoch)
    shutil.copyfile(o.checkpoint, os.path.join(out, o.name + '.tar'))
    dev = m.device
    n = 0
    for name, img in _images(o, 'target'):
        H, W = img.shape[-2] // 4 * 4, img.shape[-1] // 4 * 4
        fake, _, ddm = m.translate(img[..., :H, :W].to(dev))
        _save_png(fake[0], os.path.join(dirs['imgs_from_target'], name.rsplit('.', 1)[0] + '.png'))
        np.save(os.path.join(dirs['ddm_target'], name.split('.')[0]), ddm.double().cpu().numpy())
        n += 1
    if o.including_source_ddm:
        for name, img in _images(o, 'source'):
            _, ddm = m.ddm_of(img.to(dev))
            np.save(os.path.join(dirs['ddm_source'], name.split('.')[0]), ddm.double().cpu().numpy())
    return out, n


if __name__ == '__main__':
    main()
