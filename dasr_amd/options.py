"""JSON option surface of the SRN trainer (drop-in for codes/SRN/options/options.py:8-121).

Same file format (JSON with `//` comments), same derived keys (`is_train`, dataset phase/scale/data_type,
path.{experiments_root,models,training_state,log,val_images}, network_G.scale, debug-mode periods) and the
same NoneDict semantics (missing key -> None).  Differences, all deliberate:
  * `gpu_ids` selects devices through HIP_VISIBLE_DEVICES only when DASR_SET_VISIBLE_DEVICES=1 (one process
    per GPU is launched by torch.distributed; the reference's CUDA_VISIBLE_DEVICES export is kept for parity);
  * the shipped configs' model name "DASR_FS_ESRGAN_patchGAN" is accepted as an alias of "DASR"
    (the reference raises NotImplementedError for it).
"""
import json
import logging
import os
import os.path as osp
from collections import OrderedDict


def parse(opt_path, is_train=True):
    text = ''
    with open(opt_path, 'r') as f:
        for line in f:
            text += line.split('//')[0] + '\n'
    opt = json.loads(text, object_pairs_hook=OrderedDict)
    opt['is_train'] = is_train
    scale = opt['scale']
    for phase, ds in opt['datasets'].items():
        ds['phase'] = phase.split('_')[0]
        ds['scale'] = scale
        is_lmdb = False
        for key in ('dataroot_HR', 'dataroot_HR_bg', 'dataroot_LR'):
            if ds.get(key) is not None:
                ds[key] = osp.expanduser(ds[key])
                if key != 'dataroot_HR_bg' and ds[key].endswith('lmdb'):
                    is_lmdb = True
        ds['data_type'] = 'lmdb' if is_lmdb else 'img'
        if ds['phase'] == 'train' and ds.get('subset_file') is not None:
            ds['subset_file'] = osp.expanduser(ds['subset_file'])
    for key, path in opt['path'].items():
        if path:
            opt['path'][key] = osp.expanduser(path)
    if is_train:
        root = osp.join(opt['path']['root'], 'experiments', opt['name'])
        opt['path'].update(experiments_root=root, models=osp.join(root, 'models'),
                           training_state=osp.join(root, 'training_state'), log=root,
                           val_images=osp.join(root, 'val_images'))
        if 'debug' in opt['name']:
            opt['train']['val_freq'] = 8
            opt['logger']['print_freq'] = 2
            opt['logger']['save_checkpoint_freq'] = 8
            opt['train']['lr_decay_iter'] = 10
    else:
        root = osp.join(opt['path']['root'], 'results', opt['name'])
        opt['path']['results_root'] = root
        opt['path']['log'] = root
    opt['network_G']['scale'] = scale
    # codes/SRN/options/options.py:68-71 exports CUDA_VISIBLE_DEVICES = gpu_ids for its single-process nn.DataParallel.  HIP honours
    # that variable too, so under one-process-per-GPU launch (torch.distributed.run sets WORLD_SIZE / LOCAL_RANK) it would hide every
    # device but the listed ones from EVERY rank (the shipped JSONs say gpu_ids [0]: rank 1's set_device(1) would fail).  Rule:
    # a launcher-managed process never touches *_VISIBLE_DEVICES (rank r uses device LOCAL_RANK); a stand-alone process keeps the
    # reference's behaviour.
    gpu_list = ','.join(str(x) for x in (opt.get('gpu_ids') or []))
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 or 'LOCAL_RANK' in os.environ:
        print('distributed launch: device = LOCAL_RANK, CUDA_VISIBLE_DEVICES left as is (gpu_ids %s ignored)' % gpu_list)
    else:
        os.environ['CUDA_VISIBLE_DEVICES'] = gpu_list
        print('export CUDA_VISIBLE_DEVICES=' + gpu_list)
    return opt


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def dict2str(opt, indent_l=1):
    msg = ''
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += ' ' * (indent_l * 2) + k + ':[\n' + dict2str(v, indent_l + 1) + ' ' * (indent_l * 2) + ']\n'
        else:
            msg += ' ' * (indent_l * 2) + k + ': ' + str(v) + '\n'
    return msg


def check_resume(opt):
    """options.py:107-121 (incl. its quirk: D is only re-pointed when 'gan' is in the model name)."""
    logger = logging.getLogger('base')
    if opt['path']['resume_state']:
        if opt['path']['pretrain_model_G'] or opt['path']['pretrain_model_D']:
            logger.warning('pretrain_model path will be ignored when resuming training.')
        idx = osp.basename(opt['path']['resume_state']).split('.')[0]
        opt['path']['pretrain_model_G'] = osp.join(opt['path']['models'], '{}_G.pth'.format(idx))
        logger.info('Set [pretrain_model_G] to ' + opt['path']['pretrain_model_G'])
        if 'gan' in opt['model']:
            opt['path']['pretrain_model_D'] = osp.join(opt['path']['models'], '{}_D.pth'.format(idx))
            logger.info('Set [pretrain_model_D] to ' + opt['path']['pretrain_model_D'])
