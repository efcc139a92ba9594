"""Import the *reference* SRN code (read-only, /root/reference) in the build container.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Used only by oracle/gen_golden.py to
produce the committed fixtures in tests/golden/ and by the optional
tests/test_oracle_vs_reference.py (skipped when /root/reference is absent, e.g. on the
GPU box).  Nothing from the reference is copied: the modules are imported where they lie.

Third-party packages the reference imports but this image lacks are replaced by
``sys.modules`` stand-ins:
  cv2, skimage, IPython           -> empty modules (never called on the hot path)
  torchvision.models.vgg19        -> cfg-'E' VGG19 built by oracle.nets (random init;
                                     pretrained weights need a download)
  torchvision.models.alexnet      -> the AlexNet feature stack restated in oracle.lpips (seeded;
                                     LPIPS backbone, same reason)
  torchvision.utils.make_grid     -> unused placeholder
  pytorch_wavelets.DWTForward     -> oracle.nets.HaarDWT wrapped in the (LL, [Hc5d]) API
                                     (PARITY UNPINNED: sub-band order/sign convention)
"""
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = '/root/reference/codes'


def available():
    return os.path.isdir(os.path.join(REF_ROOT, 'SRN'))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs(vgg_seed=77):
    from . import nets

    class _VGG(nn.Module):
        def __init__(self):
            super().__init__()
            self.features = nets.vgg19_features()
            nets.vgg_init_(self.features, vgg_seed)

    class _Alex(nn.Module):   # LPIPS backbone (pretrained_networks.py:60): architecture restated in oracle.lpips, seeded weights
        def __init__(self):
            super().__init__()
            from . import lpips
            self.features = lpips.alexnet_init_(lpips.alexnet_features(), vgg_seed)

    class DWTForward(nn.Module):
        def __init__(self, J=1, mode='reflect', wave='haar'):
            super().__init__()
            assert J == 1 and wave == 'haar'
            self.h = nets.HaarDWT()

        def forward(self, x):
            ll, hc = self.h(x)
            n, c3, h, w = hc.shape
            c = c3 // 3
            return ll, [torch.stack((hc[:, :c], hc[:, c:2 * c], hc[:, 2 * c:]), 2)]

    class DWTInverse(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    _mod('cv2')
    tv = _mod('torchvision')
    tv.utils = _mod('torchvision.utils', make_grid=lambda *a, **k: None)
    tv.models = _mod('torchvision.models', vgg19=lambda pretrained=False: _VGG(),
                     vgg19_bn=lambda pretrained=False: None, alexnet=lambda pretrained=False: _Alex())
    tv.transforms = _mod('torchvision.transforms')
    _mod('pytorch_wavelets', DWTForward=DWTForward, DWTInverse=DWTInverse)
    sk = _mod('skimage')
    sk.measure = _mod('skimage.measure', compare_ssim=None)
    sk.color = _mod('skimage.color')
    sk.transform = _mod('skimage.transform')
    _mod('IPython', embed=lambda *a, **k: None)
    _mod('tensorboardX')
    _mod('lmdb')


def import_srn(vgg_seed=77):
    """Returns (options module, SRModel, DASR_Model, arch module, networks module)."""
    if not available():
        raise RuntimeError('reference tree not present at ' + REF_ROOT)
    install_stubs(vgg_seed)
    for p in (os.path.join(REF_ROOT, 'SRN'), REF_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import options.options as option
    from models.SR_model import SRModel
    from models.DASR_model import DASR_Model
    import models.modules.architecture as arch
    import models.networks as networks
    return option, SRModel, DASR_Model, arch, networks
