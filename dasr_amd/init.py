"""Weight initialisation that reproduces the reference bit-for-bit for a given torch seed.

Reference: codes/SRN/models/networks.py:30-44,62-74 (`init_weights('kaiming', scale)`), applied by
define_G (scale 0.1, networks.py:142-143) and define_D (scale 1, networks.py:191).  The reference builds
nn.Conv2d modules first (each constructor draws its default kaiming_uniform / uniform init from the global
CPU generator) and then overwrites them with kaiming_normal_(a=0, fan_in) * scale in module order, so the
final weights are a function of (seed, construction order).  We replay exactly those draws on CPU tensors
in state_dict order (= construction order = .apply() order for RRDBNet / NLayerDiscriminator).
"""
import math
from collections import OrderedDict

import torch
from torch.nn import init


def kaiming_state_dict(spec, scale):
    """spec: [(key, shape)] in construction order; returns OrderedDict of CPU tensors."""
    sd = OrderedDict()
    # pass 1: nn.Conv2d.reset_parameters() draws (values are discarded, only the RNG stream matters)
    i = 0
    convs = []
    while i < len(spec):
        k, shape = spec[i]
        assert k.endswith('weight') and len(shape) == 4, k
        w = torch.empty(shape)
        init.kaiming_uniform_(w, a=math.sqrt(5))
        b = None
        if i + 1 < len(spec) and spec[i + 1][0] == k[:-6] + 'bias':
            fan_in = shape[1] * shape[2] * shape[3]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            b = torch.empty(spec[i + 1][1])
            init.uniform_(b, -bound, bound)
            i += 1
        convs.append((k, w, b))
        i += 1
    # pass 2: weights_init_kaiming in module order
    for k, w, b in convs:
        init.kaiming_normal_(w, a=0, mode='fan_in')
        w *= scale
        sd[k] = w
        if b is not None:
            b.zero_()
            sd[k[:-6] + 'bias'] = b
    return sd
