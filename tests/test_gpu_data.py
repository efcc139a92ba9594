"""GPU: device-side input pipeline against batches produced by the reference dataset class
(tests/golden/data_pipeline.npz, oracle/gen_golden_data.py) with the same seeds."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_batches_equal_reference_dataset(golden_dir):
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from dasr_amd.data import DeviceUnpairedDataset
    from oracle.gen_golden_data import make_images
    gold = np.load(os.path.join(golden_dir, 'data_pipeline.npz'))
    imgs = {k: [torch.from_numpy(a) for a in v] for k, v in make_images().items()}
    ds = DeviceUnpairedDataset({'batch_size': 3, 'HR_size': 32, 'use_flip': True, 'use_rot': True, 'use_shuffle': False}, scale=4, images=imgs)
    for case in (0, 1):
        s1, s2 = [int(x) for x in gold['c%d_seeds' % case]]
        random.seed(s1)
        np.random.seed(s2)
        b = ds.batch([0, 3, 4])
        for key in ('LR_fake', 'LR_real', 'HR', 'HR_unpair'):
            assert torch.equal(b[key].cpu(), torch.from_numpy(gold['c%d_%s' % (case, key)])), (case, key)   # pure data movement: bit exact
        np.testing.assert_allclose(b['fake_w'].cpu().numpy(), gold['c%d_fake_w' % case], rtol=0, atol=2e-6)   # fp32 bilinear of a float64 map
    # paired LRHR mode
    from dasr_amd.data import DevicePairedDataset
    dp = DevicePairedDataset({'batch_size': 3, 'HR_size': 32, 'use_flip': True, 'use_rot': True, 'use_shuffle': False}, scale=4,
                             images={'LR': imgs['fake_LR'], 'HR': imgs['HR']})
    random.seed(31)
    b = dp.batch([2, 0, 4])
    assert torch.equal(b['LR'].cpu(), torch.from_numpy(gold['p_LR'])) and torch.equal(b['HR'].cpu(), torch.from_numpy(gold['p_HR']))
    # LRHR WITHOUT an LR folder (LRHR_dataset.py:63-88): the LR images are MATLAB-style bicubic down-samplings of the HR images (imresize_matlab, pinned to the
    # reference's util.imresize_np by tests/golden/imresize.npz): same batches as a dataset that is handed those LR images, same random draws
    from dasr_amd.data import imresize_matlab
    lr_made = [imresize_matlab(t, 0.25) for t in imgs['HR']]
    d_fly = DevicePairedDataset({'batch_size': 3, 'HR_size': 32, 'use_flip': True, 'use_rot': True, 'use_shuffle': False}, scale=4, images={'LR': None, 'HR': imgs['HR']})
    d_ref = DevicePairedDataset({'batch_size': 3, 'HR_size': 32, 'use_flip': True, 'use_rot': True, 'use_shuffle': False}, scale=4, images={'LR': lr_made, 'HR': imgs['HR']})
    random.seed(77)
    b1 = d_fly.batch([1, 3, 0])
    random.seed(77)
    b2 = d_ref.batch([1, 3, 0])
    assert torch.equal(b1['LR'], b2['LR']) and torch.equal(b1['HR'], b2['HR']) and tuple(b1['LR'].shape) == (3, 3, 8, 8)
    with pytest.raises(NotImplementedError):   # an HR size that is not a multiple of the scale goes through cv2.resize in the reference first
        DevicePairedDataset({'batch_size': 1, 'HR_size': 32, 'use_flip': False, 'use_rot': False}, scale=4, images={'LR': None, 'HR': [torch.rand(3, 50, 64)]})
    # iterator protocol / shapes / device
    batches = list(DeviceUnpairedDataset({'batch_size': 2, 'HR_size': 32, 'use_flip': False, 'use_rot': False, 'use_shuffle': True}, scale=4, images=imgs))
    assert len(batches) == 2 and batches[0]['HR'].is_cuda and tuple(batches[0]['fake_w'].shape) == (2, 1, 8, 8)


def test_device_batches_feed_the_gan_step():
    """the dict goes straight into DASR_Model.feed_data / optimize_parameters"""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from oracle import fixtures
    from dasr_amd import options
    from dasr_amd.data import DeviceUnpairedDataset
    from dasr_amd.models import create_model
    g = torch.Generator().manual_seed(0)
    imgs = {'fake_LR': [torch.rand(3, 40, 44, generator=g) for _ in range(4)], 'real_LR': [torch.rand(3, 36, 52, generator=g) for _ in range(3)],
            'HR': [torch.rand(3, 160, 176, generator=g) for _ in range(4)], 'fake_w': [torch.rand(1, 20, 22, generator=g) for _ in range(4)]}
    ds = DeviceUnpairedDataset({'batch_size': 2, 'HR_size': 128, 'use_flip': True, 'use_rot': True, 'use_shuffle': True}, scale=4, images=imgs)
    opt = fixtures.make_opt('dasr_wavelet_nf32_nb2_n2_32')
    opt['gpu_ids'] = [0]
    m = create_model(options.dict_to_nonedict(opt))
    for step, batch in enumerate(ds, 1):
        m.update_learning_rate()
        m.feed_data(batch, True)
        m.optimize_parameters(step)
    log = m.get_current_log()
    assert all(np.isfinite(v) for v in log.values()) and len(log) >= 4
