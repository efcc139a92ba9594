"""CPU: domain-distance-map restatement (oracle/dsn_dataset.py) against the fixture produced by the reference's receptive_cal.py
(SURVEY.md 8(f2)), and its closed form used on the GPU (count-normalised 17x17 box average)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.parametrize('name,fs,hw', [('gau_23x31', 'gau', (23, 31)), ('wav_40x36', 'wavelet', (40, 36)), ('avg_12x9', 'avg_pool', (12, 9))])
def test_ddm_matches_reference(name, fs, hw, golden_dir):
    from oracle import dsn_dataset as dd
    gold = np.load(os.path.join(golden_dir, 'dsn_ddm.npz'))
    d_out = gold[name + '_dout']
    got = dd.domain_distance_map(d_out, (1, 3) + hw, fs)
    np.testing.assert_allclose(got, gold[name + '_ddm'], rtol=1e-12, atol=0)
    h, w = d_out.shape[2:]
    np.testing.assert_allclose(np.array([dd.receptive(h, dd.CONVNETS['FSD']), dd.receptive(w, dd.CONVNETS['FSD'])], dtype=np.float64),
                               gold[name + '_layers'])
    # closed form: AvgPool2d(17, 1, 8, count_include_pad=False) of the discriminator map
    box = F.avg_pool2d(torch.from_numpy(d_out), 17, 1, 8, count_include_pad=False).numpy()
    np.testing.assert_allclose(box, gold[name + '_ddm'], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize('name,fs,hw,arch', [('nld_s1_gau_26x22', 'gau', (26, 22), 'nld_s1'), ('nld_s2_avg_52x44', 'avg_pool', (52, 44), 'nld_s2'),
                                             ('nld_s2_wav_54x68', 'wavelet', (54, 68), 'nld_s2'), ('nld_s1_wav_31x24', 'wavelet', (31, 24), 'nld_s1')])
def test_ddm_other_conv_tables_match_reference(name, fs, hw, arch, golden_dir):
    """the nld_s1 / nld_s2 conv tables (create_dataset_modified.py:112-121): the discriminator map is smaller than the image; the restated spread
    and the product's receptive-field walk (dasr_amd.dsn_model.receptive_walk, what parameterises dasr_ddm_spread) against the reference's
    receptive_cal.py"""
    from oracle import dsn_dataset as dd
    from dasr_amd.dsn_model import receptive_walk, DDM_CONVNETS
    gold = np.load(os.path.join(golden_dir, 'dsn_ddm.npz'))
    d_out = gold[name + '_dout']
    got = dd.domain_distance_map(d_out, (1, 3) + hw, fs, arch)
    assert got.shape == gold[name + '_ddm'].shape and not np.isnan(gold[name + '_ddm']).any()
    np.testing.assert_allclose(got, gold[name + '_ddm'], rtol=1e-12, atol=0)
    h, w = gold[name + '_ddm'].shape[2:]
    for n_px, row in ((h, 0), (w, 1)):
        np.testing.assert_allclose(np.array(dd.receptive(n_px, dd.CONVNETS[arch]), dtype=np.float64), gold[name + '_layers'][row])
        np.testing.assert_allclose(np.array(receptive_walk(n_px, DDM_CONVNETS[arch.lower()]), dtype=np.float64), gold[name + '_layers'][row])


def test_cli_flags_and_unsupported_choices():
    from dasr_amd import dsn_create_dataset as cd
    o = cd.build_parser().parse_args([])
    assert (o.generator, o.discriminator, o.kernel_size, o.filter, o.name, o.upscale_factor) == ('DeResnet', 'FSD', 5, 'gau', '0603_DSN_LRs', 4)
    for bad in (['--generator', 'SRGAN', '--checkpoint', 'x'], ['--discriminator', 'nld_s3', '--checkpoint', 'x'], ['--norm_layer', 'Group', '--checkpoint', 'x'],
                ['--no_highpass', '--checkpoint', 'x']):
        with pytest.raises(NotImplementedError):
            cd.main(bad)
