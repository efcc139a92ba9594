import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


# Order of the GPU suite (the driver runs `pytest -m gpu -x`): the comparisons against the oracle / the reference fixtures first -- kernels, then
# the RRDBNet / SR step, GAN step, DSN, LPIPS, inference and checkpoints, the exact-size steps -- then the rest; HIP-vs-HIP self comparisons
# (bit-identity of schedules, determinism, soak / fuzz / lifetime runs) LAST, so that a failure of a self comparison cannot hide a parity test
# behind `-x` (one such assertion stopped the run in front of 116 parity tests).
_FILE_ORDER = ['test_gpu_kernels', 'test_gpu_sr', 'test_gpu_gan', 'test_gpu_dsn', 'test_gpu_lpips', 'test_gpu_infer', 'test_gpu_fullsize_steps',
               'test_gpu_fullsize_gan', 'test_gpu_fullsize', 'test_gpu_wgan', 'test_gpu_dsn_val', 'test_gpu_data', 'test_gpu_trajectory', 'test_gpu_dp',
               'test_gpu_launch', 'test_gpu_fuzz_shapes', 'test_gpu_lifetime']
_SELF_COMPARISON = ('bit_identical', 'deterministic', 'interference', 'soak')


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        is_gpu = it.get_closest_marker('gpu') is not None
        last = any(w in it.name for w in _SELF_COMPARISON)
        rank = _FILE_ORDER.index(mod) if mod in _FILE_ORDER else len(_FILE_ORDER)
        return (1 if is_gpu else 0, 1 if (is_gpu and last) else 0, rank if is_gpu else 0)
    items.sort(key=key)   # (stable: the order inside a file is kept)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


@pytest.fixture(scope='session')
def margins():
    """margins('text'): record the worst observed error of a tolerance check (printed, and appended to gpurun_out/parity_margins.log
    on the GPU box) so that the distance to every tolerance with an absolute / fractional escape hatch stays visible"""
    path = os.path.join(ROOT, 'gpurun_out', 'parity_margins.log')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        fh = open(path, 'a')
    except OSError:
        fh = None

    def rec(msg):
        print('[margin] ' + msg)
        if fh:
            fh.write(msg + '\n')
            fh.flush()
    yield rec
    if fh:
        fh.close()
