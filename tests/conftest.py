import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


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
