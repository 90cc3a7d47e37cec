import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run through gpurun)')


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a host without an AMD GPU driver (no /dev/kfd) skips
    the GPU tests instead of failing in hipGetDeviceCount.  On a host WITH
    the driver nothing is skipped: a missing library or an unusable device
    must fail loudly there (there is no CPU fallback to fall through to)."""
    if os.path.exists('/dev/kfd'):
        return
    skip = pytest.mark.skip(reason='no AMD GPU driver on this host (/dev/kfd)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(REPO, 'tests', 'golden')
