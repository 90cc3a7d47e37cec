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


# The moment expansions of csrc/series.h are off by default (option "series"): the tests of the
# occupation paths run twice, with the node loops and with both expansions on.
SERIES_MODULES = {'test_gpu_fused', 'test_gpu_grouped', 'test_gpu_cross_fused'}


@pytest.fixture(autouse=True)
def expansions(request, monkeypatch):
    value = getattr(request, 'param', None)
    if value is None:
        return
    from tabcorr_amd import tabcorr, _lib
    original = tabcorr._DeviceTable.__init__

    def init(self, *args, **kwargs):
        original(self, *args, **kwargs)
        _lib.check(self.lib.tc_table_set_option(self.handle, b'series', value))
    monkeypatch.setattr(tabcorr._DeviceTable, '__init__', init)


def pytest_generate_tests(metafunc):
    if (metafunc.module.__name__ in SERIES_MODULES
            and 'expansions' in metafunc.fixturenames):
        metafunc.parametrize('expansions', [0, 3], indirect=True,
                             ids=['nodes', 'expansions'])
