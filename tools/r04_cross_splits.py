#!/usr/bin/env python3
"""predict_cross_small_kernel: us per call against the number of workgroups a launch is given
at least (option "cross_target": several workgroups per tile of 64 draws below that), the
reference's AbacusSummit table, device-resident pipelined calls."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import Device, sustained          # noqa: E402
from tabcorr_amd import Interpolator, synthetic, _lib          # noqa: E402

lib = _lib.load()
dev = Device(lib, _lib)
interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
rng = np.random.default_rng(0)
theta = synthetic.zheng07_draws(40000, seed=1)
theta[:, 0] = rng.uniform(12.5, 13.3, len(theta))
theta[:, 3] = rng.uniform(13.6, 14.4, len(theta))
d_theta = dev.upload(theta)
d_ngal, d_xi = dev.malloc(len(theta)), dev.malloc(13 * len(theta))
h = interp.tabcorr_list[0].to_device().handle
_lib.check(lib.tc_table_set_option(h, b'fused', 2))
_lib.check(lib.tc_table_set_option(h, b'fused_min_draws', 1))
# (a second of full load first: started from idle with a few workgroups per call, the chip
# stays at a low clock for the whole run -- every number 1.5x higher)
sustained(lambda: _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, 32768, 10, 0,
                                                                 d_ngal, d_xi)),
          lambda: _lib.check(lib.tc_table_synchronize(h)), seconds=1.0)
targets = (1, 160, 256, 384, 512, 768, 1024)
print('draws   ' + ''.join('%8d' % t for t in targets))
for n in (256, 1024, 2048, 4096, 6144, 8192, 10000, 16384, 32768):
    row = []
    for target in targets:
        _lib.check(lib.tc_table_set_option(h, b'cross_target', target))
        row.append(sustained(
            lambda: _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, n, 10, 0,
                                                                   d_ngal, d_xi)),
            lambda: _lib.check(lib.tc_table_synchronize(h)), seconds=0.2) * 1e6)
    print('%6d  ' % n + ''.join('%8.1f' % v for v in row), flush=True)
