#!/usr/bin/env python3
"""A burst of 20 device-resident steps between two synchronisations (the driver's bench
command), repeated: wall time per step, and -- under rocprofv3 --kernel-trace -- the kernel
timeline of one burst (tools/archive/r03_burst.sh)."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
handle = halotab.to_device().handle
for option in sys.argv[1:]:
    name, value = option.split('=')
    _lib.check(lib.tc_table_set_option(handle, name.encode(), int(value)))
n = 10000
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * 19)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))


def step(k):
    s = k % 4
    _lib.check(lib.tc_predict_zheng07_batch_device(
        handle, d_theta, 5, n, 10, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
        ctypes.c_void_p(d_xi.value + s * n * 19 * 8)))


for k in range(4000):          # settle
    step(k)
_lib.check(lib.tc_table_synchronize(handle))
times = []
for burst in range(50):
    t0 = time.perf_counter()
    for k in range(20):
        step(k)
    _lib.check(lib.tc_table_synchronize(handle))
    times.append((time.perf_counter() - t0) / 20 * 1e6)
    time.sleep(0.0005)
print('20-step bursts: median %.2f us per step (min %.2f, max %.2f)' %
      (np.median(times), min(times), max(times)))
