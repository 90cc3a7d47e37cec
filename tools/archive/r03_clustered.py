#!/usr/bin/env python3
"""Device-resident step of cfg2 (10^4 draws) with draws from the uniform prior against
posterior-like draws (tight cluster): what the occupation kernel's wave-uniform shortcuts
buy.  Run once per library build (TABCORR_AMD_LIBRARY selects it)."""
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
n = 10000
rng = np.random.default_rng(5)
sets = {'uniform prior': synthetic.zheng07_draws(n, seed=1),
        'posterior-like': np.array([12.3, 0.25, 12.6, 13.6, 1.05]) + rng.normal(0, 1, (n, 5)) *
        np.array([0.03, 0.02, 0.1, 0.05, 0.03])}
d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * 19)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
import itertools
options = [o.split('=') for o in sys.argv[1:]]
for name, value in options:
    _lib.check(lib.tc_table_set_option(handle, name.encode(), int(value)))
for name, theta in sets.items():
    theta = np.ascontiguousarray(theta)
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))

    def step(k):
        s = k % 4
        _lib.check(lib.tc_predict_zheng07_batch_device(
            handle, d_theta, 5, n, 10, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
            ctypes.c_void_p(d_xi.value + s * n * 19 * 8)))
    for k in range(4000):
        step(k)
    _lib.check(lib.tc_table_synchronize(handle))
    t0 = time.perf_counter()
    for k in range(10000):
        step(k)
    _lib.check(lib.tc_table_synchronize(handle))
    print('%-16s %.2f us per step' % (name, (time.perf_counter() - t0) / 10000 * 1e6))
