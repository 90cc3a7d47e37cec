#!/usr/bin/env python3
"""One launch against three kernels for tables with more bins than BASELINE configs[1]'s 100
(the densities of 64 draws then take more than half of a CU's LDS: one workgroup per CU):
10^4 draws, four lanes, 19 r values.  gpurun -- python3 tools/archive/r03_fused_bins.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
n = 10000
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
cases = [(n_prim, 19, 10) for n_prim in (40, 50, 52, 56, 64, 80, 100)]
if len(sys.argv) > 1 and sys.argv[1] == 'shapes':
    # (fewer r values, another n_gauss_prim)
    cases = [(50, 3, 10), (50, 8, 10), (50, 12, 10), (50, 19, 7), (50, 19, 20), (25, 19, 10)]
for n_prim, n_r, n_gauss in cases:
    table = synthetic.synthetic_table(n_prim, 1, (n_r, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    handle = halotab.to_device().handle
    d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * n_r)):
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
    row = []
    for fused in (0, 1):
        _lib.check(lib.tc_table_set_option(handle, b'fused', fused))

        def step(k):
            s = k % 4
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n, n_gauss, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
                ctypes.c_void_p(d_xi.value + s * n * n_r * 8)))
        for k in range(500):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        t0 = time.perf_counter()
        for k in range(2000):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        row.append((time.perf_counter() - t0) / 2000 * 1e6)
    launch = [ctypes.c_int() for _ in range(4)]
    lib.tc_table_last_launch(handle, *[ctypes.byref(v) for v in launch])
    print('G = %3d, R = %2d, n_gauss %2d: three kernels %7.2f us, one launch %7.2f us per step '
          '(LDS %d bytes, %s)' % (2 * n_prim, n_r, n_gauss, row[0], row[1], launch[3].value,
           'one launch taken' if launch[2].value == 0 else 'NOT taken'))
    for ptr in (d_theta, d_ngal, d_xi):
        lib.tc_device_free(ptr)
