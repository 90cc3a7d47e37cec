#!/usr/bin/env python3
"""float32 quadratic-form kernel: share of the float32 matrix peak for tables of G = 200 bins
and a growing number of r values (matrix 1.3 MB per 16 r values: one L2 holds 4 MB), 10^4
draws, kernels serialised.  gpurun -- python3 tools/archive/r03_f32_scan.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench   # noqa: E402
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
n = 10000
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
for n_r in (16, 32, 48, 64, 128, 256, 760):
    table = synthetic.synthetic_table(100, 1, (n_r, ), 'auto', seed=9)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'], compute_dtype='float32')
    handle = halotab.to_device().handle
    d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    for ptr, count in ((d_theta, n * 5), (d_ngal, n), (d_xi, n * n_r)):
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
    _lib.check(lib.tc_table_set_option(handle, b'pipeline', 0))

    def launch():
        _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, n, 10, 0, d_ngal, d_xi))

    def synchronize():
        _lib.check(lib.tc_table_synchronize(handle))
    ms, count, _ = bench.kernel_time(lib, _lib, handle, launch, synchronize, n_launches=200,
                                     max_seconds=0.5)
    flop = n * bench.pair_flops(200, n_r)
    print('R = %4d: matrix %6.1f MB, kernel %8.1f us, %.3f of the float32 peak' %
          (n_r, 200 * 201 / 2 * n_r * 4 / 1e6, ms * 1e3, flop / (ms * 1e-3) / 1e12 / 157.3))
    for ptr in (d_theta, d_ngal, d_xi):
        lib.tc_device_free(ptr)
