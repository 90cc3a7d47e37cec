#!/usr/bin/env python3
"""Device-resident pipelined step of BASELINE configs[1]'s table for several batch sizes: the
one-launch path (predict_fused_kernel) against the three-kernel path, and the asynchronous
host-to-host call.  Usage (GPU box): python tools/archive/r03_fused_scan.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
N_PRIM, N_R = int(os.environ.get('N_PRIM', '50')), int(os.environ.get('N_R', '19'))
table = synthetic.synthetic_table(N_PRIM, 1, (N_R, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
handle = halotab.to_device().handle
n_max = 100000
theta = synthetic.zheng07_draws(n_max, seed=1)
d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
for ptr, count in ((d_theta, n_max * 5), (d_ngal, 4 * n_max), (d_xi, 4 * n_max * N_R)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
_lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', int(os.environ.get('FUSED_MIN', '1'))))
_lib.check(lib.tc_table_set_option(handle, b'fused_max_draws', 10000000))
print('%8s %14s %14s' % ('draws', 'three kernels', 'one launch'))
sizes = [int(v) for v in sys.argv[1:]] or [65, 256, 512, 1024, 2048, 4096, 8192, 10000, 16384,
                                           40000, 100000]
for n in sizes:
    row = []
    for fused in (0, 1):
        _lib.check(lib.tc_table_set_option(handle, b'fused', fused))

        def step(k):
            s = k % 4
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n, 10, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
                ctypes.c_void_p(d_xi.value + s * n * N_R * 8)))
        steps = max(200, min(5000, int(2e7 / n)))
        for k in range(steps // 4):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        row.append((time.perf_counter() - t0) / steps * 1e6)
    print('%8d %11.2f us %11.2f us   (%.3g / %.3g draws/s)' %
          (n, row[0], row[1], n / row[0] * 1e6, n / row[1] * 1e6))
