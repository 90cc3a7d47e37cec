#!/usr/bin/env python3
"""predict_fused_kernel with 16 waves per workgroup (one workgroup per CU, up to 160 KB of LDS)
against 8 waves (two per CU) and against the three kernels: sustained rate on four lanes and a
burst of 20 steps between two synchronisations (the driver's command), 10^4 draws.
gpurun -- python3 tools/archive/r03_fused_waves.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
n = 10000
cases = [(50, 1, 0), (56, 1, 0), (64, 1, 0), (80, 1, 0), (50, 2, 0), (50, 2, 1),
         (50, 2, 5), (57, 2, 0)]
for n_prim, n_sec, flags in cases:
    n_theta = 7 if flags & 4 else 5
    theta = synthetic.zheng07_draws(n, seed=1)
    if flags & 4:
        theta = np.hstack([theta, np.random.default_rng(2).uniform(-1, 1, (n, 2))])
    theta = np.ascontiguousarray(theta)
    table = synthetic.synthetic_table(n_prim, n_sec, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    handle = halotab.to_device().handle
    d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    for ptr, count in ((d_theta, n * n_theta), (d_ngal, 8 * n), (d_xi, 4 * n * 19 * 3)):
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))

    def step(k):
        s = k % 4
        _lib.check(lib.tc_predict_zheng07_batch_device(
            handle, d_theta, n_theta, n, 10, flags, ctypes.c_void_p(d_ngal.value + s * n * 16),
            ctypes.c_void_p(d_xi.value + s * n * 19 * 3 * 8)))
    text = []
    for label, fused, waves in (('three kernels', 0, 0), ('8 waves', 1, 8), ('16 waves', 1, 16),
                                ('default shape', 1, 0)):
        _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
        _lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', 1))
        _lib.check(lib.tc_table_set_option(handle, b'fused_waves', waves))
        for k in range(400):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        launch = [ctypes.c_int() for _ in range(4)]
        lib.tc_table_last_launch(handle, *[ctypes.byref(v) for v in launch])
        if fused and (launch[2].value != 0 or (waves and launch[1].value != waves)):
            text.append('%s: -' % label)
            continue
        t0 = time.perf_counter()
        for k in range(2000):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        steady = (time.perf_counter() - t0) / 2000 * 1e6
        bursts = []
        for repeat in range(30):
            _lib.check(lib.tc_device_synchronize())
            t0 = time.perf_counter()
            for k in range(20):
                step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            bursts.append((time.perf_counter() - t0) / 20 * 1e6)
        text.append('%s%s: %.2f (burst of 20: %.2f)' % (
            label, '' if waves else ' = %d waves, %d workgroups' % (launch[1].value, launch[0].value),
            steady, float(np.median(bursts))))
    print('G = %3d, flags %d, us per step: %s' % (2 * n_prim * n_sec, flags, ' | '.join(text)),
          flush=True)
    for ptr in (d_theta, d_ngal, d_xi):
        lib.tc_device_free(ptr)
