#!/usr/bin/env python3
"""One-launch kernel with 32 draws per workgroup (four waves, three workgroups per CU) against
64 draws (eight waves, two per CU) and the three kernels: parity, sustained rate over batch sizes
(four lanes, device-resident draws) and bursts of 20 steps.
gpurun -- python3 tools/archive/r03_fused_half.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
for n_prim in (50, 30):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    handle = halotab.to_device().handle
    _lib.check(lib.tc_table_set_option(handle, b'fused', 2))
    _lib.check(lib.tc_table_set_option(handle, b'fused_waves', 0))
    _lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', 1))
    _lib.check(lib.tc_table_set_option(handle, b'single_draw', 0))
    theta = synthetic.zheng07_draws(333, seed=2)
    results = {}
    for draws in (64, 32):
        _lib.check(lib.tc_table_set_option(handle, b'fused_draws', draws))
        results[draws] = halotab.predict_batch(theta)
    print('G = %d: 32 against 64 draws per workgroup: max rel diff ngal %.2e, xi %.2e' % (
        2 * n_prim, np.max(np.abs(results[32][0] / results[64][0] - 1)),
        np.max(np.abs(results[32][1] / results[64][1] - 1))), flush=True)
    for n in (1024, 2048, 4096, 6144, 10000, 20000):
        th = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
        d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * 19)):
            _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        _lib.check(lib.tc_memcpy_h2d(d_theta, th.ctypes.data_as(ctypes.c_void_p), th.nbytes))

        def step(k):
            s = k % 4
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n, 10, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
                ctypes.c_void_p(d_xi.value + s * n * 19 * 8)))
        text = []
        for label, fused, draws, waves in (('three kernels', 0, 64, 0), ('64 draws', 1, 64, 0),
                                           ('32 draws', 1, 32, 0), ('32 draws x 8 waves', 1, 32, 8)):
            _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
            _lib.check(lib.tc_table_set_option(handle, b'fused_draws', draws))
            _lib.check(lib.tc_table_set_option(handle, b'fused_waves', waves))
            for k in range(300):
                step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            t0 = time.perf_counter()
            for k in range(1500):
                step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            steady = (time.perf_counter() - t0) / 1500 * 1e6
            bursts = []
            for repeat in range(30):
                _lib.check(lib.tc_device_synchronize())
                t0 = time.perf_counter()
                for k in range(20):
                    step(k)
                _lib.check(lib.tc_table_synchronize(handle))
                bursts.append((time.perf_counter() - t0) / 20 * 1e6)
            text.append('%s %.2f (burst %.2f)' % (label, steady, float(np.median(bursts))))
        print('G = %3d, %5d draws, us per step: %s' % (2 * n_prim, n, ' | '.join(text)), flush=True)
        for ptr in (d_theta, d_ngal, d_xi):
            lib.tc_device_free(ptr)
    _lib.check(lib.tc_table_set_option(handle, b'fused', 2))
    _lib.check(lib.tc_table_set_option(handle, b'fused_waves', 0))
