#!/usr/bin/env python3
"""What the synchronisations at the end of a timed region cost on an idle device, and right
behind a burst of 20 steps (tc_table_synchronize = hipStreamSynchronize of every lane;
tc_device_synchronize = hipDeviceSynchronize).  gpurun -- python3 tools/archive/r03_sync_cost.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402
lib = _lib.load()
n = 10000
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
handle = halotab.to_device().handle
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * 19)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
def step(k):
    s = k % 4
    _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, n, 10, 0,
        ctypes.c_void_p(d_ngal.value + s * n * 8), ctypes.c_void_p(d_xi.value + s * n * 19 * 8)))
for k in range(400): step(k)
lib.tc_table_synchronize(handle); lib.tc_device_synchronize()
def timed(f, reps=2000):
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e6
print('idle device: tc_table_synchronize %.2f us, tc_device_synchronize %.2f us' % (
    timed(lambda: lib.tc_table_synchronize(handle)), timed(lambda: lib.tc_device_synchronize())))
rows = []
for repeat in range(200):
    lib.tc_device_synchronize()
    t0 = time.perf_counter()
    for k in range(20): step(k)
    t1 = time.perf_counter()
    lib.tc_table_synchronize(handle)
    t2 = time.perf_counter()
    lib.tc_device_synchronize()
    t3 = time.perf_counter()
    rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
rows = np.median(np.array(rows), axis=0)
print('burst of 20 steps: enqueue %.1f us, table synchronize %.1f us, device synchronize %.1f us, total %.1f us = %.2f us per step' % (rows[0], rows[1], rows[2], rows[3], rows[3] / 20))
