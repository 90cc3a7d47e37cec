#!/usr/bin/env python3
"""Bursts of N device-resident steps between two synchronisations (the driver's bench command:
--steps 20): us per step for the workgroup shapes of the one-launch form and mixtures of them
over the burst (head / tail launches of 32-draw workgroups).  Usage: r04_burst.py [N]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
handle = halotab.to_device().handle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 10000
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * 19)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))


def option(name, value):
    _lib.check(lib.tc_table_set_option(handle, name, value))


def step(k):
    s = k % 4
    _lib.check(lib.tc_predict_zheng07_batch_device(
        handle, d_theta, 5, n, 10, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
        ctypes.c_void_p(d_xi.value + s * n * 19 * 8)))


def bursts(shape_of, repeats=60):
    """shape_of(k): draws per workgroup (0 default, 32, 64) of step k of the burst."""
    for k in range(2000):          # settle
        step(k)
    _lib.check(lib.tc_table_synchronize(handle))
    times = []
    for burst in range(repeats):
        current = None
        t0 = time.perf_counter()
        for k in range(N):
            want = shape_of(k)
            if want != current:
                option(b'fused_draws', want)
                current = want
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        times.append((time.perf_counter() - t0) / N * 1e6)
        time.sleep(0.0005)
    option(b'fused_draws', 0)
    return np.median(times), min(times)


print('bursts of %d steps, us per step: median (min)' % N)
for name, shape_of in (
        ('all 64 draws', lambda k: 64),
        ('all 32 draws', lambda k: 32),
        ('head 2 x 32', lambda k: 32 if k < 2 else 64),
        ('head 4 x 32', lambda k: 32 if k < 4 else 64),
        ('tail 2 x 32', lambda k: 32 if k >= N - 2 else 64),
        ('tail 4 x 32', lambda k: 32 if k >= N - 4 else 64),
        ('head 2 + tail 3 x 32', lambda k: 32 if (k < 2 or k >= N - 3) else 64),
        ('head 4 + tail 4 x 32', lambda k: 32 if (k < 4 or k >= N - 4) else 64)):
    print('%-24s %6.2f (%6.2f)' % ((name, ) + bursts(shape_of)), flush=True)
option(b'fused', 0)
print('%-24s %6.2f (%6.2f)' % (('three kernels', ) + bursts(lambda k: 0)), flush=True)
