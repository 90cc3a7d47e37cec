#!/usr/bin/env python3
"""A step of 10^4 draws as TWO launches -- most draws in 64-draw workgroups, the rest in
32-draw workgroups that start last and live half as long -- against one launch: us per step in
bursts of N steps between two synchronisations (the driver's --steps 20) and sustained.
Usage: r04_split.py [N]"""
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


def part(k, first, count, shape):
    s = k % 4
    option(b'fused_draws', shape)
    _lib.check(lib.tc_predict_zheng07_batch_device(
        handle, ctypes.c_void_p(d_theta.value + first * 5 * 8), 5, count, 10, 0,
        ctypes.c_void_p(d_ngal.value + (s * n + first) * 8),
        ctypes.c_void_p(d_xi.value + (s * n + first) * 19 * 8)))


def step(k, tail):
    if tail == 0:
        part(k, 0, n, 64)
    else:
        part(k, 0, n - tail, 64)
        part(k, n - tail, tail, 32)


def measure(tail, repeats=60):
    option(b'fused', 2)
    option(b'fused_min_draws', 1)
    for k in range(2000):
        step(k, tail)
    _lib.check(lib.tc_table_synchronize(handle))
    t0 = time.perf_counter()
    for k in range(3000):
        step(k, tail)
    _lib.check(lib.tc_table_synchronize(handle))
    sustained = (time.perf_counter() - t0) / 3000 * 1e6
    times = []
    for burst in range(repeats):
        t0 = time.perf_counter()
        for k in range(N):
            step(k, tail)
        _lib.check(lib.tc_table_synchronize(handle))
        times.append((time.perf_counter() - t0) / N * 1e6)
        time.sleep(0.0005)
    return sustained, np.median(times), min(times)


print('us per step of 10^4 draws: sustained, bursts of %d steps median (min)' % N)
for tail in (0, 512, 1024, 2048, 3072):
    print('last %4d draws in 32-draw workgroups of their own launch: %6.2f  %6.2f (%6.2f)'
          % ((tail, ) + measure(tail)), flush=True)
option(b'fused_draws', 0)
