#!/usr/bin/env python3
"""Round 6, VERDICT r05 item 1: the latency form of the one-launch kernel (40 draws per
workgroup, one workgroup per CU) against the throughput form (64 draws) and the three kernels for
calls that have the chip to themselves -- device-pointer calls on ONE lane over batch sizes, and
the synchronous host-array call tc_predict_zheng07_batch on pageable NumPy arrays over the number
of chunks.  gpurun -- python3 tools/r06_latency.py [n_prim n_r]"""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
n_prim = int(sys.argv[1]) if len(sys.argv) > 1 else 50
n_r = int(sys.argv[2]) if len(sys.argv) > 2 else 19
table = synthetic.synthetic_table(n_prim, 1, (n_r, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
h = halotab.to_device().handle
N = 40960
theta = synthetic.zheng07_draws(N, seed=1)
pointers = [ctypes.c_void_p() for _ in range(3)]
for ptr, count in zip(pointers, (theta.size, N, N * n_r)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
d_theta, d_ngal, d_xi = pointers
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))


def option(name, value):
    _lib.check(lib.tc_table_set_option(h, name.encode(), value))


def last_launch():
    launch = [ctypes.c_int() for _ in range(4)]
    lib.tc_table_last_launch(h, *[ctypes.byref(v) for v in launch])
    return tuple(v.value for v in launch)


def device_us(n, seconds=0.15):
    def call():
        _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, n, 10, 0, d_ngal, d_xi))
    for _ in range(50):
        call()
    _lib.check(lib.tc_table_synchronize(h))
    count, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            call()
        count += 20
    _lib.check(lib.tc_table_synchronize(h))
    return (time.perf_counter() - t0) / count * 1e6


print('table: %d bins, %d r values' % (2 * n_prim, n_r))
print('-- device-pointer calls on ONE lane (each launch alone on the chip), us per call')
option('lanes', 1)
forms = {'three kernels': [('fused', 0)],
         '64 draws': [('fused', 2), ('fused_min_draws', 1), ('fused_draws', 64)],
         '32 draws': [('fused', 2), ('fused_min_draws', 1), ('fused_draws', 32)],
         '40 draws (latency form)': [('fused', 2), ('fused_min_draws', 1), ('fused_draws', 40)],
         'default': [('fused', 1), ('fused_min_draws', 0), ('fused_draws', 0)]}
sizes = (1024, 2048, 4096, 6144, 8192, 10000, 10240, 12288, 16384, 20480, 40960)
print('%-26s' % 'draws' + ''.join('%8d' % n for n in sizes))
for name, options in forms.items():
    for key, value in options:
        option(key, value)
    row = []
    for n in sizes:
        row.append(device_us(n))
    print('%-26s' % name + ''.join('%8.1f' % us for us in row) + '   ' + str(last_launch()),
          flush=True)
option('lanes', 4)
for key, value in forms['default']:
    option(key, value)
print('-- pipelined (four lanes), default options: %.1f us per 10^4 draws  %s' % (
    device_us(10000, 0.5), last_launch()))

print('-- synchronous host-array calls, pageable NumPy arrays, us per call of 10^4 draws')
ngal, xi = np.empty(10000), np.empty((10000, n_r))
th = theta[:10000].copy()


def host_us(n=300):
    def call():
        _lib.check(lib.tc_predict_zheng07_batch(h, _lib.as_double_p(th), 5, 10000, 10, 0,
                                                _lib.as_double_p(ngal), _lib.as_double_p(xi)))
    for _ in range(30):
        call()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    return (time.perf_counter() - t0) / n * 1e6


for spread in (0, 1):
    option('fused_spread', spread)
    for chunks in (-1, 1, 2, 3, 4, 0):
        option('sync_chunks', chunks)
        us = host_us()
        print('fused_spread %d sync_chunks %2d: %6.1f us   last launch %s' % (
            spread, chunks, us, last_launch()), flush=True)
