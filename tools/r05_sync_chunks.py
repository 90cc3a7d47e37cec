#!/usr/bin/env python3
"""A synchronous host-array call of 10^4 draws of BASELINE configs[1] (pageable NumPy arrays in,
pageable arrays out) through tc_predict_zheng07_batch: the serial path of rounds 1-4 and 1 .. 8
chunks with the kernels storing the results themselves or copy commands (options "sync_chunks",
"sync_direct_out"), whether the bits depend on the number of chunks, and the oracle.
gpurun -- python3 tools/r05_sync_chunks.py"""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402
from oracle import tabcorr_oracle as oracle        # noqa: E402

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
h = halotab.to_device().handle
theta = synthetic.zheng07_draws(10000, seed=1)
ngal, xi = np.empty(10000), np.empty((10000, 19))


def call():
    _lib.check(lib.tc_predict_zheng07_batch(h, _lib.as_double_p(theta), 5, 10000, 10, 0,
                                            _lib.as_double_p(ngal), _lib.as_double_p(xi)))


def timeit(n=300):
    for _ in range(30):
        call()
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    return (time.perf_counter() - t0) / n * 1e6


def option(name, value):
    _lib.check(lib.tc_table_set_option(h, name.encode(), value))


option('sync_chunks', -1)
print('serial path (sync_chunks = -1): %.1f us per call' % timeit(), flush=True)
reference = None
for chunks in (1, 2, 3, 4, 8, 0):
    for direct in (1, 2, 0):
        option('sync_chunks', chunks)
        option('sync_direct_out', direct)
        us = timeit()
        call()
        if reference is None:
            reference = (ngal.copy(), xi.copy())
        launch = [ctypes.c_int() for _ in range(4)]
        lib.tc_table_last_launch(h, *[ctypes.byref(v) for v in launch])
        print('sync_chunks %d sync_direct_out %d: %6.1f us per call   last chunk: %d workgroups of '
              '%d waves   same bits as one chunk: %s' % (
                  chunks, direct, us, launch[0].value, launch[1].value,
                  np.array_equal(ngal, reference[0]) and np.array_equal(xi, reference[1])),
              flush=True)
expect = oracle.predict_zheng07_batch(table, theta[:4])
print('max rel. difference to the oracle (4 draws): %.2e' % np.max(np.abs(xi[:4] / expect[1] - 1)))
