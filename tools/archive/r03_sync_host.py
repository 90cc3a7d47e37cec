#!/usr/bin/env python3
"""Synchronous host-to-host calls (pageable arrays in and out) for several batch sizes and the
many-r table of BASELINE configs[4] (61 MB of results per 10^4 draws): us per call.
TABCORR_AMD_LIBRARY selects the build.  gpurun -- python3 tools/archive/r03_sync_host.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic   # noqa: E402


def timeit(call, seconds=0.4):
    call()
    call()
    t0 = time.perf_counter()
    call()
    per = max(time.perf_counter() - t0, 1e-6)
    n = max(5, int(seconds / per))
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    return (time.perf_counter() - t0) / n * 1e6


table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
for n in (3000, 10000, 40000, 100000):
    theta = synthetic.zheng07_draws(n, seed=1)
    us = timeit(lambda: halotab.predict_batch(theta))
    print('19 bins, %6d draws: %8.1f us per call = %.3g calls/s' % (n, us, n / us * 1e6))
table = synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9)
for dtype in ('float32', ):
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'], compute_dtype=dtype)
    theta = synthetic.zheng07_draws(10000, seed=1)
    us = timeit(lambda: halotab.predict_batch(theta), seconds=1.0)
    print('760 bins (%s), 10000 draws: %8.1f us per call = %.3g calls/s' % (dtype, us, 1e10 / us))
