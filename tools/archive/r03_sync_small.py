#!/usr/bin/env python3
"""Synchronous host calls (which run alone on their lane): three kernels against the one-launch
kernel forced (fused = 2) for small tables.  gpurun -- python3 tools/archive/r03_sync_small.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
def timeit(call, seconds=0.3):
    for _ in range(20): call()
    t0 = time.perf_counter(); call(); per = max(time.perf_counter() - t0, 1e-6)
    n = max(20, int(seconds / per)); t0 = time.perf_counter()
    for _ in range(n): call()
    return (time.perf_counter() - t0) / n * 1e6
for n_prim in (20, 30, 50):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    handle = halotab.to_device().handle
    for n in (200, 1000, 4000, 10000):
        theta = synthetic.zheng07_draws(n, seed=1)
        row = []
        for fused, draws in ((1, 0), (2, 64), (2, 32)):
            _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
            _lib.check(lib.tc_table_set_option(handle, b'fused_draws', draws))
            _lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', 1 if fused == 2 else 0))
            row.append(timeit(lambda: halotab.predict_batch(theta)))
        print('G = %3d, %5d draws, synchronous host call: three kernels %7.1f us, one launch %7.1f us '
              '(64 draws per workgroup), %7.1f us (32)' % (2 * n_prim, n, row[0], row[1], row[2]))
