#!/usr/bin/env python3
"""Round 6: synchronous predict_batch calls of an ensemble sampler's size (64 ... 4096 walkers in
pageable NumPy arrays) -- the default dispatch against the one-launch forms forced (32 / 40 / 64
draws per workgroup) and the three kernels, on the reference's example-table shape (G = 60) and
the benchmark's table (G = 100).  gpurun -- python3 tools/r06_walkers.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
for n_prim in (30, 50):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    h = halotab.to_device().handle

    def option(name, value):
        _lib.check(lib.tc_table_set_option(h, name.encode(), value))
    forms = {'default': [('fused', 1), ('fused_min_draws', 0), ('fused_draws', 0)],
             'three kernels': [('fused', 0)],
             '32 draws': [('fused', 2), ('fused_min_draws', 1), ('fused_draws', 32)],
             '40 draws': [('fused', 2), ('fused_min_draws', 1), ('fused_draws', 40)],
             '64 draws': [('fused', 2), ('fused_min_draws', 1), ('fused_draws', 64)]}
    sizes = (65, 128, 256, 512, 1024, 2048, 4096)
    print('G = %d, us per synchronous predict_batch(theta) call' % (2 * n_prim))
    print('%-16s' % 'walkers' + ''.join('%8d' % n for n in sizes))
    for name, options in forms.items():
        for key, value in options:
            option(key, value)
        row = []
        for n in sizes:
            theta = synthetic.zheng07_draws(n, seed=3)
            for _ in range(30):
                halotab.predict_batch(theta)
            t0 = time.perf_counter()
            for _ in range(200):
                halotab.predict_batch(theta)
            row.append((time.perf_counter() - t0) / 200 * 1e6)
        print('%-16s' % name + ''.join('%8.1f' % us for us in row), flush=True)
