#!/usr/bin/env python3
"""Several walkers in one launch (tc_predict_zheng07_many): workgroups per launch.
gpurun -- python3 tools/archive/r03_many_blocks.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402
lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
handle = halotab.to_device().handle
for n in (4, 8, 16, 32, 64):
    theta = synthetic.zheng07_draws(n, seed=n)
    first = halotab.predict_batch(theta)
    row = []
    for blocks in (128, 256, 512, 128, 256):
        _lib.check(lib.tc_table_set_option(handle, b'many_blocks', blocks))
        got = halotab.predict_batch(theta)
        assert np.allclose(got[1], first[1], rtol=1e-12)
        for _ in range(300):
            halotab.predict_batch(theta)
        t0 = time.perf_counter()
        for _ in range(3000):
            halotab.predict_batch(theta)
        row.append((time.perf_counter() - t0) / 3000 * 1e6)
    print('%2d walkers: 128 workgroups %.2f / %.2f us, 256: %.2f / %.2f, 512: %.2f' % (n, row[0], row[3], row[1], row[4], row[2]), flush=True)
