#!/usr/bin/env python3
"""Ensemble-sized synchronous calls (theta on the host -> results on the host): us per call of
tc_predict_zheng07_batch for n walkers, default route and with the one-launch form forced
(options fused=2, fused_min_draws=1), G = 100 / R = 19 and the reference's G = 60 table."""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tabcorr_amd import TabCorr, synthetic, _lib          # noqa: E402

lib = _lib.load()
_lib.require_device()


def time_calls(call, seconds=0.25, warm=30):
    for _ in range(warm):
        call()
    t0 = time.perf_counter()
    call()
    per = max(time.perf_counter() - t0, 1e-6)
    n = max(10, int(seconds / per))
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    return (time.perf_counter() - t0) / n * 1e6


for name, halotab in (
        ('G=100', TabCorr.from_arrays(**{k: v for k, v in zip(
            ('gal_type', 'tpcf_matrix', 'tpcf_shape', 'attrs'),
            [synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)[k]
             for k in ('gal_type', 'tpcf_matrix', 'tpcf_shape', 'attrs')])})),
        ('G=60', TabCorr.read(os.path.join(REPO, 'tests', 'golden', 'bolplanck_wp.hdf5')))):
    h = halotab.to_device().handle
    for n in (1, 16, 64, 65, 128, 256, 512, 1024, 2048):
        theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=n))
        ngal, xi = np.empty(n), np.empty((n, 19))
        row = []
        for fused, min_draws in ((1, 0), (2, 1)):
            _lib.check(lib.tc_table_set_option(h, b'fused', fused))
            _lib.check(lib.tc_table_set_option(h, b'fused_min_draws', min_draws))
            row.append(time_calls(lambda: _lib.check(lib.tc_predict_zheng07_batch(
                h, _lib.as_double_p(theta), 5, n, 10, 0, _lib.as_double_p(ngal),
                _lib.as_double_p(xi)))))
        python = time_calls(lambda: halotab.predict_batch(theta))
        print('%-6s %5d walkers: C call %6.1f us, one launch forced %6.1f us, predict_batch '
              '%6.1f us' % (name, n, row[0], row[1], python), flush=True)
    _lib.check(lib.tc_table_set_option(h, b'fused', 1))
    _lib.check(lib.tc_table_set_option(h, b'fused_min_draws', 0))
