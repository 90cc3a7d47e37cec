#!/usr/bin/env python3
"""The resident ensemble kernel (option "resident", 2 .. 256 walkers per synchronous call):
agreement with the batched path, independence of a walker's result from the size of the
ensemble, us per call (C entry point and predict_batch) with and without it, and the phase
stamps of workgroup 0 (10 ns ticks from the sight of the call)."""
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
ONLY_CHECK = '--check' in sys.argv
ONLY_TIME = '--time' in sys.argv


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


def table_g100():
    t = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    return TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs'])


for name, halotab in (
        ('G=100', table_g100()),
        ('G=60', TabCorr.read(os.path.join(REPO, 'tests', 'golden', 'bolplanck_wp.hdf5')))):
    h = halotab.to_device().handle
    _lib.check(lib.tc_table_set_option(h, b'resident_min_walkers', 2))
    n_r = halotab.predict_batch(synthetic.zheng07_draws(2, seed=1))[1].shape[1]

    def call(theta, resident):
        n = len(theta)
        ngal, xi = np.empty(n), np.empty((n, n_r))
        _lib.check(lib.tc_table_set_option(h, b'resident', resident))
        _lib.check(lib.tc_predict_zheng07_batch(
            h, _lib.as_double_p(theta), 5, n, 10, 0, _lib.as_double_p(ngal),
            _lib.as_double_p(xi)))
        return ngal, xi

    big = np.ascontiguousarray(synthetic.zheng07_draws(256, seed=7))
    ref_ngal, ref_xi = call(big, 0)
    worst = 0.0
    first = {}
    for n in () if ONLY_TIME else (2, 3, 63, 64, 65, 100, 128, 129, 192, 193, 255, 256):
        ngal, xi = call(np.ascontiguousarray(big[:n]), 1)
        err = max(np.max(np.abs(xi / ref_xi[:n] - 1)), np.max(np.abs(ngal / ref_ngal[:n] - 1)))
        worst = max(worst, err)
        same = all(np.array_equal(xi[i], first.setdefault(i, xi[i])) for i in range(n))
        again = call(np.ascontiguousarray(big[:n]), 1)
        print('%-6s %4d walkers: max rel. difference to the batched path %.2e, results '
              'independent of the ensemble size %s, identical on repetition %s'
              % (name, n, err, same, np.array_equal(again[1], xi)), flush=True)
    assert worst < 1e-12
    if ONLY_TIME and name != 'G=60':
        continue
    if ONLY_CHECK:
        _lib.check(lib.tc_table_set_option(h, b'resident', 0))
        continue
    for aperture, n in [(a, n) for a in ((1, ) if ONLY_TIME else (0, 1))
                        for n in ((2, 8, 16, 32, 48, 64, 128, 256) if ONLY_TIME else (2, 16, 64, 128, 192, 256))]:
        _lib.check(lib.tc_table_set_option(h, b'resident_aperture', aperture))
        print('mailbox in %s' % ('device memory' if aperture else 'page-locked memory'))
        theta = np.ascontiguousarray(big[:n])
        ngal, xi = np.empty(n), np.empty((n, n_r))
        row = []
        for resident in (0, 1):
            _lib.check(lib.tc_table_set_option(h, b'resident', resident))
            row.append(time_calls(lambda: _lib.check(lib.tc_predict_zheng07_batch(
                h, _lib.as_double_p(theta), 5, n, 10, 0, _lib.as_double_p(ngal),
                _lib.as_double_p(xi)))))
            row.append(time_calls(lambda: halotab.predict_batch(theta)))
        stamps = (ctypes.c_uint64 * 11)()
        rows = []
        host = []
        for _ in range(300):
            _lib.check(lib.tc_predict_zheng07_batch(
                h, _lib.as_double_p(theta), 5, n, 10, 0, _lib.as_double_p(ngal),
                _lib.as_double_p(xi)))
            _lib.check(lib.tc_debug_ensemble_stamps(h, stamps))
            rows.append([(stamps[i] - stamps[0]) / 100.0 for i in range(1, 8)])
            host.append([stamps[8 + i] / 1000.0 for i in range(3)])
        phases = np.median(np.array(rows), axis=0)
        print('    host, us from the begin of the call (medians): published %.2f, every row '
              'combined %.2f, of which spent on rows that were there %.2f' % tuple(np.median(np.array(host), axis=0)))
        print('%-6s %4d walkers: C call %6.1f -> %6.1f us, predict_batch %6.1f -> %6.1f us; '
              'workgroup 0, us after the sight of the call (medians): occupation stored %.1f, '
              'group seen %.1f, densities in LDS %.1f, quarters summed %.1f, partial sums '
              'stored %.1f, finished %.1f (developer build; zeros otherwise)'
              % (name, n, row[0], row[2], row[1], row[3], *phases[:5], phases[6]), flush=True)
    _lib.check(lib.tc_table_set_option(h, b'resident', 0))
