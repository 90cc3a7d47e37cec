#!/usr/bin/env python3
"""Stress of the round-3 host paths on one table: random interleavings of synchronous,
un-batched / many-walker, asynchronous (prediction and likelihood, random waits) and
device-pointer calls with random batch sizes, every result checked against a reference
computed once through the synchronous path.  Catches workspace / ticket / epoch hazards.

    gpurun -- python3 tools/r03_stress.py [seconds]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabcorr_amd import TabCorr, synthetic, _lib, pinned_array, pinned_empty   # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(123)
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
device = halotab.to_device()
lib, handle = device.lib, device.handle
pool = synthetic.zheng07_draws(20000, seed=4)
ref_ngal, ref_xi = halotab.predict_batch(pool)
data = ref_xi[0] * 1.05
precision = np.diag(1.0 / (0.1 * ref_xi[0])**2)
delta = ref_xi - data
ref_chi2 = np.einsum('bi,ij,bj->b', delta, precision, delta)
pinned_pool = pinned_array(pool)


def close(a, b, what):
    if not np.allclose(a, b, rtol=1e-11, atol=0):
        raise SystemExit('MISMATCH in %s: %g' % (what, np.max(np.abs(np.asarray(a) / b - 1))))


pending = []
counts = {}
start = time.time()
while time.time() - start < seconds:
    kind = rng.choice(['sync', 'many', 'one', 'async', 'async_chi2', 'wait', 'device', 'chi2'])
    counts[kind] = counts.get(kind, 0) + 1
    n = int(rng.choice([1, 3, 17, 64, 65, 200, 1000, 2500, 7167, 7168, 10000, 16001]))
    lo = int(rng.integers(0, len(pool) - n))
    if kind == 'sync':
        ngal, xi = halotab.predict_batch(pool[lo:lo + n])
        close(xi, ref_xi[lo:lo + n], 'sync xi')
        close(ngal, ref_ngal[lo:lo + n], 'sync ngal')
    elif kind == 'chi2':
        ngal, chi2 = halotab.chi2_batch(pool[lo:lo + n], data, precision)
        close(chi2, ref_chi2[lo:lo + n], 'sync chi2')
    elif kind == 'many':
        n = min(n, 64)
        ngal, xi = halotab.predict_batch(pool[lo:lo + n])
        close(xi, ref_xi[lo:lo + n], 'many xi')
    elif kind == 'one':
        ngal, xi = halotab.predict_batch(pool[lo:lo + 1])
        close(xi, ref_xi[lo:lo + 1], 'one xi')
    elif kind == 'async' and len(pending) < 40:
        out = (pinned_empty(n), pinned_empty((n, 19)))
        out[1][:] = np.nan
        pending.append(('xi', lo, n, halotab.predict_batch_async(pinned_pool[lo:lo + n], out=out)))
    elif kind == 'async_chi2' and len(pending) < 40:
        out = (pinned_empty(n), pinned_empty(n))
        pending.append(('chi2', lo, n, halotab.chi2_batch_async(pinned_pool[lo:lo + n], data,
                                                                precision, out=out)))
    elif kind == 'wait' and pending:
        what, lo, n, item = pending.pop(int(rng.integers(0, len(pending))))
        first, second = item.wait()
        close(first, ref_ngal[lo:lo + n], 'async ngal')
        close(second, (ref_xi if what == 'xi' else ref_chi2)[lo:lo + n], 'async ' + what)
    elif kind == 'device':
        d = [ctypes.c_void_p() for _ in range(3)]
        for ptr, count in zip(d, (n * 5, n, n * 19)):
            _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        chunk = np.ascontiguousarray(pool[lo:lo + n])
        _lib.check(lib.tc_memcpy_h2d(d[0], chunk.ctypes.data_as(ctypes.c_void_p), chunk.nbytes))
        for _ in range(int(rng.integers(1, 6))):
            _lib.check(lib.tc_predict_zheng07_batch_device(handle, d[0], 5, n, 10, 0, d[1], d[2]))
        _lib.check(lib.tc_table_synchronize(handle))
        xi = np.empty((n, 19))
        _lib.check(lib.tc_memcpy_d2h(xi.ctypes.data_as(ctypes.c_void_p), d[2], xi.nbytes))
        close(xi, ref_xi[lo:lo + n], 'device xi')
        for ptr in d:
            lib.tc_device_free(ptr)
for what, lo, n, item in pending:
    first, second = item.wait()
    close(second, (ref_xi if what == 'xi' else ref_chi2)[lo:lo + n], 'final async ' + what)
print('stress ok:', counts)
