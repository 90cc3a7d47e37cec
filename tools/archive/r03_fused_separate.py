#!/usr/bin/env python3
"""One launch against three kernels with and without separate_gal_type (10^4 draws, four lanes):
the reference's example table size (G = 60) and BASELINE configs[1]'s (G = 100).
gpurun -- python3 tools/archive/r03_fused_separate.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
n = 10000
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
for n_prim in (30, 50):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    handle = halotab.to_device().handle
    d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    for ptr, count in ((d_theta, n * 5), (d_ngal, 8 * n), (d_xi, 4 * n * 19 * 3)):
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
    for flags in (0, 1):
        row = []
        for fused in (0, 1):
            _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
            def step(k):
                s = k % 4
                _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, n, 10, flags,
                    ctypes.c_void_p(d_ngal.value + s * n * 16), ctypes.c_void_p(d_xi.value + s * n * 19 * 3 * 8)))
            for k in range(500): step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            t0 = time.perf_counter()
            for k in range(2000): step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            row.append((time.perf_counter() - t0) / 2000 * 1e6)
        print('G = %3d, %s: three kernels %7.2f us, one launch %7.2f us per step' % (2 * n_prim, 'separated' if flags else 'total    ', row[0], row[1]))
