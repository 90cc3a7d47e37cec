#!/usr/bin/env python3
"""Where the one-launch form (32 draws x eight waves) starts to pay against the three kernels:
small batches, four lanes.  gpurun -- python3 tools/archive/r03_fused_low.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402
lib = _lib.load()
for n_prim, n_r in ((20, 19), (30, 19), (50, 19), (50, 3), (52, 19)):
    table = synthetic.synthetic_table(n_prim, 1, (n_r, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    handle = halotab.to_device().handle
    text = []
    for n in (256, 512, 768, 1024, 1280, 1536, 2048, 3072):
        theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
        d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * n_r)):
            _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
        def step(k):
            s = k % 4
            _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, n, 10, 0,
                ctypes.c_void_p(d_ngal.value + s * n * 8), ctypes.c_void_p(d_xi.value + s * n * n_r * 8)))
        row = []
        for fused in (0, 1):
            _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
            _lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', 1))
            for k in range(300): step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            t0 = time.perf_counter()
            for k in range(2000): step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            row.append((time.perf_counter() - t0) / 2000 * 1e6)
        text.append('%d: %.1f / %.1f' % (n, row[0], row[1]))
        for ptr in (d_theta, d_ngal, d_xi):
            lib.tc_device_free(ptr)
    print('G = %3d, R = %2d, draws: three kernels / one launch (us per step): %s' % (2 * n_prim, n_r, ', '.join(text)), flush=True)
