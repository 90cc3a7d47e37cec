#!/usr/bin/env python3
"""Tables of more than 104 bins: the one-launch kernel with workgroups of eight waves x 32 draws
against the three kernels over batch sizes (four lanes, sustained).
gpurun -- python3 tools/archive/r03_fused_wide.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402
lib = _lib.load()
for n_prim, n_sec, flags in ((56, 1, 0), (50, 2, 0), (50, 2, 5)):
    n_theta = 7 if flags & 4 else 5
    table = synthetic.synthetic_table(n_prim, n_sec, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    handle = halotab.to_device().handle
    for n in (1024, 2048, 4096, 6144, 8192, 10000, 20000):
        theta = synthetic.zheng07_draws(n, seed=1)
        if flags & 4:
            theta = np.hstack([theta, np.random.default_rng(2).uniform(-1, 1, (n, 2))])
        theta = np.ascontiguousarray(theta)
        d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        for ptr, count in ((d_theta, n * n_theta), (d_ngal, 8 * n), (d_xi, 4 * n * 19 * 3)):
            _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
        def step(k):
            s = k % 4
            _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, n_theta, n, 10, flags,
                ctypes.c_void_p(d_ngal.value + s * n * 16), ctypes.c_void_p(d_xi.value + s * n * 19 * 3 * 8)))
        row = []
        for fused in (0, 1):
            _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
            _lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', 1))
            for k in range(200): step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            t0 = time.perf_counter()
            for k in range(1000): step(k)
            _lib.check(lib.tc_table_synchronize(handle))
            row.append((time.perf_counter() - t0) / 1000 * 1e6)
        print('G = %3d, flags %d, %5d draws: three kernels %7.2f us, one launch %7.2f us per step' % (
            2 * n_prim * n_sec, flags, n, row[0], row[1]), flush=True)
        for ptr in (d_theta, d_ngal, d_xi):
            lib.tc_device_free(ptr)
