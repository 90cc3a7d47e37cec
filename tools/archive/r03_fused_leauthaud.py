#!/usr/bin/env python3
"""The Leauthaud11 family: one launch against three kernels (four lanes, device-resident
draws), total and modulate_with_cenocc, the reference's example table size (G = 60) and
BASELINE configs[1]'s (G = 100), batches of 1000 ... 10^4 draws.
gpurun -- python3 tools/archive/r03_fused_leauthaud.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tabcorr_amd import TabCorr, Leauthaud11Model, synthetic, _lib
lib = _lib.load()
base = Leauthaud11Model().device_theta()
rng = np.random.default_rng(1)
for n_prim in (30, 50):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    handle = halotab.to_device().handle
    for n in (1000, 2000, 4000, 10000):
        theta = np.tile(base, (n, 1))
        theta[:, :12] *= 1.0 + 0.01 * rng.normal(size=(n, 12))
        theta = np.ascontiguousarray(theta)
        d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        for ptr, count in ((d_theta, n * 14), (d_ngal, 8 * n), (d_xi, 4 * n * 19)):
            _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
        for flags in (16, 18):
            row = []
            for fused in (0, 2):
                _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
                _lib.check(lib.tc_table_set_option(handle, b'fused_min_draws', 1))
                def step(k):
                    s = k % 4
                    _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 14, n, 10, flags,
                        ctypes.c_void_p(d_ngal.value + s * n * 16), ctypes.c_void_p(d_xi.value + s * n * 19 * 8)))
                for k in range(200): step(k)
                _lib.check(lib.tc_table_synchronize(handle))
                t0 = time.perf_counter()
                for k in range(1000): step(k)
                _lib.check(lib.tc_table_synchronize(handle))
                row.append((time.perf_counter() - t0) / 1000 * 1e6)
            print('G = %3d, %5d draws, %s: three kernels %7.2f us, one launch %7.2f us per step' % (
                2 * n_prim, n, 'modulated' if flags & 2 else 'plain    ', row[0], row[1]), flush=True)
        for ptr in (d_theta, d_ngal, d_xi):
            _lib.check(lib.tc_device_free(ptr))
