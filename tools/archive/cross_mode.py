"""Developer measurement: mode 'cross' tables (e.g. galaxy-galaxy lensing), device-resident
pipelined steps of 10^4 draws, against the mode 'auto' table of the same bins."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

n_draws = 10000
theta = _lib.contiguous(synthetic.zheng07_draws(n_draws, seed=1))
for mode in ('auto', 'cross'):
    table = synthetic.synthetic_table(50, 1, (19, ), mode, seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    dev = halotab.to_device()
    lib = dev.lib
    d = ctypes.c_void_p()
    _lib.check(lib.tc_device_malloc(ctypes.byref(d), theta.nbytes + 8 * n_draws * 20 * 8))
    _lib.check(lib.tc_memcpy_h2d(d, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))

    def out(slot):
        base = d.value + theta.nbytes + slot * n_draws * 20 * 8
        return ctypes.c_void_p(base), ctypes.c_void_p(base + n_draws * 8)

    def burst(n):
        for k in range(n):
            ngal, xi = out(k % 8)
            _lib.check(lib.tc_predict_zheng07_batch_device(dev.handle, d, 5, n_draws, 10, 0, ngal, xi))
        _lib.check(lib.tc_table_synchronize(dev.handle))

    for _ in range(10):
        burst(256)
    t0 = time.perf_counter()
    for _ in range(10):
        burst(256)
    dt = (time.perf_counter() - t0) / 2560
    _lib.check(lib.tc_table_set_option(dev.handle, b'pipeline', 0))
    burst(256)
    t0 = time.perf_counter()
    burst(1024)
    ds = (time.perf_counter() - t0) / 1024
    print('mode %-5s  %.2f us per step pipelined (%.3g calls/s), %.2f us serialised' % (
        mode, dt * 1e6, n_draws / dt, ds * 1e6))
