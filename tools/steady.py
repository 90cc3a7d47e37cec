"""Developer tool: steady-state rate of the contraction kernel alone (large batch, many
scheduling rounds; needs the knob-enabled build and TC_SKIP_OCC=1 TC_SKIP_FINALIZE=1)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
dev = halotab.to_device()
for n_draws in (10000, 100000, 200000):
    theta = synthetic.zheng07_draws(n_draws, seed=1)
    d_theta = ctypes.c_void_p(); d_out = ctypes.c_void_p()
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_theta), theta.nbytes))
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_out), n_draws * 20 * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
    def step():
        _lib.check(lib.tc_predict_zheng07_batch_device(
            dev.handle, d_theta, 5, n_draws, 10, 0, d_out, ctypes.c_void_p(d_out.value + n_draws * 8)))
    for _ in range(12):
        step()
    _lib.check(lib.tc_table_synchronize(dev.handle))
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _lib.check(lib.tc_table_synchronize(dev.handle))
    dt = (time.perf_counter() - t0) / steps
    print('%7d draws: %8.1f us per launch = %.1f us per 10^4 draws = %.1f TFLOP/s' % (
        n_draws, dt * 1e6, dt * 1e6 * 1e4 / n_draws, n_draws * 2.0705e5 / dt / 1e12))
    lib.tc_device_free(d_theta); lib.tc_device_free(d_out)
