"""Developer measurement: step time of consecutive timed regions of K steps (each bracketed
by a device sync), to separate pipeline fill / drain from clock ramp-up after idling."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
dev = halotab.to_device()
n_draws = 10000
theta = synthetic.zheng07_draws(n_draws, seed=1)
d_theta = ctypes.c_void_p(); d_out = ctypes.c_void_p()
_lib.check(lib.tc_device_malloc(ctypes.byref(d_theta), theta.nbytes))
_lib.check(lib.tc_device_malloc(ctypes.byref(d_out), n_draws * 20 * 8))
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))


def region(k):
    t0 = time.perf_counter()
    for _ in range(k):
        _lib.check(lib.tc_predict_zheng07_batch_device(
            dev.handle, d_theta, 5, n_draws, 10, 0, d_out, ctypes.c_void_p(d_out.value + n_draws * 8)))
    _lib.check(lib.tc_table_synchronize(dev.handle))
    return (time.perf_counter() - t0) / k * 1e6


region(50)
for k in (50, 200, 200, 200, 1000, 200, 5000, 200, 200):
    print('%5d steps: %.2f us per step' % (k, region(k)))
time.sleep(0.5)
print('after 0.5 s idle: %5d steps: %.2f us per step' % (200, region(200)))
print('                  %5d steps: %.2f us per step' % (200, region(200)))
