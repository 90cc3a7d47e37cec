"""Developer measurement: where a short timed region (the driver's --steps 20) stalls on the
host: per-step enqueue times after settle + drain + warm-up + drain."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
handle = halotab.to_device().handle
n = 10000
theta = synthetic.zheng07_draws(n, seed=1)
def dmalloc(count):
    p = ctypes.c_void_p(); _lib.check(lib.tc_device_malloc(ctypes.byref(p), count * 8)); return p
d_theta = dmalloc(theta.size)
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
d_out = dmalloc(64 * n * 20)
def step(i):
    slot = i % 64
    o = ctypes.c_void_p(d_out.value + slot * n * 20 * 8)
    x = ctypes.c_void_p(o.value + n * 8)
    _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, n, 10, 0, o, x))
def drain():
    _lib.check(lib.tc_table_synchronize(handle)); _lib.check(lib.tc_device_synchronize())
settle = int(sys.argv[1]) if len(sys.argv) > 1 else 4100
for i in range(settle): step(i)
drain()
for i in range(5): step(i)
drain()
stamps = [time.perf_counter()]
for i in range(20):
    step(i)
    stamps.append(time.perf_counter())
drain()
stamps.append(time.perf_counter())
d = np.diff(stamps) * 1e6
print('settle %5d: total %.1f us/step | per-step enqueue us: %s | drain %.0f' % (
    settle, (stamps[-1] - stamps[0]) / 20 * 1e6, ' '.join('%.0f' % v for v in d[:-1]), d[-1]))
