"""Developer measurement: the driver's short timed region (20 steps between drains) repeated
many times in one process -- distribution of the per-region time."""
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
for i in range(int(os.environ.get('SETTLE', '6000'))): step(i)
drain()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
times = []
for rep in range(60):
    for i in range(5): step(i)
    drain()
    t0 = time.perf_counter()
    for i in range(K): step(i)
    t1 = time.perf_counter()
    drain()
    t2 = time.perf_counter()
    times.append(((t2 - t0) / K * 1e6, (t1 - t0) / K * 1e6, (t2 - t1) * 1e6))
t = np.array(times)
print('first regions (us/step):', ' '.join('%.1f' % v for v in t[:6, 0]))
print('%d-step regions: us/step min %.1f  p50 %.1f  p90 %.1f  max %.1f | enqueue us/step p50 %.1f | final drain us p50 %.1f max %.1f' % (
    K, t[:, 0].min(), np.median(t[:, 0]), np.percentile(t[:, 0], 90), t[:, 0].max(), np.median(t[:, 1]), np.median(t[:, 2]), t[:, 2].max()))
