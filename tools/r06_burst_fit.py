import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
h = halotab.to_device().handle
n = 10000
theta = np.ascontiguousarray(synthetic.zheng07_draws(n, seed=1))
d = [ctypes.c_void_p() for _ in range(3)]
for ptr, count in zip(d, (n * 5, 4 * n, 4 * n * 19)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
_lib.check(lib.tc_memcpy_h2d(d[0], theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
def step(k):
    s = k % 4
    _lib.check(lib.tc_predict_zheng07_batch_device(h, d[0], 5, n, 10, 0, ctypes.c_void_p(d[1].value + s*n*8), ctypes.c_void_p(d[2].value + s*n*19*8)))
for k in range(2000): step(k)
lib.tc_table_synchronize(h)
# idle costs
for name, f in (('tc_table_synchronize', lambda: lib.tc_table_synchronize(h)), ('tc_device_synchronize', lambda: lib.tc_device_synchronize())):
    t0 = time.perf_counter()
    for _ in range(1000): f()
    print('%s on an idle device: %.2f us' % (name, (time.perf_counter() - t0) / 1000 * 1e6))
for N in (20, 40, 80, 160, 320):
    ts = []
    for rep in range(40):
        t0 = time.perf_counter()
        for k in range(N): step(k)
        t1 = time.perf_counter()
        lib.tc_table_synchronize(h)
        t2 = time.perf_counter()
        lib.tc_device_synchronize()
        t3 = time.perf_counter()
        ts.append(((t3 - t0) * 1e6, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6))
        time.sleep(0.0005)
    m = np.median(np.array(ts), axis=0)
    print('burst of %3d: %.1f us = %.2f per step; enqueue %.1f, table sync %.1f, device sync %.1f' % (N, m[0], m[0] / N, m[1], m[2], m[3]))
