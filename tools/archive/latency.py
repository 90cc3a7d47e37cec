"""Developer measurement: where the time of one un-batched predict() goes."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Zheng07Model, synthetic, _lib

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
model = Zheng07Model()
halotab.predict(model)
dev = halotab.to_device()
lib = dev.lib
theta = _lib.contiguous(synthetic.zheng07_draws(1, seed=1))
ngal = np.empty(1); xi = np.empty(19)
N = 3000

def timeit(fn):
    for _ in range(50): fn()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    return (time.perf_counter() - t0) / N * 1e6

print('predict(model)                  %.1f us' % timeit(lambda: halotab.predict(model)))
print('predict(model, no consistency)  %.1f us' % timeit(lambda: halotab.predict(model, check_consistency=False)))
print('predict_batch(theta[1])         %.1f us' % timeit(lambda: halotab.predict_batch(theta)))
tp, np_, xp = _lib.as_double_p(theta), _lib.as_double_p(ngal), _lib.as_double_p(xi)
print('C call tc_predict_zheng07_batch %.1f us' % timeit(lambda: lib.tc_predict_zheng07_batch(dev.handle, tp, 5, 1, 10, 0, np_, xp)))
d = ctypes.c_void_p(); lib.tc_device_malloc(ctypes.byref(d), 8 * 64)
lib.tc_memcpy_h2d(d, theta.ctypes.data_as(ctypes.c_void_p), 40)
do = ctypes.c_void_p(d.value + 64); dx = ctypes.c_void_p(d.value + 128)
def dev_call():
    lib.tc_predict_zheng07_batch_device(dev.handle, d, 5, 1, 10, 0, do, dx)
    lib.tc_table_synchronize(dev.handle)
print('device call + sync              %.1f us' % timeit(dev_call))
def dev_nosync():
    lib.tc_predict_zheng07_batch_device(dev.handle, d, 5, 1, 10, 0, do, dx)
t = timeit(dev_nosync); lib.tc_table_synchronize(dev.handle)
print('device call, no sync (issue)    %.1f us' % t)
