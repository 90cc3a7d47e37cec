import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, ctypes
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19,), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
h = halotab.to_device().handle
theta = synthetic.zheng07_draws(10000, seed=1)
ngal = np.empty(10000); xi = np.empty((10000, 19))
def call():
    _lib.check(lib.tc_predict_zheng07_batch(h, _lib.as_double_p(theta), 5, 10000, 10, 0, _lib.as_double_p(ngal), _lib.as_double_p(xi)))
def timeit(n=300):
    for _ in range(30): call()
    t0 = time.perf_counter()
    for _ in range(n): call()
    return (time.perf_counter() - t0) / n * 1e6
def opt(name, v): _lib.check(lib.tc_table_set_option(h, name.encode(), v))
ref = None
for chunks in (-1, 1, 2, 4, 6, 8, 12, 16, 0):
    for direct in (2, 1, 0):
        opt('sync_chunks', chunks); opt('sync_direct_out', direct)
        us = timeit()
        call()
        if chunks == 1 and direct == 2: ref = (ngal.copy(), xi.copy())
        same = '' if ref is None else ' same bits as 1 chunk: %s %s' % (np.array_equal(ngal, ref[0]), np.array_equal(xi, ref[1]))
        launch = [ctypes.c_int() for _ in range(4)]
        lib.tc_table_last_launch(h, *[ctypes.byref(v) for v in launch])
        print('chunks %3d direct_out %d: %6.1f us  (wg %d waves %d slabs %d)%s' % (chunks, direct, us, launch[0].value, launch[1].value, launch[2].value, same), flush=True)
        if chunks == -1: break
from oracle import tabcorr_oracle as oracle
e = oracle.predict_zheng07_batch(table, theta[:3])
print('vs oracle', np.max(np.abs(xi[:3] / e[1] - 1)))
