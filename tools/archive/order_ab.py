"""Developer measurement: schedule orders of the quadratic-form kernel for a table with
several r tiles (G = 100, R = 160: 8 r tiles, 8.3 MB of matrix), 10^4 draws, device."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
n_r = int(sys.argv[1]) if len(sys.argv) > 1 else 160
table = synthetic.synthetic_table(50, 1, (n_r, ), 'auto', seed=9)
theta = synthetic.zheng07_draws(10000, seed=1)
def dmalloc(a):
    p = ctypes.c_void_p(); _lib.check(lib.tc_device_malloc(ctypes.byref(p), a.nbytes))
    _lib.check(lib.tc_memcpy_h2d(p, a.ctypes.data_as(ctypes.c_void_p), a.nbytes)); return p
d_theta = dmalloc(theta); d_ngal = dmalloc(np.zeros(10000)); d_xi = dmalloc(np.zeros((10000, n_r)))
flop = 10000 * (2.0 * n_r * 5050 + 3 * 5050)
tab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
h = tab.to_device().handle
ref = None
for order in (0, 2, 0, 2):
    _lib.check(lib.tc_table_set_option(h, b'quad_order', order))
    def step():
        _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, 10000, 10, 0, d_ngal, d_xi))
    for _ in range(50): step()
    _lib.check(lib.tc_table_synchronize(h))
    t0 = time.perf_counter()
    for _ in range(300): step()
    _lib.check(lib.tc_table_synchronize(h))
    dt = (time.perf_counter() - t0) / 300
    xi = np.empty((10000, n_r)); _lib.check(lib.tc_memcpy_d2h(xi.ctypes.data_as(ctypes.c_void_p), d_xi, xi.nbytes))
    if ref is None: ref = xi
    print('R = %d order %d: %.1f us per 10^4 draws, %.1f TFLOP/s over the whole step, max rel diff vs order 0 %.1e' % (
        n_r, order, dt * 1e6, flop / dt / 1e12, np.max(np.abs(xi / ref - 1))))
