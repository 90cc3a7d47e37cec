"""Developer measurement: BASELINE configs[4] (G = 200, R = 760, 10^4 draws) on device,
float64 and float32, sustained."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
table = synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9)
theta = synthetic.zheng07_draws(10000, seed=1)
def dmalloc(a):
    p = ctypes.c_void_p(); _lib.check(lib.tc_device_malloc(ctypes.byref(p), a.nbytes))
    _lib.check(lib.tc_memcpy_h2d(p, a.ctypes.data_as(ctypes.c_void_p), a.nbytes)); return p
d_theta = dmalloc(theta); d_ngal = dmalloc(np.zeros(10000)); d_xi = dmalloc(np.zeros((10000, 760)))
flop = 10000 * (2.0 * 760 * 20100 + 3 * 20100)
options = [a for a in sys.argv[1:] if '=' in a]
for dtype in [a for a in sys.argv[1:] if '=' not in a] or ['float64', 'float32']:
    t0 = time.perf_counter()
    tab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'], compute_dtype=dtype)
    h = tab.to_device().handle
    for option in options:
        name, value = option.split('=')
        _lib.check(lib.tc_table_set_option(h, name.encode(), int(value)))
    print('%s: table upload %.1f s' % (dtype, time.perf_counter() - t0))
    def step():
        _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, 10000, 10, 0, d_ngal, d_xi))
    for _ in range(5): step()
    _lib.check(lib.tc_table_synchronize(h))
    t0 = time.perf_counter()
    for _ in range(40): step()
    _lib.check(lib.tc_table_synchronize(h))
    dt = (time.perf_counter() - t0) / 40
    print('%s: %.2f ms per 10^4 draws, %.3g calls/s, %.1f TFLOP/s over the whole step' % (dtype, dt * 1e3, 1e4 / dt, flop / dt / 1e12))
    ngal, xi = tab.predict_batch(theta[:8])
    print('   xi[0, :3] =', xi[0].ravel()[:3])
    del tab
