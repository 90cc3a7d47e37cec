"""Developer measurement: the interpolator (BASELINE configs[3]: 5 x 5 grid of cfg2 tables,
12 500 draws per GPU) through the device-pointer API, sustained."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Interpolator, synthetic, _lib

def make(table):
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])

lib = _lib.load()
tables, keys, points = synthetic.synthetic_interpolator((5, 5), 50, 1, (19, ), 'auto', seed=7)
interp = Interpolator([make(x) for x in tables], {k: points[:, d] for d, k in enumerate(keys)})
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
theta = synthetic.zheng07_draws(n, seed=5)
rng = np.random.default_rng(6)
x = np.ascontiguousarray(np.stack([rng.uniform(xp[0], xp[-1], size=n) for xp in interp.xp], axis=-1))
ngal, xi = interp.predict_batch(theta, x)          # builds the device handle
handle = interp._device.handle if hasattr(interp, '_device') else None
dev = interp.to_device() if hasattr(interp, 'to_device') else None
handle = dev.handle if dev is not None else handle

def dmalloc(a):
    p = ctypes.c_void_p()
    _lib.check(lib.tc_device_malloc(ctypes.byref(p), a.nbytes))
    _lib.check(lib.tc_memcpy_h2d(p, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
    return p

d_theta, d_x = dmalloc(theta), dmalloc(x)
d_ngal, d_xi = dmalloc(np.zeros(n)), dmalloc(np.zeros((n, 19)))
def step():
    _lib.check(lib.tc_interp_predict_zheng07_batch_device(handle, d_theta, 5, d_x, n, 10, 0, d_ngal, d_xi))
for _ in range(20):
    step()
_lib.check(lib.tc_interp_synchronize(handle))
steps = 300
t0 = time.perf_counter()
for _ in range(steps):
    step()
_lib.check(lib.tc_interp_synchronize(handle))
dt = (time.perf_counter() - t0) / steps
flops = n * 25 * 2.0705e5
print('%d draws: %.1f us per call, %.3g calls/s, %.1f TFLOP/s' % (n, dt * 1e6, n / dt, flops / dt / 1e12))
