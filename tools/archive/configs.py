"""Developer measurement: the five BASELINE.json configurations through the host-buffer
API (draws and results cross PCIe; table resident)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Interpolator, synthetic


def make(table, **kw):
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'], **kw)


def timeit(fn, reps):
    for _ in range(5):       # buffers grow and pages fault in on the first calls
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


theta = synthetic.zheng07_draws(10000, seed=1)
t = make(synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0))
dt = timeit(lambda: t.predict_batch(theta), 50)
print('cfg2  G=100 R=19  10^4 draws            : %8.1f us/batch  %.3g calls/s' % (dt * 1e6, 1e4 / dt))
t3 = make(synthetic.synthetic_table(50, 2, (19, ), 'auto', seed=3))
theta7 = np.hstack([theta, np.random.default_rng(0).uniform(-1, 1, (10000, 2))])
dt = timeit(lambda: t3.predict_batch(theta7, separate_gal_type=True, assembias=True), 20)
print('cfg3  G=200 R=19  separate + assembias   : %8.1f us/batch  %.3g calls/s' % (dt * 1e6, 1e4 / dt))
tables, keys, points = synthetic.synthetic_interpolator((5, 5), 50, 1, (19, ), 'auto', seed=7)
interp = Interpolator([make(x) for x in tables], {k: points[:, d] for d, k in enumerate(keys)})
th = synthetic.zheng07_draws(12500, seed=5)
rng = np.random.default_rng(6)
x = np.stack([rng.uniform(xp[0], xp[-1], size=len(th)) for xp in interp.xp], axis=-1)
dt = timeit(lambda: interp.predict_batch(th, x), 10)
print('cfg4  5x5 interpolator, 12500 draws/GPU  : %8.1f us/batch  %.3g calls/s' % (dt * 1e6, 12500 / dt))
t5 = make(synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9), compute_dtype='float32')
dt = timeit(lambda: t5.predict_batch(theta), 5)
print('cfg5  G=200 R=760 float32 MFMA           : %8.1f us/batch  %.3g calls/s' % (dt * 1e6, 1e4 / dt))
t5d = make(synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9))
dt = timeit(lambda: t5d.predict_batch(theta), 3)
print('cfg5 f64  G=200 R=760 float64            : %8.1f us/batch  %.3g calls/s' % (dt * 1e6, 1e4 / dt))
tc = make(synthetic.synthetic_table(50, 1, (19, ), 'cross', seed=2))
dt = timeit(lambda: tc.predict_batch(theta), 50)
print('cross G=100 R=19  10^4 draws             : %8.1f us/batch  %.3g calls/s' % (dt * 1e6, 1e4 / dt))
