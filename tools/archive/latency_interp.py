"""Developer measurement: un-batched Interpolator call."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Interpolator, synthetic

def make(table):
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])

tables, keys, points = synthetic.synthetic_interpolator((5, 5), 50, 1, (19, ), 'auto', seed=7)
interp = Interpolator([make(x) for x in tables], {k: points[:, d] for d, k in enumerate(keys)})
theta = synthetic.zheng07_draws(1, seed=5)
x = np.array([[0.5 * (xp[0] + xp[-1]) for xp in interp.xp]])
for n in (1, 64, 1000):
    th = np.repeat(theta, n, axis=0); xx = np.repeat(x, n, axis=0)
    for _ in range(20):
        interp.predict_batch(th, xx)
    t0 = time.perf_counter()
    for _ in range(300):
        interp.predict_batch(th, xx)
    print('%5d draws: %.1f us per call' % (n, (time.perf_counter() - t0) / 300 * 1e6))
