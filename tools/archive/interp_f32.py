"""Developer measurement: Interpolator over float32 tables with hundreds of r values (a 3 x 3
... 4 x 4 grid of BASELINE configs[4]-like tables), device-resident batches."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Interpolator, synthetic, _lib

n_prim, tpcf_shape, grid = 50, (19, 20), (4, 4)
tables, keys, points = synthetic.synthetic_interpolator(grid, n_prim, 1, tpcf_shape, 'auto', seed=3,
                                                        dtype=np.float32)
interp = Interpolator([TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs'],
                                           compute_dtype='float32') for t in tables],
                      {key: points[:, d] for d, key in enumerate(keys)})
n_draws = 4096
theta = synthetic.zheng07_draws(n_draws, seed=1)
rng = np.random.default_rng(2)
x = np.stack([rng.uniform(xp[0], xp[-1], size=n_draws) for xp in interp.xp], axis=-1)
for _ in range(5):
    ngal, xi = interp.predict_batch(theta, x)
t0 = time.perf_counter()
for _ in range(20):
    ngal, xi = interp.predict_batch(theta, x)
dt = (time.perf_counter() - t0) / 20
n_bins = 2 * n_prim
flop = n_draws * len(tables) * (2 * 380 * n_bins * (n_bins + 1) // 2)
print('%d tables, G = %d, R = %d, %d draws: %.2f ms per call (host arrays), %.1f TFLOP/s of contraction' % (
    len(tables), n_bins, 380, n_draws, dt * 1e3, flop / dt / 1e12))
