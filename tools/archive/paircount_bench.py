"""Developer measurement: the pair counter at tabulation scale.

Clustered points in a 250 Mpc/h box, 19 r_p bins up to 30, pi_max = 40: the single-pair
entry point (what one Corrfunc.theory.DDrppi call of tabcorr/corrfunc.py:62-84 does), the
all-bin-pairs entry point (the whole loop of tabcorr/tabcorr.py:846-922 in one pass), and
the brute-force NumPy oracle on a subsample for scale."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import corrfunc
from oracle import paircount_oracle as oracle

rng = np.random.default_rng(3)
box = 250.0
rp_bins = np.logspace(-1, np.log10(30.0), 20)
for n in (100000, 400000, 1000000):
    centres = rng.uniform(0, box, (n // 60, 3))
    pos = np.mod(centres[rng.integers(0, len(centres), n)] + rng.normal(0, 3.0, (n, 3)), box)
    label = rng.integers(0, 100, n)
    corrfunc.pair_count_rppi(pos[:1000], rp_bins, 40.0, None, box)
    t0 = time.perf_counter()
    counts = corrfunc.pair_count_rppi(pos, rp_bins, 40.0, None, box)
    dt = time.perf_counter() - t0
    pairs = int(counts.sum())
    print('%8d points: auto count   %8.1f ms  %.3g pairs counted  %.3g pairs/s' % (n, dt * 1e3, pairs, pairs / dt))
    order = np.argsort(label, kind='stable')
    bins = np.split(pos[order], np.cumsum(np.bincount(label, minlength=100))[:-1])
    t0 = time.perf_counter()
    matrix = corrfunc.pair_count_matrix(bins, rp_bins, 40.0, box)
    dt = time.perf_counter() - t0
    assert int(matrix.sum()) == pairs
    print('%8d points: 100 x 100 bin pairs in one pass %8.1f ms  %.3g pairs/s' % (n, dt * 1e3, pairs / dt))
if '--no-oracle' in sys.argv:
    sys.exit(0)
sub = pos[:20000]
t0 = time.perf_counter()
expect = oracle.pair_count_rppi(sub, None, box, rp_bins, 40.0)
dt = time.perf_counter() - t0
got = corrfunc.pair_count_rppi(sub, rp_bins, 40.0, None, box)
print('oracle (NumPy brute force, 1 core) on %d points: %.1f s, %.3g pairs/s; GPU bit-exact: %s' % (
    len(sub), dt, expect.sum() / dt, np.array_equal(got, expect)))
