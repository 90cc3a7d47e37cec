#!/usr/bin/env python3
"""mean_delta_sigma at tabulation scale: clustered halos and particles in a 250 Mpc/h box, 13
annuli to 30 Mpc/h (the shape of the reference's ds tables), per-object mass in cylinders on the
GPU against the NumPy oracle on a subsample."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import corrfunc   # noqa: E402
from oracle import paircount_oracle as oracle   # noqa: E402

rng = np.random.default_rng(9)
box = 250.0
rp_bins = np.logspace(-1, np.log10(30.0), 14)
for n_gal, n_ptcl in ((20000, 500000), (100000, 2000000)):
    centres = rng.uniform(0, box, (n_gal // 20, 3))
    galaxies = np.mod(centres[rng.integers(0, len(centres), n_gal)] +
                      rng.normal(0, 2.0, (n_gal, 3)), box)
    particles = np.mod(np.vstack([
        centres[rng.integers(0, len(centres), n_ptcl // 2)] + rng.normal(0, 4.0, (n_ptcl // 2, 3)),
        rng.uniform(0, box, (n_ptcl - n_ptcl // 2, 3))]), box)
    corrfunc.mass_in_cylinders(galaxies[:100], particles[:1000], 1.0, rp_bins, box)
    t0 = time.perf_counter()
    mass = corrfunc.mass_in_cylinders(galaxies, particles, 1.0, rp_bins, box)
    dt = time.perf_counter() - t0
    pairs = float(mass[:, -1].sum())
    print('%7d objects x %8d particles: %8.1f ms, %.3g pairs inside the largest cylinder, '
          '%.3g pairs/s' % (n_gal, n_ptcl, dt * 1e3, pairs, pairs / dt))
    t0 = time.perf_counter()
    ds = corrfunc.mean_delta_sigma(galaxies, particles, 1.0, rp_bins, period=box)
    print('        mean_delta_sigma (incl. the per-object post-processing on the host): %.1f ms'
          % ((time.perf_counter() - t0) * 1e3))
sub = galaxies[:300]
t0 = time.perf_counter()
expect = oracle.mass_in_cylinders(sub, particles, 1.0, rp_bins, box)
dt = time.perf_counter() - t0
got = corrfunc.mass_in_cylinders(sub, particles, 1.0, rp_bins, box)
print('oracle on %d objects: %.1f s (%.3g pairs/s, 1 core); GPU exact: %s' %
      (len(sub), dt, expect[:, -1].sum() / dt, np.array_equal(got, expect)))
