"""An ensemble MCMC on a tabulated w_p table with the likelihood fused on the device: the
use the reference describes in its README ("fast enough for an MCMC", README.md:7,72-75),
with all walkers of a step evaluated as ONE batch.

    python examples/example_mcmc.py tests/golden/bolplanck_wp.hdf5

A plain affine-invariant stretch move (Goodman & Weare 2010) in NumPy: no sampler package is
needed to see the call pattern -- `chi2_batch(theta, data, precision)` returns `(ngal, chi2)`
per walker; draws outside the prior or with a NaN prediction are rejected by their
likelihood.  Under `torchrun --nproc-per-node N` swap in
`tabcorr_amd.parallel.chi2_batch_sharded` to spread the walkers over N GPUs.
"""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tabcorr_amd import TabCorr  # noqa: E402

fname = sys.argv[1] if len(sys.argv) > 1 else 'tests/golden/bolplanck_wp.hdf5'
halotab = TabCorr.read(fname)
rng = np.random.default_rng(0)

# mock data: the prediction of a fiducial model with 5 % errors
truth = np.array([12.1, 0.3, 11.8, 13.2, 1.05])      # logMmin, sigma_logM, logM0, logM1, alpha
ngal_true, wp_true = halotab.predict_batch(truth[np.newaxis])
sigma = 0.05 * wp_true[0]
data = wp_true[0] + sigma * rng.normal(size=sigma.shape)
precision = np.diag(1.0 / sigma**2)
low = np.array([11.0, 0.05, 10.5, 12.0, 0.5])
high = np.array([13.5, 1.0, 13.0, 14.5, 1.6])


def log_probability(theta):
    """All walkers at once: one call into the library per half-step."""
    inside = np.all((theta >= low) & (theta <= high), axis=1)
    ngal, chi2 = halotab.chi2_batch(np.clip(theta, low, high), data, precision)
    chi2 = chi2 + ((ngal - ngal_true[0]) / (0.05 * ngal_true[0]))**2
    logp = np.where(inside & np.isfinite(chi2), -0.5 * chi2, -np.inf)
    return logp


n_walkers, n_steps = 2048, 200
walkers = truth + 0.02 * rng.normal(size=(n_walkers, 5))
logp = log_probability(walkers)
start = time.perf_counter()
n_accepted = 0
for step in range(n_steps):
    for half in (0, 1):                                # update one half against the other
        mine = np.arange(half, n_walkers, 2)
        others = walkers[np.arange(1 - half, n_walkers, 2)]
        z = (1 + rng.uniform(size=len(mine)))**2 / 2   # g(z) ~ 1 / sqrt(z) on [1/2, 2]
        partner = others[rng.integers(0, len(others), len(mine))]
        proposal = partner + z[:, np.newaxis] * (walkers[mine] - partner)
        logp_new = log_probability(proposal)
        accept = np.log(rng.uniform(size=len(mine))) < 4 * np.log(z) + logp_new - logp[mine]
        walkers[mine[accept]] = proposal[accept]
        logp[mine[accept]] = logp_new[accept]
        n_accepted += int(accept.sum())
elapsed = time.perf_counter() - start
print('%d walkers x %d steps = %d likelihood evaluations in %.2f s (%.3g per second), '
      'acceptance %.2f' % (n_walkers, n_steps, n_walkers * n_steps, elapsed,
                           n_walkers * n_steps / elapsed, n_accepted / (n_walkers * n_steps)))
for name, mean, std, true in zip(('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha'),
                                 walkers.mean(axis=0), walkers.std(axis=0), truth):
    print('%-10s %.3f +- %.3f   (truth %.3f)' % (name, mean, std, true))
