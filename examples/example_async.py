"""Several independent ensembles sampled in turn through the asynchronous API: while the device
evaluates the proposals of one ensemble, the host accepts / rejects and proposes for the
others -- the pattern `tc_chi2_zheng07_batch_async` + `tc_table_wait` is built for
(`include/tabcorr_amd.h`; the reference's usage is a loop of predict() calls, README.md:72-75).

    python examples/example_async.py tests/golden/bolplanck_wp.hdf5

Every ensemble owns page-locked arrays for its proposals and its results
(`tabcorr_amd.pinned_empty`), so nothing is copied between the sampler and the library.
"""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tabcorr_amd import TabCorr, pinned_empty  # noqa: E402

fname = sys.argv[1] if len(sys.argv) > 1 else 'tests/golden/bolplanck_wp.hdf5'
halotab = TabCorr.read(fname)
rng = np.random.default_rng(0)
truth = np.array([12.1, 0.3, 11.8, 13.2, 1.05])      # logMmin, sigma_logM, logM0, logM1, alpha
ngal_true, wp_true = halotab.predict_batch(truth[np.newaxis])
sigma = 0.05 * wp_true[0]
data = wp_true[0] + sigma * rng.normal(size=sigma.shape)
precision = np.diag(1.0 / sigma**2)
low = np.array([11.0, 0.05, 10.5, 12.0, 0.5])
high = np.array([13.5, 1.0, 13.0, 14.5, 1.6])

n_ensembles, n_walkers, n_steps = 4, 4096, 100


class Ensemble:
    """A Metropolis ensemble: every walker proposes a Gaussian step, all proposals of a step
    are evaluated as one asynchronous batch."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.walkers = truth + 0.02 * self.rng.normal(size=(n_walkers, 5))
        self.logp = np.full(n_walkers, -np.inf)
        self.proposal = pinned_empty((n_walkers, 5))          # read by the device
        self.out = (pinned_empty(n_walkers), pinned_empty(n_walkers))   # ngal, chi2
        self.pending = None
        self.accepted = 0

    def propose(self):
        self.proposal[:] = self.walkers + 0.01 * self.rng.normal(size=self.walkers.shape)
        self.inside = np.all((self.proposal >= low) & (self.proposal <= high), axis=1)
        np.clip(self.proposal, low, high, out=self.proposal)
        self.pending = halotab.chi2_batch_async(self.proposal, data, precision, out=self.out)

    def update(self):
        ngal, chi2 = self.pending.wait()
        chi2 = chi2 + ((ngal - ngal_true[0]) / (0.05 * ngal_true[0]))**2
        logp_new = np.where(self.inside & np.isfinite(chi2), -0.5 * chi2, -np.inf)
        accept = np.log(self.rng.uniform(size=n_walkers)) < logp_new - self.logp
        self.walkers[accept] = self.proposal[accept]
        self.logp[accept] = logp_new[accept]
        self.accepted += int(accept.sum())


ensembles = [Ensemble(seed) for seed in range(n_ensembles)]
for ensemble in ensembles:
    ensemble.propose()
start = time.perf_counter()
for step in range(n_steps):
    for ensemble in ensembles:       # the others' batches are in flight meanwhile
        ensemble.update()
        ensemble.propose()
for ensemble in ensembles:
    ensemble.update()
elapsed = time.perf_counter() - start
total = n_ensembles * n_walkers * (n_steps + 1)
print('%d ensembles x %d walkers x %d steps = %d likelihood evaluations in %.2f s '
      '(%.3g per second), acceptance %.2f' %
      (n_ensembles, n_walkers, n_steps, total, elapsed, total / elapsed,
       sum(e.accepted for e in ensembles) / total))
samples = np.concatenate([e.walkers for e in ensembles])
for name, mean, std, true in zip(('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha'),
                                 samples.mean(axis=0), samples.std(axis=0), truth):
    print('%-10s %.3f +- %.3f   (truth %.3f)' % (name, mean, std, true))
