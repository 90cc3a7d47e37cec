#!/usr/bin/env python3
"""Random interleaving around BOTH resident kernels (tools/archive/r03_resident_stress.py with
the ensemble kernel behind the calls of 4 and 200 walkers): 30 000 calls with pauses around the
idle time-out (20 - 400 us), single draws / ensembles / other flags / switching the option off
and on in between, three tables alternating -- every result against the one recorded before;
an ensemble result that differs is reported with whether it equals the LAUNCHED path's (the
fallback when the kernel could not serve the call).
gpurun -- timeout -k 10 300 python3 tools/r04_resident_stress.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabcorr_amd import TabCorr, synthetic   # noqa: E402

rng = np.random.default_rng(0)
tables = [synthetic.synthetic_table(n_prim, 1, (n_r, ), 'auto', seed=n_prim)
          for n_prim, n_r in ((30, 19), (50, 19), (12, 5))]
halotabs = [TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs'])
            for t in tables]
theta = synthetic.zheng07_draws(512, seed=3)
expect = []
for halotab in halotabs:
    from tabcorr_amd import _lib
    _lib.check(_lib.load().tc_table_set_option(halotab.to_device().handle,
                                               b'resident_min_walkers', 2))
    halotab.set_resident(True, idle_us=100)
    expect.append([halotab.predict_batch(theta[i:i + 1]) for i in range(512)])
batch = [halotab.predict_batch(theta[:200]) for halotab in halotabs]
launched = []
for halotab in halotabs:
    halotab.set_resident(False)
    launched.append(halotab.predict_batch(theta[:200]))
    halotab.set_resident(True, idle_us=100)
bad = []
t0 = time.time()
counts = {'resident': 0, 'batch': 0, 'walkers': 0, 'modulate': 0, 'toggle': 0, 'pause': 0}
for call in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30000):
    k = int(rng.integers(0, len(halotabs)))
    halotab = halotabs[k]
    what = rng.random()
    if what < 0.9:
        i = int(rng.integers(0, 512))
        ngal, xi = halotab.predict_batch(theta[i:i + 1])
        assert np.array_equal(ngal, expect[k][i][0]) and np.array_equal(xi, expect[k][i][1]), (call, k, i)
        counts['resident'] += 1
    elif what < 0.93:
        ngal, xi = halotab.predict_batch(theta[:200])
        if not np.array_equal(xi, batch[k][1]):
            rows, cols = np.nonzero(xi != batch[k][1])
            print('call', call, 'table', k, 'max rel', np.abs(xi / batch[k][1] - 1).max(), 'walkers', len(set(rows)), 'of 200; equal to the launched result:', np.array_equal(xi, launched[k][1]), flush=True)
            bad.append(call)
            if len(bad) > 6: break
        counts['batch'] += 1
    elif what < 0.95:
        ngal, xi = halotab.predict_batch(theta[5:9])
        assert np.allclose(xi, np.concatenate([expect[k][i][1] for i in range(5, 9)]), rtol=1e-13)
        counts['walkers'] += 1
    elif what < 0.97:
        halotab.predict_batch(theta[7:8], modulate_with_cenocc=True)
        counts['modulate'] += 1
    elif what < 0.98:
        halotab.set_resident(False)
        halotab.predict_batch(theta[1:2])
        halotab.set_resident(True, idle_us=int(rng.integers(20, 400)))
        counts['toggle'] += 1
    else:
        time.sleep(float(rng.uniform(0, 0.0004)))
        counts['pause'] += 1
    if call % 50000 == 0:
        print(call, 'calls, %.1f s' % (time.time() - t0), flush=True)
print('all results as recorded:', counts, '%.1f s' % (time.time() - t0))
