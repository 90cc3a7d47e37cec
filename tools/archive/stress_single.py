"""Developer stress test: thousands of un-batched calls (single fused launch) against the
batched path -- would expose a missing fence in the last-workgroup reduction."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic

worst = 0.0
for n_prim, n_sec, shape, mode in [(50, 1, (19, ), 'auto'), (30, 1, (19, ), 'cross'),
                                   (17, 2, (7, ), 'auto'), (50, 2, (19, ), 'auto')]:
    table = synthetic.synthetic_table(n_prim, n_sec, shape, mode, seed=3)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    theta = synthetic.zheng07_draws(3000, seed=11)
    ngal, xi = halotab.predict_batch(theta)
    for i in range(len(theta)):
        n1, x1 = halotab.predict_batch(theta[i:i + 1])
        worst = max(worst, abs(n1[0] / ngal[i] - 1.0), np.max(np.abs(x1[0] / xi[i] - 1.0)))
    print(n_prim, n_sec, shape, mode, 'max relative difference so far %.3g' % worst)
assert worst < 1e-12
print('ok')
