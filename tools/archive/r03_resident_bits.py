import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from tabcorr_amd import TabCorr, synthetic
for n_prim in (30, 50):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(20, seed=8)
    h = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    plain = [h.predict_batch(theta[i:i+1]) for i in range(20)]
    h.set_resident(True)
    for i in range(20):
        ngal, xi = h.predict_batch(theta[i:i+1])
        d = np.abs(xi / plain[i][1] - 1)
        print(n_prim, i, ngal[0] == plain[i][0][0], d.max(), np.nonzero(d[0])[0][:8])
    h.set_resident(False)
