"""Drop-in version of the reference's clustering example (docs/examples/example_wp.py
of johannesulf/TabCorr): read a tabulated w_p table, predict for one model, then sweep a
parameter -- once through the reference's scalar ``predict`` signature, once through the
batched call.

    python examples/example_wp.py tests/golden/bolplanck_wp.hdf5
"""

import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tabcorr_amd import TabCorr, Zheng07Model  # noqa: E402

fname = sys.argv[1] if len(sys.argv) > 1 else 'tests/golden/bolplanck_wp.hdf5'
halotab = TabCorr.read(fname)

# halotools users pass PrebuiltHodModelFactory('zheng07', ...) here instead
model = Zheng07Model(prim_haloprop_key=halotab.attrs['prim_haloprop_key'],
                     redshift=halotab.attrs['redshift'])
ngal, wp = halotab.predict(model)
print('ngal = %.4e, wp[:3] =' % ngal, wp[:3])

ngal_sep, wp_sep = halotab.predict(model, separate_gal_type=True)
for key in wp_sep:
    print('%-24s wp[0] = %8.2f' % (key, wp_sep[key][0]))

# the reference's usage pattern: 1000 sequential calls ...
values = np.linspace(12.0, 13.0, 1000)
start = time.perf_counter()
for value in values:
    model.param_dict['logM1'] = value
    ngal, wp = halotab.predict(model)
print('1000 predict(model) calls: %.1f ms' % ((time.perf_counter() - start) * 1e3))

# ... and the same sweep as one batch
theta = np.tile([model.param_dict[k] for k in
                 ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')], (1000, 1))
theta[:, 3] = values
start = time.perf_counter()
ngal_all, wp_all = halotab.predict_batch(theta)
print('predict_batch of 1000 draws: %.2f ms' % ((time.perf_counter() - start) * 1e3))
assert np.allclose(wp_all[-1], wp, rtol=1e-12)
