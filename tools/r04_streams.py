#!/usr/bin/env python3
"""How well the launches of a table's four lanes overlap depends on WHICH streams of the
process they are: us per call of pipelined device-resident calls for a table created first in
the process, or after N dummy tables (four streams each).  Usage: r04_streams.py [N] [shape]
shape: cfg2 (BASELINE configs[1], default) | ds1 (AbacusSummit table) | wp"""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import Device, sustained          # noqa: E402
from tabcorr_amd import TabCorr, Interpolator, synthetic, _lib          # noqa: E402

lib = _lib.load()
dev = Device(lib, _lib)
n_dummy = int(sys.argv[1]) if len(sys.argv) > 1 else 0
shape = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
dummies = []
if 'touch' in sys.argv:
    # a copy on the null stream before anything else (what a caller who uploads draws before
    # creating the table does)
    dev.upload(np.zeros(16))
if 'streams' in sys.argv:
    # N idle streams created straight through the HIP runtime before the table exists
    hip = ctypes.CDLL('libamdhip64.so')
    _lib.require_device()
    for k in range(n_dummy):
        stream = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0
        dummies.append(stream)
else:
    for k in range(n_dummy):
        tb = synthetic.synthetic_table(3, 1, (2, ), 'auto', seed=k)
        d = TabCorr.from_arrays(tb['gal_type'], tb['tpcf_matrix'], tb['tpcf_shape'], tb['attrs'])
        d.to_device()
        dummies.append(d)
theta = synthetic.zheng07_draws(40000, seed=1)
if shape == 'cfg2':
    tb = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(tb['gal_type'], tb['tpcf_matrix'], tb['tpcf_shape'], tb['attrs'])
elif shape == 'wp':
    halotab = TabCorr.read(os.path.join(REPO, 'tests', 'golden', 'bolplanck_wp.hdf5'))
else:
    halotab = Interpolator.read(os.path.join(REPO, 'tests', 'golden',
                                             'ds_efficient.hdf5')).tabcorr_list[0]
    rng = np.random.default_rng(0)
    theta[:, 0] = rng.uniform(12.5, 13.3, len(theta))
    theta[:, 3] = rng.uniform(13.6, 14.4, len(theta))
h = halotab.to_device().handle
if 'destroy' in sys.argv:
    for stream in dummies:
        assert hip.hipStreamDestroy(stream) == 0
n_r = int(np.prod(halotab.tpcf_shape))
d_theta = dev.upload(theta)
d_ngal, d_xi = dev.malloc(len(theta)), dev.malloc(n_r * len(theta))
for n in (256, 1024, 10000):
    s = sustained(lambda: _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, n, 10, 0,
                                                                         d_ngal, d_xi)),
                  lambda: _lib.check(lib.tc_table_synchronize(h)), seconds=0.3)
    print('%s after %d dummy tables: %6d draws %7.2f us' % (shape, n_dummy, n, s * 1e6),
          flush=True)
