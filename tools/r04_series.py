#!/usr/bin/env python3
"""A/B of the moment expansion of the central bins' node sums (option "series"): us per step of
device-resident pipelined calls with the expansion on / off, and the largest difference of the
results.  Shapes: cfg2 (BASELINE configs[1]), wp (bolplanck G = 60), ds1 / ds4 (AbacusSummit
table / interpolator), posterior (cfg2's table, draws clustered as a sampler's ensemble)."""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import Device, sustained          # noqa: E402
from tabcorr_amd import TabCorr, Interpolator, synthetic, _lib          # noqa: E402

lib = _lib.load()
_lib.require_device()
dev = Device(lib, _lib)
rng = np.random.default_rng(0)
theta = synthetic.zheng07_draws(10000, seed=1)


def ab(name, handles, call, sync, download):
    rows = []
    for on in (1, 0):
        for h in handles:
            _lib.check(lib.tc_table_set_option(h, b'series', on))
        seconds = sustained(call, sync, seconds=0.4)
        call()
        sync()
        rows.append((seconds * 1e6, download()))
    diff = np.max(np.abs(rows[0][1] - rows[1][1]) / np.maximum(np.abs(rows[1][1]),
                                                              1e-14 * np.max(np.abs(rows[1][1]))))
    print('%-10s series %7.2f us   node loop %7.2f us   (x%.3f)   max rel diff %.1e' % (
        name, rows[0][0], rows[1][0], rows[1][0] / rows[0][0], diff), flush=True)


def table_case(name, halotab, th, n=10000):
    h = halotab.to_device().handle
    n_r = int(np.prod(halotab.tpcf_shape))
    d_theta = dev.upload(th)
    d_ngal, d_xi = dev.malloc(n), dev.malloc(n * n_r)
    ab(name, [h],
       lambda: _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, n, 10, 0, d_ngal,
                                                              d_xi)),
       lambda: _lib.check(lib.tc_table_synchronize(h)), lambda: dev.download(d_xi, n * n_r))


def make(table):
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                               table['attrs'])


which = sys.argv[1:] or ['cfg2', 'posterior', 'wp', 'ds1', 'ds4', 'cfg3tot']
if 'cfg2' in which:
    table_case('cfg2', make(synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)), theta)
if 'posterior' in which:
    centre = np.array([12.4, 0.35, 11.9, 13.3, 1.1])
    th = centre + rng.normal(0, 1, (10000, 5)) * np.array([0.03, 0.02, 0.1, 0.04, 0.03])
    table_case('posterior', make(synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)), th)
if 'wp' in which:
    table_case('wp G=60', TabCorr.read(os.path.join(REPO, 'tests', 'golden', 'bolplanck_wp.hdf5')),
               theta)
if 'cfg3tot' in which:
    table_case('cfg3tot', make(synthetic.synthetic_table(50, 2, (19, ), 'auto', seed=3)), theta)
if 'ds1' in which or 'ds4' in which:
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    th = theta.copy()
    th[:, 0] = rng.uniform(12.5, 13.3, 10000)
    th[:, 3] = rng.uniform(13.6, 14.4, 10000)
    if 'ds1' in which:
        table_case('ds1', interp.tabcorr_list[0], th)
    if 'ds4' in which:
        x = np.ascontiguousarray(np.stack([rng.uniform(xp[0], xp[-1], size=10000)
                                           for xp in interp.xp], axis=-1))
        device = interp.to_device()
        h = device.handle
        d_theta, d_x = dev.upload(th), dev.upload(x)
        d_ngal, d_xi = dev.malloc(10000), dev.malloc(13 * 10000)
        ab('ds4', [t.handle for t in device.tables],
           lambda: _lib.check(lib.tc_interp_predict_zheng07_batch_device(
               h, d_theta, 5, d_x, 10000, 10, 0, d_ngal, d_xi)),
           lambda: _lib.check(lib.tc_interp_synchronize(h)),
           lambda: dev.download(d_xi, 13 * 10000))
dev.free_all()
