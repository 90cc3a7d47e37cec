#!/usr/bin/env python3
"""A/B of the grouped occupation evaluation (option "grouped": bins that share their quadrature
nodes have them evaluated once per group) on the shapes that have such bins: us per step of
device-resident calls, grouped / per bin.

  cfg3      BASELINE configs[2]: 50 x 2 x {cen,sat} auto table, separate + assembly bias
  cfg3tot   the same table, total prediction, no decoration
  ds        the reference's AbacusSummit fixture (ds_efficient.hdf5: cross, G = 1104, K = 4
            interpolator, R = 13)
  db        database-shaped synthetic: 30 x 2 x {cen,sat} auto (G = 120), 4 x 4 x 4 = 64 tables
  wp        the reference's example table (bolplanck_wp.hdf5, G = 60: no groups -- unchanged)
  cfg2      BASELINE configs[1] (no groups -- unchanged)
"""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

from bench import Device, sustained          # noqa: E402
from tabcorr_amd import TabCorr, Interpolator, synthetic, _lib          # noqa: E402

lib = _lib.load()
_lib.require_device()
dev = Device(lib, _lib)


def make(table, **kwargs):
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                               table['attrs'], **kwargs)


def ab_table(name, halotab, theta, flags, n_comp, sizes=(10000, 4096, 1024)):
    device = halotab.to_device()
    h = device.handle
    n_r = int(np.prod(halotab.tpcf_shape))
    d_theta = dev.upload(theta)
    d_ngal, d_xi = dev.malloc(2 * len(theta)), dev.malloc(n_comp * n_r * len(theta))
    for n in SIZES or sizes:
        row = []
        for grouped in SETTINGS:
            _lib.check(lib.tc_table_set_option(h, b'grouped', grouped))
            seconds = sustained(
                lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                    h, d_theta, theta.shape[1], n, 10, flags, d_ngal, d_xi)),
                lambda: _lib.check(lib.tc_table_synchronize(h)), seconds=0.3)
            shape = [ctypes.c_int() for _ in range(4)]
            lib.tc_table_last_launch(h, *[ctypes.byref(v) for v in shape])
            row.append((seconds * 1e6, shape[0].value, shape[1].value, shape[2].value))
        if len(row) == 1:
            row.append(row[0])
        print('%-8s %6d draws: grouped %8.1f us  per bin %8.1f us  (x%.2f)  launch %s' % (
            name, n, row[0][0], row[1][0], row[1][0] / row[0][0], row[0][1:]), flush=True)


def ab_interp(name, interp, theta, x, sizes=(10000, 1024)):
    device = interp.to_device()
    h = device.handle
    n_r = int(np.prod(interp.tabcorr_list[0].tpcf_shape))
    d_theta, d_x = dev.upload(theta), dev.upload(x)
    d_ngal, d_xi = dev.malloc(len(theta)), dev.malloc(n_r * len(theta))
    for n in SIZES or sizes:
        row = []
        for grouped in SETTINGS:
            for table in device.tables:
                _lib.check(lib.tc_table_set_option(table.handle, b'grouped', grouped))
            seconds = sustained(
                lambda: _lib.check(lib.tc_interp_predict_zheng07_batch_device(
                    h, d_theta, 5, d_x, n, 10, 0, d_ngal, d_xi)),
                lambda: _lib.check(lib.tc_interp_synchronize(h)), seconds=0.3)
            row.append(seconds * 1e6)
        if len(row) == 1:
            row.append(row[0])
        print('%-8s %6d draws: grouped %8.1f us  per bin %8.1f us  (x%.2f)' % (
            name, n, row[0], row[1], row[1] / row[0]), flush=True)


# (R04_SIZES=10000 R04_GROUPED=1: one batch size, one setting -- for rocprofv3 passes)
SIZES = tuple(int(v) for v in os.environ['R04_SIZES'].split(',')) if 'R04_SIZES' in os.environ else None
SETTINGS = tuple(int(v) for v in os.environ.get('R04_GROUPED', '1,0').split(','))
which = sys.argv[1:] or ['cfg3', 'cfg3tot', 'ds', 'db', 'wp', 'cfg2']
theta = synthetic.zheng07_draws(10000, seed=1)
rng = np.random.default_rng(0)
if 'cfg3' in which or 'cfg3tot' in which:
    table3 = synthetic.synthetic_table(50, 2, (19, ), 'auto', seed=3)
    theta7 = np.hstack([theta, rng.uniform(-1, 1, (10000, 2))])
    if 'cfg3' in which:
        ab_table('cfg3', make(table3), theta7, 1 | 4, 3)
    if 'cfg3tot' in which:
        ab_table('cfg3tot', make(table3), theta, 0, 1)
if 'ds' in which:
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    x = np.ascontiguousarray(np.stack([rng.uniform(xp[0], xp[-1], size=10000)
                                       for xp in interp.xp], axis=-1))
    th = theta.copy()
    th[:, 0] = rng.uniform(12.5, 13.3, 10000)
    th[:, 3] = rng.uniform(13.6, 14.4, 10000)
    if 'R04_DS_ONE' not in os.environ:
        ab_interp('ds K=4', interp, th, x)
    ab_table('ds one', interp.tabcorr_list[0], th, 0, 1)
if 'db' in which:
    tables, keys, points = synthetic.synthetic_interpolator((4, 4, 4), 30, 2, (19, ), 'auto',
                                                            seed=11)
    interp = Interpolator([make(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    x = np.ascontiguousarray(np.stack([rng.uniform(xp[0], xp[-1], size=10000)
                                       for xp in interp.xp], axis=-1))
    ab_interp('db K=64', interp, theta, x)
    ab_table('db one', make(tables[0]), theta, 0, 1)
if 'wp' in which:
    ab_table('wp G=60', TabCorr.read(os.path.join(REPO, 'tests', 'golden', 'bolplanck_wp.hdf5')),
             theta, 0, 1)
if 'cfg2' in which:
    ab_table('cfg2', make(synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)), theta, 0, 1)
dev.free_all()
