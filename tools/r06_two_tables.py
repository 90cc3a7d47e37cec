#!/usr/bin/env python3
"""Round 6, VERDICT r05 item 2: the reference's documented per-step usage -- TWO tables per
likelihood evaluation (docs/guides/overview.rst:86-92: halotab_wp.predict(model), then
halotab_ds.predict(model)) -- with default options: bolplanck_wp (mode auto) and bolplanck_ds
(mode cross) alternately, per-pair time, per-call time distribution, agreement with the launched
path, device-wide synchronisations in between.  gpurun -- python3 tools/r06_two_tables.py"""
import ctypes
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tabcorr_amd import TabCorr, Zheng07Model, _lib   # noqa: E402

lib = _lib.load()
golden = os.path.join(REPO, 'tests', 'golden')
wp = TabCorr.read(os.path.join(golden, 'bolplanck_wp.hdf5'))
ds = TabCorr.read(os.path.join(golden, 'bolplanck_ds.hdf5'))
model = Zheng07Model(redshift=wp.attrs['redshift'])
rng = np.random.default_rng(0)
thetas = np.column_stack([rng.uniform(11.8, 12.6, 400), rng.uniform(0.2, 0.6, 400),
                          rng.uniform(11.0, 12.0, 400), rng.uniform(13.0, 13.8, 400),
                          rng.uniform(0.9, 1.2, 400)])
keys = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')


def set_theta(i):
    for key, value in zip(keys, thetas[i % 400]):
        model.param_dict[key] = value


for tab in (wp, ds):
    tab.set_resident(False)
launched = []
for i in range(400):
    set_theta(i)
    launched.append((wp.predict(model), ds.predict(model)))


def loop(n, label, joint=None):
    times = np.empty(n)
    worst = (0.0, -1)
    bad = 0
    for i in range(n):
        set_theta(i)
        t0 = time.perf_counter()
        if joint is not None:
            a, b = joint(model)
        else:
            a = wp.predict(model)
            b = ds.predict(model)
        times[i] = time.perf_counter() - t0
        if not (np.array_equal(a[1], launched[i % 400][0][1]) and
                np.array_equal(b[1], launched[i % 400][1][1])):
            bad += 1
    us = times * 1e6
    stats = []
    for tab in (wp, ds):
        values = [ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()]
        running = ctypes.c_int()
        lib.tc_table_resident_stats(tab.to_device().handle, *[ctypes.byref(v) for v in values],
                                    ctypes.byref(running))
        stats.append(tuple(v.value for v in values) + (running.value, ))
    print('    resident (launches, relaunches, fall-backs, running) wp %s ds %s' % tuple(stats))
    print('%-44s pair: median %6.2f us  mean %6.2f  p99 %7.1f  max %8.1f   results differing '
          'from the launched path: %d of %d' % (label, np.median(us), us.mean(),
                                                np.percentile(us, 99), us.max(), bad, n), flush=True)


loop(3000, 'one launch per call (resident off)')
for tab in (wp, ds):
    tab.set_resident('auto')
loop(3000, 'default options (resident by itself)')
loop(3000, 'default options, second loop')
# a caller that synchronises the device after every pair
waits = []
for i in range(1500):
    set_theta(i)
    wp.predict(model)
    ds.predict(model)
    t0 = time.perf_counter()
    _lib.check(lib.tc_device_synchronize())
    waits.append(time.perf_counter() - t0)
waits = np.array(waits) * 1e6
print('device-wide synchronisation after every pair: median wait %.1f us, max %.1f us, last 500 '
      'median %.1f us' % (np.median(waits), waits.max(), np.median(waits[-500:])), flush=True)
for tab in (wp, ds):
    tab.set_resident(True)
loop(3000, 'option resident set on both')
if hasattr(TabCorr, 'predict_joint'):
    for tab in (wp, ds):
        tab.set_resident('auto')
    loop(3000, 'predict_joint, default options', joint=lambda m: TabCorr.predict_joint([wp, ds], m))
    loop(3000, 'predict_joint, second loop', joint=lambda m: TabCorr.predict_joint([wp, ds], m))
    for tab in (wp, ds):
        tab.set_resident(False)
    loop(3000, 'predict_joint, resident off', joint=lambda m: TabCorr.predict_joint([wp, ds], m))
