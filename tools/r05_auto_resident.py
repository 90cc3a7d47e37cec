#!/usr/bin/env python3
"""The resident kernel by itself (option "resident" = 2, the default): a plain loop of
predict(model) with no set_resident, the same with the option off / on, a caller that
synchronises the device between its calls (what it waits, what a call costs) and a caller that
pauses 400 us between calls.  gpurun -- python3 tools/r05_auto_resident.py"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabcorr_amd import TabCorr, Zheng07Model, synthetic, _lib   # noqa: E402

lib = _lib.load()
for n_prim in (30, 50):
    table = synthetic.synthetic_table(n_prim, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    model = Zheng07Model()
    halotab.predict(model)
    count = [0]

    def call():
        count[0] += 1
        model.param_dict['logMmin'] = 12.0 + 1e-5 * (count[0] % 1000)
        return halotab.predict(model)

    def loop(n=20000):
        for _ in range(200):
            call()
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        return (time.perf_counter() - t0) / n * 1e6
    expect = call()
    for mode in ('auto', False, True, 'auto'):
        halotab.set_resident(mode)
        us = loop()
        got = call()
        print('G = %3d  resident = %-5s: %6.2f us per predict(model) = %.3g calls/s   same bits as '
              'the first call: %s' % (2 * n_prim, mode, us, 1e6 / us,
                                      np.array_equal(got[1], halotab.predict(model)[1])), flush=True)
    # a caller that synchronises the device between its calls
    halotab.set_resident('auto')
    waits, calls = [], []
    for i in range(3000):
        t0 = time.perf_counter()
        call()
        t1 = time.perf_counter()
        _lib.check(lib.tc_device_synchronize())
        t2 = time.perf_counter()
        calls.append(t1 - t0)
        waits.append(t2 - t1)
    waits, calls = np.array(waits) * 1e6, np.array(calls) * 1e6
    print('        device-wide synchronisation after every call: call %.1f us (median), the '
          'synchronisation waits %.1f us (median), %.1f us (max), first 40 calls max %.1f us; '
          'last 2000 calls: median wait %.1f us' % (np.median(calls), np.median(waits), waits.max(),
                                                    waits[:40].max(), np.median(waits[-2000:])),
          flush=True)
    # a caller that pauses longer than the idle time between its calls
    t_call = []
    for i in range(600):
        t0 = time.perf_counter()
        call()
        t_call.append(time.perf_counter() - t0)
        end = time.perf_counter() + 400e-6
        while time.perf_counter() < end:
            pass
    print('        400 us between calls: %.1f us per call (median), %.1f (max)'
          % (np.median(t_call) * 1e6, np.max(t_call) * 1e6), flush=True)
