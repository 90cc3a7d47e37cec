#!/usr/bin/env python3
"""Un-batched predict(model) through the resident kernel: mailbox in page-locked host memory
against device memory behind the PCIe aperture (option "resident_aperture"), 1 / 2 / 4 polling
waves; us per call (C entry point through ctypes, predict(model)) and agreement with one
launch per call."""
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
    handle = halotab.to_device().handle
    theta = np.ascontiguousarray(synthetic.zheng07_draws(1, seed=1))
    ngal, xi = np.zeros(1), np.zeros(19)
    model = Zheng07Model(redshift=table['attrs']['redshift'])
    expect = halotab.predict(model)

    def c_call():
        _lib.check(lib.tc_predict_zheng07_batch(
            handle, _lib.as_double_p(theta), 5, 1, 10, 0, _lib.as_double_p(ngal),
            _lib.as_double_p(xi)))

    def timed(call, n=20000):
        for _ in range(2000):
            call()
        t0 = time.perf_counter()
        for _ in range(n):
            call()
        return (time.perf_counter() - t0) / n * 1e6
    halotab.set_resident(False)
    print('G = %d: one launch per call: C %.2f us, predict(model) %.2f us'
          % (2 * n_prim, timed(c_call), timed(lambda: halotab.predict(model))), flush=True)
    for aperture in (0, 1):
        for waves in (1, 2, 4):
            _lib.check(lib.tc_table_set_option(handle, b'resident_aperture', aperture))
            _lib.check(lib.tc_table_set_option(handle, b'resident_poll_waves', waves))
            halotab.set_resident(True)
            got = halotab.predict(model)
            worst = max(abs(got[0] / expect[0] - 1), np.max(np.abs(got[1] / expect[1] - 1)))
            print('G = %d: resident, mailbox in %s, %d polling wave(s): C %.2f us, '
                  'predict(model) %.2f us (largest difference to one launch per call %.1e)'
                  % (2 * n_prim, 'device memory' if aperture else 'page-locked memory', waves,
                     timed(c_call), timed(lambda: halotab.predict(model)), worst), flush=True)
    halotab.set_resident(False)
