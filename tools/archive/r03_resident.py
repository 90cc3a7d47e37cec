#!/usr/bin/env python3
"""Un-batched calls: one launch per call against the resident kernel (option "resident"),
C entry point and Python predict(model); us per call.
gpurun -- python3 tools/archive/r03_resident.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    row = {}
    for resident in (0, 1, 0, 1, 2, 4):
        if resident > 1:
            _lib.check(lib.tc_table_set_option(handle, b'resident_poll_waves', resident))
        halotab.set_resident(bool(resident))

        def c_call():
            _lib.check(lib.tc_predict_zheng07_batch(
                handle, _lib.as_double_p(theta), 5, 1, 10, 0, _lib.as_double_p(ngal),
                _lib.as_double_p(xi)))
        for call, name in ((c_call, 'C'), (lambda: halotab.predict(model), 'predict(model)')):
            for _ in range(2000):
                call()
            t0 = time.perf_counter()
            n = 20000
            for _ in range(n):
                call()
            row.setdefault((name, resident), []).append((time.perf_counter() - t0) / n * 1e6)
    import ctypes
    ticks = (ctypes.c_uint64 * 64)()
    count = ctypes.c_int64()
    _lib.check(lib.tc_debug_resident_ticks(handle, ticks, 64, ctypes.byref(count)))
    print('  workgroups %d, device time per call (sight of the parameters -> completion word): '
          '%s us' % (count.value, ' '.join('%.2f' % (ticks[b] / 100.0) for b in range(count.value))))
    halotab.set_resident(False)
    print('G = %3d: ' % (2 * n_prim) + '; '.join(
        '%s %s: %s us' % (name, ('resident' if resident == 1 else 'resident, %d polling waves' % resident) if resident else 'one launch per call',
                          ' / '.join('%.2f' % v for v in values))
        for (name, resident), values in sorted(row.items())), flush=True)
