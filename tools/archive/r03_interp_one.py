#!/usr/bin/env python3
"""Un-batched Interpolator.predict(model) over grids of BASELINE configs[1] tables: the launch
sized for ONE round of workgroups (option "single_round" of the first table) against one pass
per workgroup.  gpurun -- python3 tools/archive/r03_interp_one.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, Interpolator, Zheng07Model, synthetic, _lib   # noqa: E402
lib = _lib.load()
for shape in ((5, 5), (4, 4), (4, 4, 4), (6, 6)):
    tables, keys, points = synthetic.synthetic_interpolator(shape, 50, 1, (19, ), 'auto', seed=7)
    make = lambda t: TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs'])
    interp = Interpolator([make(t) for t in tables], {k: points[:, d] for d, k in enumerate(keys)})
    model = Zheng07Model()
    for d, key in enumerate(keys):
        model.param_dict[key] = float(np.mean(points[:, d]))
    first = interp.predict(model)
    handle = interp.to_device().tables[0].handle
    row = []
    for single_round in (0, 1, 0, 1):
        _lib.check(lib.tc_table_set_option(handle, b'single_round', single_round))
        got = interp.predict(model)
        assert np.allclose(got[1], first[1], rtol=1e-12)
        for _ in range(500):
            interp.predict(model)
        t0 = time.perf_counter()
        for _ in range(3000):
            interp.predict(model)
        row.append((time.perf_counter() - t0) / 3000 * 1e6)
    print('%s tables: one pass per workgroup %.2f / %.2f us, one round %.2f / %.2f us' % (
        'x'.join(map(str, shape)), row[0], row[2], row[1], row[3]), flush=True)
