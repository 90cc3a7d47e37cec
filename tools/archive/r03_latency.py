#!/usr/bin/env python3
"""Un-batched latency (README.md:72-75 usage): predict(model) through Python, the C call
for one draw, and n walkers per call through the one-launch path (tc_predict_zheng07_many)
against the three-kernel batched path -- where the crossover lies.

    gpurun -- python3 tools/archive/r03_latency.py
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tabcorr_amd import TabCorr, Interpolator, Zheng07Model, synthetic, _lib   # noqa: E402


def timeit(call, seconds=0.25, warm=200):
    for _ in range(warm):
        call()
    t0 = time.perf_counter()
    call()
    per = max(time.perf_counter() - t0, 1e-6)
    n = max(20, int(seconds / per))
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    lib = _lib.load()
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    device = halotab.to_device()
    handle = device.handle
    model = Zheng07Model()
    out = {}
    out['predict_model_us'] = timeit(lambda: halotab.predict(model))
    theta = synthetic.zheng07_draws(4096, seed=3)
    for n in (1, 2, 4, 8, 16, 32, 64):
        ngal, xi = np.empty(n), np.empty((n, 19))
        t = np.ascontiguousarray(theta[:n])
        args = (handle, _lib.as_double_p(t), 5, n, 10, 0, _lib.as_double_p(ngal),
                _lib.as_double_p(xi))
        many = timeit(lambda: lib.tc_predict_zheng07_many(*args))
        check = xi.copy()
        _lib.check(lib.tc_table_set_option(handle, b'single_draw', 0))
        batch = timeit(lambda: lib.tc_predict_zheng07_batch(*args))
        _lib.check(lib.tc_table_set_option(handle, b'single_draw', 1))
        assert np.allclose(check, xi, rtol=1e-12, atol=0)
        out['n=%d' % n] = {'one_launch_us': many, 'per_walker_us': many / n,
                           'three_kernels_us': batch}
    tables, keys, points = synthetic.synthetic_interpolator((5, 5), 50, 1, (19, ), 'auto', seed=7)
    interp = Interpolator([TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'],
                                               t['attrs']) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    for d, key in enumerate(keys):
        model.param_dict[key] = float(np.mean(points[:, d]))
    out['interp5x5_predict_model_us'] = timeit(lambda: interp.predict(model), warm=50)
    first = interp.to_device().tables[0].handle
    for poll in (0, 1, 0, 1):
        _lib.check(lib.tc_table_set_option(first, b'poll_done', poll))
        _lib.check(lib.tc_table_set_option(handle, b'poll_done', poll))
        out.setdefault('poll=%d' % poll, []).append(
            {'interp5x5_us': timeit(lambda: interp.predict(model), warm=50),
             'predict_model_us': timeit(lambda: halotab.predict(model))})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
