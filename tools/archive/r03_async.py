#!/usr/bin/env python3
"""Host-to-host rates of cfg2 (10^4 draws per call): the synchronous entry point against
the asynchronous one at several pipeline depths (tc_predict_zheng07_batch_async +
tc_table_wait on page-locked buffers), prediction and fused likelihood, plus the
device-resident step with ordered / unordered finalisations.

    gpurun -- python3 tools/archive/r03_async.py [--draws 10000] [--seconds 0.5]
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from tabcorr_amd import TabCorr, synthetic, _lib, pinned_empty, pinned_array   # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--draws', type=int, default=10000)
    parser.add_argument('--seconds', type=float, default=0.5)
    parser.add_argument('--lanes', type=int, default=0)
    parser.add_argument('--option', action='append', default=[],
                        help='tc_table_set_option name=value (repeatable)')
    parser.add_argument('--adjacent', type=int, default=0,
                        help='1: ngal and xi of a call adjacent in one pinned block')
    parser.add_argument('--only', default='',
                        help='e.g. predict:4:400 = only the prediction at depth 4, 400 calls '
                             '(for a rocprofv3 trace)')
    parser.add_argument('--depths', default='1,2,4,6,8')
    parser.add_argument('--device-step', type=int, default=1)
    args = parser.parse_args()
    args.depths = [int(d) for d in args.depths.split(',')]
    lib = _lib.load()
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    device = halotab.to_device()
    handle = device.handle
    if args.lanes:
        _lib.check(lib.tc_table_set_option(handle, b'lanes', args.lanes))
    for option in args.option:
        name, value = option.split('=')
        _lib.check(lib.tc_table_set_option(handle, name.encode(), int(value)))
    n = args.draws
    n_r = 19
    out = {'draws_per_call': n, 'options': args.option, 'lanes': args.lanes,
           'adjacent': args.adjacent}

    theta = synthetic.zheng07_draws(n, seed=1)
    ngal, xi = np.empty(n), np.empty((n, n_r))

    def sync_call():
        _lib.check(lib.tc_predict_zheng07_batch(handle, _lib.as_double_p(theta), 5, n, 10, 0,
                                                _lib.as_double_p(ngal), _lib.as_double_p(xi)))
    for _ in range(20):
        sync_call()
    t0 = time.perf_counter()
    count = 0
    while time.perf_counter() - t0 < args.seconds:
        sync_call()
        count += 1
    per = (time.perf_counter() - t0) / count
    out['sync'] = {'us_per_call': per * 1e6, 'calls_per_sec': n / per}

    ring = 8
    thetas = [pinned_array(synthetic.zheng07_draws(n, seed=10 + i)) for i in range(ring)]
    if args.adjacent:
        blocks = [pinned_empty(n * (1 + n_r)) for _ in range(ring)]
        ngals = [b[:n] for b in blocks]
        xis = [b[n:].reshape(n, n_r) for b in blocks]
    else:
        ngals = [pinned_empty(n) for _ in range(ring)]
        xis = [pinned_empty((n, n_r)) for _ in range(ring)]
    chis = [pinned_empty(n) for _ in range(ring)]
    data = np.full(n_r, 50.0)
    precision = np.eye(n_r) * 1e-2
    data_p, precision_p = _lib.as_double_p(data), _lib.as_double_p(precision)
    p_theta = [_lib.as_double_p(a) for a in thetas]
    p_ngal = [_lib.as_double_p(a) for a in ngals]
    p_xi = [_lib.as_double_p(a) for a in xis]
    p_chi = [_lib.as_double_p(a) for a in chis]

    def run(depth, chi2, total):
        tickets = [None] * ring
        ticket = ctypes.c_int64()
        ref = ctypes.byref(ticket)
        enqueue_time = 0.0
        t0 = time.perf_counter()
        for k in range(total):
            s = k % ring
            if k >= depth:
                _lib.check(lib.tc_table_wait(handle, tickets[(k - depth) % ring]))
            t1 = time.perf_counter()
            if chi2:
                status = lib.tc_chi2_zheng07_batch_async(handle, p_theta[s], 5, n, 10, 0, data_p,
                                                         precision_p, p_ngal[s], p_chi[s], ref)
            else:
                status = lib.tc_predict_zheng07_batch_async(handle, p_theta[s], 5, n, 10, 0,
                                                            p_ngal[s], p_xi[s], ref)
            enqueue_time += time.perf_counter() - t1
            _lib.check(status)
            tickets[s] = ticket.value
        for k in range(max(0, total - depth), total):
            _lib.check(lib.tc_table_wait(handle, tickets[k % ring]))
        spent = time.perf_counter() - t0
        return spent / total, enqueue_time / total

    if args.only:
        kind, depth, total = args.only.split(':')
        run(int(depth), kind == 'chi2', 200)
        per, enq = run(int(depth), kind == 'chi2', int(total))
        print(json.dumps({'us_per_call': per * 1e6, 'enqueue_us': enq * 1e6}))
        return
    for chi2 in (False, True):
        for depth in args.depths:
            run(depth, chi2, 200)
            per, _ = run(depth, chi2, 50)
            total = max(50, int(args.seconds / per))
            per, enq = run(depth, chi2, total)
            out['%s_depth%d' % ('chi2' if chi2 else 'predict', depth)] = {
                'us_per_call': per * 1e6, 'calls_per_sec': n / per, 'enqueue_us': enq * 1e6}

    # parity of the last ring contents against the synchronous call
    for s in (0, ring - 1):
        expect = halotab.predict_batch(np.array(thetas[s]))
        assert np.allclose(xis[s], expect[1], rtol=1e-12, atol=0), s

    if not args.device_step:
        print(json.dumps(out))
        return
    # device-resident step, ordered against unordered finalisations
    d_theta, d_ngal, d_xi = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    for ptr, count in ((d_theta, n * 5), (d_ngal, 4 * n), (d_xi, 4 * n * n_r)):
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, thetas[0].ctypes.data_as(ctypes.c_void_p), n * 5 * 8))
    for ordered in (1, 0):
        _lib.check(lib.tc_table_set_option(handle, b'ordered', ordered))

        def step(k):
            s = k % 4
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n, 10, 0, ctypes.c_void_p(d_ngal.value + s * n * 8),
                ctypes.c_void_p(d_xi.value + s * n * n_r * 8)))
        for k in range(3000):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        total = 8000
        t0 = time.perf_counter()
        for k in range(total):
            step(k)
        _lib.check(lib.tc_table_synchronize(handle))
        per = (time.perf_counter() - t0) / total
        out['device_ordered%d' % ordered] = {'us_per_step': per * 1e6, 'calls_per_sec': n / per}
    _lib.check(lib.tc_table_set_option(handle, b'ordered', 1))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
