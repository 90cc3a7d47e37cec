"""Developer sweep: time the batched predict path under different launch
decompositions (env knobs read by the library).  Not part of the product."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib  # noqa: E402


def run(n_prim, n_sec, tpcf_shape, n_draws, steps, label, flags=0, dtype='float64'):
    table = synthetic.synthetic_table(n_prim, n_sec, tpcf_shape, 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                  table['tpcf_shape'], table['attrs'],
                                  compute_dtype=dtype)
    dev = halotab.to_device()
    lib = dev.lib
    theta = synthetic.zheng07_draws(n_draws, seed=1)
    n_r = dev.n_r
    d_theta = ctypes.c_void_p()
    d_ngal = ctypes.c_void_p()
    d_xi = ctypes.c_void_p()
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_theta), theta.nbytes))
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_ngal), n_draws * 2 * 8))
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_xi), n_draws * 3 * n_r * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))

    def step():
        _lib.check(lib.tc_predict_zheng07_batch_device(
            dev.handle, d_theta, 5, n_draws, 10, flags, d_ngal, d_xi))
    for _ in range(3):
        step()
    _lib.check(lib.tc_table_synchronize(dev.handle))
    ms = ctypes.c_float()
    _lib.check(lib.tc_table_timer_begin(dev.handle, 0))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _lib.check(lib.tc_table_timer_end(dev.handle, ctypes.byref(ms)))
    wall = (time.perf_counter() - t0) * 1e3
    _lib.check(lib.tc_table_timer_begin(dev.handle, 1))
    for _ in range(steps):
        step()
    ms2 = ctypes.c_float()
    _lib.check(lib.tc_table_timer_end(dev.handle, ctypes.byref(ms2)))
    n = ctypes.c_int()
    kms = ctypes.c_float()
    _lib.check(lib.tc_table_kernel_time(dev.handle, ctypes.byref(n), ctypes.byref(kms)))
    wg = ctypes.c_int(); wv = ctypes.c_int(); sp = ctypes.c_int(); lds = ctypes.c_int()
    lib.tc_table_last_launch(dev.handle, ctypes.byref(wg), ctypes.byref(wv), ctypes.byref(sp), ctypes.byref(lds))
    G = 2 * n_prim * n_sec
    P = G * (G + 1) // 2
    R = int(np.prod(tpcf_shape))
    flops = n_draws * (2.0 * R * P + 3 * P)
    print('%-28s step %8.1f us (wall %8.1f) | contract %8.1f us = %6.2f TF/s | %6.3g calls/s | wg=%d waves=%d groups=%d lds=%d' % (
        label, ms.value / steps * 1e3, wall / steps * 1e3, kms.value * 1e3,
        flops / (kms.value * 1e-3) / 1e12, n_draws / (ms.value / steps * 1e-3),
        wg.value, wv.value, sp.value, lds.value), flush=True)
    for p in (d_theta, d_ngal, d_xi):
        lib.tc_device_free(p)


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
    if which == 'cfg2':
        for tw in [1024, 2048, 4096, 8192]:
            for nw in [4, 8, 16]:
                os.environ['TC_TARGET_WAVES'] = str(tw)
                os.environ['TC_NWAVES'] = str(nw)
                run(50, 1, (19, ), 10000, 20, 'cfg2 tw=%d nw=%d' % (tw, nw))
    elif which == 'cfg3':
        for tw in [2048, 4096, 8192]:
            for nw in [8, 16]:
                os.environ['TC_TARGET_WAVES'] = str(tw)
                os.environ['TC_NWAVES'] = str(nw)
                run(50, 2, (19, ), 10000, 10, 'cfg3 tw=%d nw=%d' % (tw, nw), flags=1)
    elif which == 'big':
        run(50, 1, (19, ), 100000, 10, 'cfg2 B=1e5')
        run(50, 1, (19, ), 1000000, 3, 'cfg2 B=1e6')
    elif which == 'one':
        run(50, 1, (19, ), 10000, 50, 'cfg2 default')
    elif which == 'stats':
        run(50, 1, (19, ), 10000, 200, 'cfg2 default')
    elif which == 'groups':
        for nw in [4, 8]:
            for ng in [0, 4, 6, 8, 10, 12, 16]:
                os.environ['TC_NWAVES'] = str(nw)
                os.environ['TC_NGROUPS'] = str(ng)
                run(50, 1, (19, ), 10000, 50, 'cfg2 nw=%d ngroups=%d' % (nw, ng))
    elif which == 'cfg5':
        for dtype in ['float64', 'float32']:
            run(100, 1, (19, 40), 10000, 5, 'cfg5 %s' % dtype, dtype=dtype)
        for nw in [4, 8]:
            os.environ['TC_NWAVES'] = str(nw)
            run(100, 1, (19, 40), 10000, 5, 'cfg5 float32 nwaves(arg)=%d' % nw, dtype='float32')
    elif which == 'budget':
        for budget in [56, 80, 104, 128, 200]:
            os.environ['TC_ROW_BUDGET'] = str(budget)
            for nw in [4, 8]:
                os.environ['TC_NWAVES'] = str(nw)
                run(50, 2, (19, ), 10000, 10, 'cfg3 budget=%d nw=%d' % (budget, nw), flags=1)
    elif which == 'auto':
        run(50, 1, (19, ), 10000, 50, 'cfg2 1e4')
        run(50, 2, (19, ), 10000, 20, 'cfg3 1e4 separate', flags=1)
        run(50, 1, (19, ), 1000, 50, 'cfg2 1e3')
        run(50, 1, (19, ), 64, 50, 'cfg2 64')
        run(50, 1, (19, ), 100000, 10, 'cfg2 1e5')
        run(30, 1, (19, ), 10000, 50, 'bolplanck-like G=60')
    elif which == 'cfg3one':
        run(50, 2, (19, ), 10000, 2000, 'cfg3 splits=%s nw=%s ng=%s' % (os.environ.get('TC_OCC_SPLITS'), os.environ.get('TC_NWAVES'), os.environ.get('TC_NGROUPS')), flags=1)
    elif which == 'cfg5one':
        run(100, 1, (19, 40), 10000, 3, 'cfg5 float32', dtype='float32')
    elif which == 'f32':
        for ng in [0, 4, 8, 12, 16]:
            os.environ['TC_NGROUPS'] = str(ng)
            run(100, 1, (19, 40), 10000, 5, 'cfg5 float32 ngroups=%d' % ng, dtype='float32')
    elif which == 'mfma':
        for nw in [4]:
            for ng in [6, 8, 10, 12, 13, 14, 16, 20, 26]:
                os.environ['TC_NWAVES'] = str(nw)
                os.environ['TC_NGROUPS'] = str(ng)
                run(50, 1, (19, ), 10000, 50, 'cfg2 nw=%d ngroups=%d' % (nw, ng))
    elif which == 'waves':
        for nw, ng in [(4, 8), (5, 6), (6, 5), (6, 6), (7, 5), (8, 4), (6, 7), (5, 8)]:
            os.environ['TC_NWAVES'] = str(nw)
            os.environ['TC_NGROUPS'] = str(ng)
            run(50, 1, (19, ), 10000, 50, 'cfg2 nw=%d ngroups=%d' % (nw, ng))
