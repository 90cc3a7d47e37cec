#!/usr/bin/env python3
"""The form a batch takes -- three kernels, one launch with 64-draw or 32-draw workgroups --
as chosen by the built-in estimate (launch.hip: fused_eligible: what the first 255 pipelined
calls get) and by the library's DEFAULT behaviour from the 256th pipelined call on (it measures
by itself: option "autotune_after"; nobody calls autotune() here), against the best forced form,
over a grid of table shapes and batch sizes (device-resident pipelined calls, us per call).
Prints every point where a choice is more than 5 % behind the best forced form, and the worst
ratios.  gpurun -- python3 tools/r05_dispatch.py [--quick]"""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench_legs import Device, sustained          # noqa: E402
from tabcorr_amd import TabCorr, synthetic, _lib          # noqa: E402

lib = _lib.load()
_lib.require_device()
dev = Device(lib, _lib)
quick = '--quick' in sys.argv
bins = (40, 100, 200) if quick else (40, 60, 80, 100, 120, 160, 200, 240)
r_values = (3, 19) if quick else (3, 8, 13, 19)
draws = (256, 3000, 10000, 40000) if quick else (256, 1024, 3000, 6000, 10000, 20000, 40000)
theta = synthetic.zheng07_draws(40000, seed=1)
d_theta = dev.upload(theta)
d_ngal, d_xi = dev.malloc(40000), dev.malloc(40000 * 20)


def option(h, name, value):
    _lib.check(lib.tc_table_set_option(h, name, value))


def timed(h, n):
    return sustained(
        lambda: _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, n, 10, 0, d_ngal,
                                                               d_xi)),
        lambda: _lib.check(lib.tc_table_synchronize(h)), seconds=0.08, warm_seconds=0.03) * 1e6


worst = {'formula': (0, None), 'default': (0, None)}
for g in bins:
    for n_r in r_values:
        table = synthetic.synthetic_table(g // 2, 1, (n_r, ), 'auto', seed=g + n_r)
        halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                      table['tpcf_shape'], table['attrs'])
        h = halotab.to_device().handle
        rows = []
        for n in draws:
            forced = {}
            option(h, b'fused_min_draws', 1)
            option(h, b'fused_max_draws', 1 << 30)
            for name, fused, shape in (('three', 0, 0), ('64', 2, 64), ('32', 2, 32)):
                option(h, b'fused', fused)
                option(h, b'fused_draws', shape)
                us = timed(h, n)
                launch = [ctypes.c_int() for _ in range(4)]
                lib.tc_table_last_launch(h, *[ctypes.byref(v) for v in launch])
                ran = ('three' if launch[2].value > 0 else
                       '32' if launch[0].value == (n + 31) // 32 else '64')
                if ran == name:
                    forced[name] = us
            option(h, b'fused', 1)
            option(h, b'fused_draws', 0)
            option(h, b'fused_min_draws', 0)
            option(h, b'fused_max_draws', 30720)
            option(h, b'autotune_after', 0)           # (the estimate alone)
            rows.append([n, forced, timed(h, n), None])
        # the default: the 256th pipelined call measures
        option(h, b'autotune_after', 256)
        assert halotab.autotune(measure=False) is None
        for _ in range(300):
            _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, 3000, 10, 0, d_ngal, d_xi))
        _lib.check(lib.tc_table_synchronize(h))
        assert halotab.autotune(measure=False) is not None
        for row in rows:
            row[3] = timed(h, row[0])
        for n, forced, formula, tuned in rows:
            best = min(forced.values())
            for name, value in (('formula', formula), ('default', tuned)):
                ratio = value / best
                if ratio > worst[name][0]:
                    worst[name] = (ratio, (g, n_r, n))
                if ratio > 1.05:
                    print('G=%3d R=%2d %6d draws: %-8s %7.1f us, best forced %7.1f (%s)  x%.2f' % (
                        g, n_r, n, name, value, best,
                        ' '.join('%s %.1f' % kv for kv in sorted(forced.items())), ratio),
                        flush=True)
        del halotab
print('worst ratio to the best forced form: estimate alone x%.3f at %s, default (measured at the '
      '256th call) x%.3f at %s' % (worst['formula'] + worst['default']))
