#!/usr/bin/env python3
"""Round 6: where a workgroup of the one-launch kernel spends its time -- 100 MHz stamps of wave 0
at the phase boundaries and of waves 0 and 4 inside the occupation phase (developer build,
TC_FUSED_STAMPS=1; a stamp costs 0.1 - 0.3 us), the launch alone on the chip, 10^4 draws of the
benchmark's table.
    gpurun -- 'bash tools/build_dev.sh && TABCORR_AMD_LIBRARY=build/ab/dev.so TC_FUSED_STAMPS=1 \
               python3 tools/r06_stamps.py [draws per workgroup] [draws]'"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tabcorr_amd import TabCorr, synthetic, _lib   # noqa: E402

lib = _lib.load()
draws = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
h = halotab.to_device().handle
theta = synthetic.zheng07_draws(N, seed=1)
pointers = [ctypes.c_void_p() for _ in range(3)]
for ptr, count in zip(pointers, (theta.size, N, N * 19)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
d_theta, d_ngal, d_xi = pointers
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
for name, value in (('fused', 2), ('fused_draws', draws), ('lanes', 1)):
    _lib.check(lib.tc_table_set_option(h, name.encode(), value))
names = ['math table staged', 'draws set up', 'bins + deferred pairs', 'sums exchanged',
         'matrix phase', 'wait for the other waves', "the wave's part to LDS", 'wait (parts)',
         'parts added, normalised', 'results written']
rows = []
per_wave = []      # (runs, workgroups, waves, {bins done, deferred pairs done}) after 'draws set up'
for _ in range(20):
    for _ in range(5):
        _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, N, 10, 0, d_ngal, d_xi))
    _lib.check(lib.tc_table_synchronize(h))
    n = ctypes.c_int64()
    _lib.check(lib.tc_debug_trace(h, None, 0, ctypes.byref(n)))
    raw = np.zeros(n.value * 6, dtype=np.uint64)
    _lib.check(lib.tc_debug_trace(h, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value,
                                  ctypes.byref(n)))
    n_wg = (N + draws - 1) // draws
    stamps = raw[:n_wg * 32].reshape(n_wg, 32).astype(np.int64)
    per_wave.append(stamps[:, 16:32] - stamps[:, 2:3])
    # (in order of time: 0 .. 5, 8 all waves through the matrix phase, 9 the wave's part stored,
    # 10 every part stored, 6 parts added and normalised, 7 results written)
    rows.append(stamps[:, [0, 1, 2, 3, 4, 5, 8, 9, 10, 6, 7]])
stamps = rows[-1]
first = stamps[:, 0].min()
print('%d workgroups of %d draws; entry of the workgroups after the first: median %.2f us, last '
      '%.2f us; end of the last workgroup %.2f us' %
      (len(stamps), draws, np.median(stamps[:, 0] - first) / 100, (stamps[:, 0].max() - first) / 100,
       (stamps[:, -1].max() - first) / 100))
phase = np.stack([np.diff(s, axis=1) for s in rows])        # (runs, workgroups, phases)
print('%-24s %8s %8s %8s' % ('phase (us)', 'median', 'p10', 'p90'))
for k, name in enumerate(names):
    values = phase[:, :, k].ravel() / 100.0
    print('%-24s %8.2f %8.2f %8.2f' % (name, np.median(values), np.percentile(values, 10),
                                       np.percentile(values, 90)))
if len(stamps) > 256:
    # (more workgroups than CUs: those of the second round run code the first round has brought
    # into the instruction caches)
    late = np.stack([(s[:, 0] - s[:, 0].min()) > 2000 for s in rows])
    print('workgroups that start 20 us and more after the first (%d of %d):' %
          (late[-1].sum(), len(stamps)))
    for k, name in enumerate(names):
        values = phase[:, :, k][late] / 100.0
        print('%-24s %8.2f %8.2f %8.2f' % (name, np.median(values), np.percentile(values, 10),
                                           np.percentile(values, 90)))
whole = np.stack([s[:, -1] - s[:, 0] for s in rows]).ravel() / 100.0
print('%-24s %8.2f %8.2f %8.2f' % ('workgroup', np.median(whole), np.percentile(whole, 10),
                                   np.percentile(whole, 90)))
waves = np.stack(per_wave) / 100.0          # (runs, workgroups, 8 slots x 2 waves)
waves = waves.reshape(waves.shape[0], waves.shape[1], 2, 8)
print('waves 0 and 4, us after the draws were set up (median): bins done, pairs counted, list '
      'built, node loops done, sums added')
for w in range(2):
    m = np.median(waves[:, :, w, :], axis=(0, 1))
    print('wave %d: %7.2f %7.2f %7.2f %7.2f %7.2f' % (4 * w, m[0], m[2], m[3], m[4], m[1]))
