#!/usr/bin/env python3
"""Round 6: where a workgroup of the mode-cross one-launch kernel (predict_cross_fused_kernel)
spends its time -- 100 MHz stamps of wave 0 (developer build, TC_FUSED_STAMPS=1), 10^4 draws of
the reference's AbacusSummit fixture (tests/golden/ds_efficient.hdf5): its first table, or the
interpolator over its four tables; the launch alone on the chip (one lane) or pipelined.
    gpurun -- 'bash tools/build_dev.sh && TABCORR_AMD_LIBRARY=build/ab/dev.so TC_FUSED_STAMPS=1 \
               python3 tools/r06_stamps_cross.py [ds1|ds4] [lanes]'"""
import ctypes
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from tabcorr_amd import Interpolator, synthetic, _lib   # noqa: E402

lib = _lib.load()
which = sys.argv[1] if len(sys.argv) > 1 else 'ds1'
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
N, N_R = 10000, 13
rng = np.random.default_rng(0)
theta = synthetic.zheng07_draws(N, seed=1)
theta[:, 0] = rng.uniform(12.5, 13.3, N)
theta[:, 3] = rng.uniform(13.6, 14.4, N)
x = np.ascontiguousarray(np.stack([rng.uniform(xp[0], xp[-1], size=N) for xp in interp.xp], axis=-1))
pointers = [ctypes.c_void_p() for _ in range(4)]
for ptr, count in zip(pointers, (theta.size, x.size, N, N * N_R)):
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
d_theta, d_x, d_ngal, d_xi = pointers
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
_lib.check(lib.tc_memcpy_h2d(d_x, x.ctypes.data_as(ctypes.c_void_p), x.nbytes))
if which == 'ds4':
    idev = interp.to_device()
    h = idev.tables[0].handle

    def call():
        _lib.check(lib.tc_interp_predict_zheng07_batch_device(idev.handle, d_theta, 5, d_x, N, 10, 0,
                                                              d_ngal, d_xi))

    def sync():
        _lib.check(lib.tc_interp_synchronize(idev.handle))
else:
    h = interp.tabcorr_list[0].to_device().handle

    def call():
        _lib.check(lib.tc_predict_zheng07_batch_device(h, d_theta, 5, N, 10, 0, d_ngal, d_xi))

    def sync():
        _lib.check(lib.tc_table_synchronize(h))
if lanes != 4:
    _lib.check(lib.tc_table_set_option(h, b'lanes', lanes))
rows = []
for _ in range(20):
    for _ in range(8):
        call()
    sync()
    n = ctypes.c_int64()
    _lib.check(lib.tc_debug_trace(h, None, 0, ctypes.byref(n)))
    raw = np.zeros(n.value * 6, dtype=np.uint64)
    _lib.check(lib.tc_debug_trace(h, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value,
                                  ctypes.byref(n)))
    n_wg = n.value * 6 // 16
    rows.append(raw[:n_wg * 16].reshape(n_wg, 16).astype(np.int64))
stamps = np.concatenate(rows)
launch = [ctypes.c_int() for _ in range(4)]
lib.tc_table_last_launch(h, *[ctypes.byref(v) for v in launch])
print('%s, %d lane(s): %d workgroups per launch, last launch %s' %
      (which, lanes, len(rows[-1]), tuple(v.value for v in launch)))
if len(rows[-1]) == 0:
    sys.exit('the last launch was not predict_cross_fused_kernel')
names = ['math table + draws set up', 'chunks (occupations, barriers, products)',
         'sums to LDS', 'deferred pairs', 'weights and norms', 'results']
print('%-44s %8s %8s %8s' % ('phase (us)', 'median', 'p10', 'p90'))
for k, name in enumerate(names):
    values = (stamps[:, k + 1] - stamps[:, k]) / 100.0
    print('%-44s %8.2f %8.2f %8.2f' % (name, np.median(values), np.percentile(values, 10),
                                       np.percentile(values, 90)))
for slot, name in ((8, '  of the chunks: wave 0 in the occupations'), (9, '  waiting at the barriers'),
                   (10, '  in the products')):
    values = stamps[:, slot] / 100.0
    print('%-44s %8.2f %8.2f %8.2f' % (name, np.median(values), np.percentile(values, 10),
                                       np.percentile(values, 90)))
print('deferred pairs per workgroup: median %d (largest %d) in %d passes of 64 over its 8 waves' %
      (np.median(stamps[:, 12]), stamps[:, 12].max(), np.median(stamps[:, 13])))
whole = (stamps[:, 6] - stamps[:, 0]) / 100.0
print('%-44s %8.2f %8.2f %8.2f' % ('workgroup', np.median(whole), np.percentile(whole, 10),
                                   np.percentile(whole, 90)))
if os.environ.get('TC_STAMPS_RAW'):
    print(rows[-1][:3])
