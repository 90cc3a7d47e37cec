"""Developer measurement: shader clock and per-wave main-loop time of contract_quad_kernel
while the four-lane pipeline is running (stamps of the LAST launch of a long burst), against
the same kernel with the launches serialised."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
n_draws = 10000
theta = _lib.contiguous(synthetic.zheng07_draws(n_draws, seed=1))
dev = halotab.to_device()
lib = dev.lib
d = ctypes.c_void_p()
_lib.check(lib.tc_device_malloc(ctypes.byref(d), theta.nbytes + 8 * n_draws * 20 * 8))
_lib.check(lib.tc_memcpy_h2d(d, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))


def out(slot):
    base = d.value + theta.nbytes + slot * n_draws * 20 * 8
    return ctypes.c_void_p(base), ctypes.c_void_p(base + n_draws * 8)


def burst(n):
    for k in range(n):
        ngal, xi = out(k % 8)
        _lib.check(lib.tc_predict_zheng07_batch_device(dev.handle, d, 5, n_draws, 10, 0, ngal, xi))


def report(label):
    nw = ctypes.c_int64()
    _lib.check(lib.tc_debug_wave_trace(dev.handle, None, 0, ctypes.byref(nw)))
    w = np.zeros((nw.value, 6), dtype=np.uint64)
    _lib.check(lib.tc_debug_wave_trace(
        dev.handle, w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nw.value, ctypes.byref(nw)))
    t = (w[:, :5].astype(np.int64) - int(w[:, 0].min())) / 100.0
    main = t[:, 2] - t[:, 1]
    clock = w[:, 5].astype(np.float64) / (main * 1e3)
    print('%-12s span %.2f us  main loop min %.2f median %.2f max %.2f us  clock median %.3f GHz '
          '(p10 %.3f p90 %.3f)  cycles/wave median %.0f' % (
              label, t[:, 4].max(), main.min(), np.median(main), main.max(), np.median(clock),
              np.percentile(clock, 10), np.percentile(clock, 90), np.median(w[:, 5].astype(float))))


for rep in range(3):
    for pipeline in (1, 0):
        _lib.check(lib.tc_table_set_option(dev.handle, b'pipeline', pipeline))
        _lib.check(lib.tc_table_set_option(dev.handle, b'trace', 0))
        for _ in range(8):
            burst(256)
            _lib.check(lib.tc_table_synchronize(dev.handle))
        burst(64)
        _lib.check(lib.tc_table_set_option(dev.handle, b'trace', 1))
        burst(1)                      # the stamped launch sits in the middle of the burst
        _lib.check(lib.tc_table_set_option(dev.handle, b'trace', 0))
        burst(15)
        _lib.check(lib.tc_table_synchronize(dev.handle))
        report('pipelined' if pipeline else 'serialised')
