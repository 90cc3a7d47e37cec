"""Developer measurement: per-wave timeline of contract_quad_kernel (100 MHz stamps):
entry, first operands in, main loop done, sums flushed, end."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
theta = synthetic.zheng07_draws(10000, seed=1)
dev = halotab.to_device()
lib = dev.lib
for _ in range(300):
    halotab.predict_batch(theta)
_lib.check(lib.tc_table_set_option(dev.handle, b'trace', 1))
for rep in range(3):
    halotab.predict_batch(theta)
    nw = ctypes.c_int64()
    _lib.check(lib.tc_debug_wave_trace(dev.handle, None, 0, ctypes.byref(nw)))
    w = np.zeros((nw.value, 6), dtype=np.uint64)
    _lib.check(lib.tc_debug_wave_trace(dev.handle, w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nw.value, ctypes.byref(nw)))
    t = (w[:, :5].astype(np.int64) - int(w[:, 0].min())) / 100.0
    print('waves %d  kernel span (first entry -> last end) %.2f us' % (len(w), t[:, 4].max()))
    for name, col in (('entry', t[:, 0]), ('first operands in', t[:, 1]), ('main loop done', t[:, 2]),
                      ('sums flushed', t[:, 3]), ('end', t[:, 4])):
        print('  %-18s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f' % (
            name, col.min(), np.percentile(col, 10), np.median(col), np.percentile(col, 90), col.max()))
    main = t[:, 2] - t[:, 1]
    clock = w[:, 5].astype(np.float64) / (main * 1e3)
    print('  shader clock during the main loops: median %.3f GHz (p10 %.3f, p90 %.3f)' % (np.median(clock), np.percentile(clock, 10), np.percentile(clock, 90)))
    print('  main loop duration: min %.2f median %.2f max %.2f us' % (main.min(), np.median(main), main.max()))
    print('  prologue (entry -> operands): median %.2f max %.2f;  epilogue (main done -> end): median %.2f max %.2f' % (
        np.median(t[:, 1] - t[:, 0]), (t[:, 1] - t[:, 0]).max(), np.median(t[:, 4] - t[:, 2]), (t[:, 4] - t[:, 2]).max()))
