"""Developer tool: steady-state rate of the contraction kernel alone (large batch, many
scheduling rounds; needs the knob-enabled build and TC_SKIP_OCC=1 TC_SKIP_FINALIZE=1)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
dev = halotab.to_device()
for n_draws in (10000, 100000, 200000):
    theta = synthetic.zheng07_draws(n_draws, seed=1)
    d_theta = ctypes.c_void_p(); d_out = ctypes.c_void_p()
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_theta), theta.nbytes))
    _lib.check(lib.tc_device_malloc(ctypes.byref(d_out), n_draws * 20 * 8))
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
    def step():
        _lib.check(lib.tc_predict_zheng07_batch_device(
            dev.handle, d_theta, 5, n_draws, 10, 0, d_out, ctypes.c_void_p(d_out.value + n_draws * 8)))
    for _ in range(12):
        step()
    _lib.check(lib.tc_table_synchronize(dev.handle))
    steps = max(20, int(0.6e5 * 1e4 / n_draws / 5))      # about 0.6 s of work
    for _ in range(steps // 4):
        step()
    _lib.check(lib.tc_table_synchronize(dev.handle))
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _lib.check(lib.tc_table_synchronize(dev.handle))
    dt = (time.perf_counter() - t0) / steps
    print('%7d draws: %8.1f us per launch = %.1f us per 10^4 draws = %.1f TFLOP/s' % (
        n_draws, dt * 1e6, dt * 1e6 * 1e4 / n_draws, n_draws * 2.0705e5 / dt / 1e12))
    lib.tc_device_free(d_theta); lib.tc_device_free(d_out)
if os.environ.get('TC_TRACE'):
    n = ctypes.c_int64()
    _lib.check(lib.tc_debug_trace(dev.handle, None, 0, ctypes.byref(n)))
    rec = np.zeros((n.value, 6), dtype=np.uint64)
    _lib.check(lib.tc_debug_trace(dev.handle, rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value, ctypes.byref(n)))
    rec = rec[rec[:, 0] > 0]
    staged = rec[:, 1].astype(np.int64); main = rec[:, 2].astype(np.int64)
    cycles = (rec[:, 5] >> np.uint64(4)).astype(np.float64)
    ok = main > staged
    clock = cycles[ok] / ((main[ok] - staged[ok]) * 10.0)
    print('shader clock during the main loops of the last launch: median %.3f GHz (10%% %.3f, 90%% %.3f)' % tuple(np.percentile(clock, [50, 10, 90])))
    print('block phases (us): stage %.2f main %.2f tail %.2f' % (
        np.median(staged - rec[:, 0].astype(np.int64)) / 100.0, np.median(main - staged) / 100.0,
        np.median(rec[:, 3].astype(np.int64) - main) / 100.0))
