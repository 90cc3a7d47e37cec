"""Developer tool: per-workgroup timeline of the LAST contraction of a sustained,
pipelined sequence of device-pointer predict calls (TC_TRACE=1)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ['TC_TRACE'] = '1'
from tabcorr_amd import TabCorr, synthetic, _lib

lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
dev = halotab.to_device()
n_draws = 10000
theta = synthetic.zheng07_draws(n_draws, seed=1)


def dmalloc(count):
    ptr = ctypes.c_void_p()
    _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    return ptr


d_theta = dmalloc(theta.size)
_lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
d_out = dmalloc(n_draws * 20)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
start = time.perf_counter()
for _ in range(steps):
    _lib.check(lib.tc_predict_zheng07_batch_device(
        dev.handle, d_theta, 5, n_draws, 10, 0, d_out, ctypes.c_void_p(d_out.value + n_draws * 8)))
_lib.check(lib.tc_table_synchronize(dev.handle))
print('steps %d: %.1f us per step' % (steps, (time.perf_counter() - start) / steps * 1e6))
n = ctypes.c_int64()
_lib.check(lib.tc_debug_trace(dev.handle, None, 0, ctypes.byref(n)))
rec = np.zeros((n.value, 6), dtype=np.uint64)
_lib.check(lib.tc_debug_trace(dev.handle, rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value, ctypes.byref(n)))
t0 = rec[:, 0].min()
start_t = (rec[:, 0] - t0) / 100.0
staged = (rec[:, 1] - t0) / 100.0
main = (rec[:, 2] - t0) / 100.0
end = (rec[:, 3] - t0) / 100.0
cycles = (rec[:, 5] >> np.uint64(4)).astype(np.float64)
print('blocks', n.value, 'kernel span %.1f us' % end.max())
print('start: median %.1f 90%% %.1f max %.1f' % (np.median(start_t), np.percentile(start_t, 90), start_t.max()))
print('stage dur: median %.2f max %.2f' % (np.median(staged - start_t), (staged - start_t).max()))
print('main dur: median %.2f min %.2f max %.2f' % (np.median(main - staged), (main - staged).min(), (main - staged).max()))
print('shader clock during main loop: median %.3f GHz (min %.3f max %.3f)' % tuple(
    np.percentile(cycles / ((main - staged) * 1e3), [50, 0, 100])))
print('tail dur: median %.2f max %.2f' % (np.median(end - main), (end - main).max()))
