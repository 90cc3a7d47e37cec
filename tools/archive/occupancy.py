"""Developer tool: residency of contraction workgroups per CU over several consecutive
launches of a sustained, pipelined sequence (TC_TRACE=<ring>)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
RING = int(os.environ.get('TC_TRACE', '12'))
os.environ['TC_TRACE'] = str(RING)
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
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
rec = rec[rec[:, 0] > 0]
t0 = rec[:, 0].min()
start_t = (rec[:, 0].astype(np.int64) - int(t0)) / 100.0
staged = (rec[:, 1].astype(np.int64) - int(t0)) / 100.0
main = (rec[:, 2].astype(np.int64) - int(t0)) / 100.0
end = (rec[:, 3].astype(np.int64) - int(t0)) / 100.0
hw = rec[:, 4].astype(np.int64)
xcc = rec[:, 5].astype(np.int64) & 0xf
cu = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 10 + ((hw >> 8) & 0xf)
print('blocks recorded', len(rec), 'over %.1f us' % end.max())
# steady window: drop the first and last 15 %
lo, hi = np.percentile(start_t, 15), np.percentile(end, 85)
ts = np.linspace(lo, hi, 400)
resident = np.array([np.sum((start_t <= t) & (end > t)) for t in ts])
in_main = np.array([np.sum((staged <= t) & (main > t)) for t in ts])
print('window %.1f .. %.1f us: resident contraction blocks per CU mean %.2f (min %.2f max %.2f); in main loop %.2f' % (
    lo, hi, resident.mean() / 256, resident.min() / 256, resident.max() / 256, in_main.mean() / 256))
print('block phases (us): stage %.2f  main %.2f  tail %.2f  lifetime %.2f' % (
    np.median(staged - start_t), np.median(main - staged), np.median(end - main), np.median(end - start_t)))
# per CU: gaps between consecutive blocks when fewer than 4 are resident
sel = (start_t >= lo) & (end <= hi)
print('blocks in window per CU per us: %.4f -> one block per CU every %.2f us' % (
    sel.sum() / 256 / (hi - lo), 256 * (hi - lo) / max(sel.sum(), 1)))
