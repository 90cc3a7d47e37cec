"""Developer tool: per-workgroup timeline of the contraction kernel (TC_TRACE=1)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
os.environ['TC_TRACE'] = '1'
from tabcorr_amd import TabCorr, synthetic, _lib

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
theta = synthetic.zheng07_draws(10000, seed=1)
for _ in range(3):
    halotab.predict_batch(theta)
dev = halotab.to_device()
n = ctypes.c_int64()
_lib.check(dev.lib.tc_debug_trace(dev.handle, None, 0, ctypes.byref(n)))
rec = np.zeros((n.value, 6), dtype=np.uint64)
_lib.check(dev.lib.tc_debug_trace(dev.handle, rec.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n.value, ctypes.byref(n)))
t0 = rec[:, 0].min()
start = (rec[:, 0] - t0) / 100.0   # us
staged = (rec[:, 1] - t0) / 100.0
main = (rec[:, 2] - t0) / 100.0
end = (rec[:, 3] - t0) / 100.0
hw = rec[:, 4].astype(np.int64)
xcc = rec[:, 5].astype(np.int64) & 0xf
cycles = (rec[:, 5] >> np.uint64(4)).astype(np.float64)
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
unit = xcc * 1000 + se * 100 + sh * 10 + cu
print('blocks', n.value, 'kernel span %.1f us' % end.max())
print('start: median %.1f  90%% %.1f max %.1f' % (np.median(start), np.percentile(start, 90), start.max()))
print('stage dur: median %.2f max %.2f' % (np.median(staged - start), (staged - start).max()))
print('main dur: median %.2f  min %.2f max %.2f' % (np.median(main - staged), (main - staged).min(), (main - staged).max()))
print('shader clock during main loop: median %.3f GHz (min %.3f max %.3f)' % tuple(np.percentile(cycles / ((main - staged) * 1e3), [50, 0, 100])))
print('tail dur: median %.2f max %.2f' % (np.median(end - main), (end - main).max()))
units, counts = np.unique(unit, return_counts=True)
print('distinct CUs', len(units), 'blocks per CU: min %d median %d max %d' % (counts.min(), np.median(counts), counts.max()))
print('xcc histogram', np.bincount(xcc))
# concurrency over time
ts = np.linspace(0, end.max(), 30)
print('t(us): resident blocks')
for t in ts:
    print('  %6.1f %5d' % (t, np.sum((start <= t) & (end > t))))
# per CU busy span
last = np.array([end[unit == u].max() for u in units])
print('per-CU last end: min %.1f median %.1f max %.1f' % (last.min(), np.median(last), last.max()))

# ---- per-wave progress -------------------------------------------------------------
nw = ctypes.c_int64()
_lib.check(dev.lib.tc_debug_wave_trace(dev.handle, None, 0, ctypes.byref(nw)))
w = np.zeros((nw.value, 6), dtype=np.uint64)
_lib.check(dev.lib.tc_debug_wave_trace(dev.handle, w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nw.value, ctypes.byref(nw)))
w = w[w[:, 0] > 0]
ts = (w[:, :5].astype(np.int64) - int(t0)) / 100.0
hwid = w[:, 5].astype(np.int64)
simd = (hwid >> 4) & 0x3
print('waves traced', len(w))
q = np.diff(ts, axis=1)      # duration of each quarter of the main loop
for name, col in zip(['q1', 'q2', 'q3', 'q4'], q.T):
    print('quarter %s: median %.2f us  10%% %.2f  90%% %.2f' % (name, np.median(col), np.percentile(col, 10), np.percentile(col, 90)))
order = np.argsort(ts[:, 4])
for frac in [0.1, 0.3, 0.5, 0.7, 0.9, 1.0]:
    k = order[int(frac * (len(order) - 1))]
    print('wave finishing at %5.1f us: start %.1f quarters %s' % (ts[k, 4], ts[k, 0], np.round(q[k], 1)))

# ---- wave placement: how many waves of the launch share each SIMD ---------------------
w_all = np.zeros((nw.value, 6), dtype=np.uint64)
_lib.check(dev.lib.tc_debug_wave_trace(dev.handle, w_all.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nw.value, ctypes.byref(nw)))
waves_per_block = nw.value // n.value
valid = w_all[:, 0] > 0
block_of = np.arange(nw.value) // waves_per_block
hw_w = w_all[:, 5].astype(np.int64)
key = (xcc[block_of] * 100000 + ((hw_w >> 13) & 7) * 10000 + ((hw_w >> 12) & 1) * 1000 +
       ((hw_w >> 8) & 0xf) * 10 + ((hw_w >> 4) & 3))[valid]
simds, inverse, per_simd = np.unique(key, return_inverse=True, return_counts=True)
print('SIMDs used', len(simds), 'waves per SIMD histogram', np.bincount(per_simd))
finish = (w_all[valid, 4].astype(np.int64) - int(t0)) / 100.0
for count in np.unique(per_simd):
    sel = per_simd[inverse] == count
    print('  waves on SIMDs holding %d: finish median %.1f us (n=%d)' % (count, np.median(finish[sel]), sel.sum()))
