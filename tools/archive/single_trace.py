"""Developer measurement: phase timeline of single_draw_kernel (100 MHz stamps per block)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Zheng07Model, synthetic, _lib

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
model = Zheng07Model()
for _ in range(200):
    halotab.predict(model)
dev = halotab.to_device()
lib = dev.lib
_lib.check(lib.tc_table_set_option(dev.handle, b'trace', 1))
rows = []
for i in range(50):
    halotab.predict(model)
    out = np.zeros(64 * 8 + 8, dtype=np.uint64)
    n = ctypes.c_int64()
    _lib.check(lib.tc_debug_trace(dev.handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 64 * 8 // 6, ctypes.byref(n)))
    blocks = min(n.value, (min(n.value, 64 * 8 // 6) * 6) // 8)
    s = out[:blocks * 8].reshape(blocks, 8).astype(np.int64)
    rows.append(s)
s = rows[-1]
t0 = s[:, 0].min()
print('blocks', len(s))
print('phase: 0 start, 1 tables staged, 2 nodes done, 3 contraction done, 4 partial sums written (host memory)')
for b in range(len(s)):
    print(b, ' '.join('%6.2f' % ((v - t0) / 100.0) if v else '   -  ' for v in s[b, :5]))
last = [r[:, 4].max() - r[:, 0].min() for r in rows[5:]]
print('kernel span (first start -> last end): mean %.2f us' % (np.mean(last) / 100.0))
for ph in range(1, 5):
    print('phase %d -> mean over blocks of (stamp - own start): %.2f us' % (ph, np.mean([np.mean(r[:, ph] - r[:, 0]) for r in rows[5:]]) / 100.0))
