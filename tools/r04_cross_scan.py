import ctypes, os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from bench import Device, sustained
from tabcorr_amd import Interpolator, synthetic, _lib
lib = _lib.load(); dev = Device(lib, _lib)
interp = Interpolator.read('/root/repo/tests/golden/ds_efficient.hdf5')
device = interp.to_device(); h = device.handle; h0 = device.tables[0].handle
rng = np.random.default_rng(0)
theta = synthetic.zheng07_draws(40000, seed=1)
theta[:, 0] = rng.uniform(12.5, 13.3, len(theta)); theta[:, 3] = rng.uniform(13.6, 14.4, len(theta))
x = np.ascontiguousarray(np.stack([rng.uniform(xp[0], xp[-1], size=len(theta)) for xp in interp.xp], axis=-1))
d_theta, d_x = dev.upload(theta), dev.upload(x)
d_ngal, d_xi = dev.malloc(len(theta)), dev.malloc(13 * len(theta))
def last(hh):
    v = [ctypes.c_int() for _ in range(4)]
    lib.tc_table_last_launch(hh, *[ctypes.byref(q) for q in v]); return tuple(q.value for q in v)
_lib.check(lib.tc_table_set_option(h0, b'cross_min_draws', int(os.environ.get('CROSS_MIN', 6144))))
for n in (256, 1024, 2048, 4096, 6144, 8192, 10000, 16384, 32768):
    row = []
    for fused in (1, 0):
        _lib.check(lib.tc_table_set_option(h0, b'fused', fused))
        s = sustained(lambda: _lib.check(lib.tc_interp_predict_zheng07_batch_device(h, d_theta, 5, d_x, n, 10, 0, d_ngal, d_xi)),
                      lambda: _lib.check(lib.tc_interp_synchronize(h)), seconds=0.2)
        row.append((s * 1e6, last(h0)))
    print('K=4 %6d draws: fused %8.1f us %s   three kernels %8.1f us %s' % (n, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
t = interp.tabcorr_list[0]; ht = t.to_device().handle
for n in (256, 512, 1024, 2048, 4096, 6144, 8192, 10000, 16384, 32768):
    row = []
    for fused in (2, 0):
        _lib.check(lib.tc_table_set_option(ht, b'fused', fused))
        _lib.check(lib.tc_table_set_option(ht, b'fused_min_draws', 1))
        s = sustained(lambda: _lib.check(lib.tc_predict_zheng07_batch_device(ht, d_theta, 5, n, 10, 0, d_ngal, d_xi)),
                      lambda: _lib.check(lib.tc_table_synchronize(ht)), seconds=0.2)
        row.append((s * 1e6, last(ht)))
    print('one %6d draws: fused %8.1f us %s   three kernels %8.1f us %s' % (n, row[0][0], row[0][1], row[1][0], row[1][1]), flush=True)
