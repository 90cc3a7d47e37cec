import ctypes, os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from bench import Device, sustained
from tabcorr_amd import Interpolator, synthetic, _lib
lib = _lib.load(); dev = Device(lib, _lib)
interp = Interpolator.read('/root/repo/tests/golden/ds_efficient.hdf5')
make_interp = '--interp' in sys.argv
if make_interp:
    device = interp.to_device()
rng = np.random.default_rng(0)
theta = synthetic.zheng07_draws(40000, seed=1)
theta[:, 0] = rng.uniform(12.5, 13.3, len(theta)); theta[:, 3] = rng.uniform(13.6, 14.4, len(theta))
d_theta = dev.upload(theta)
d_ngal, d_xi = dev.malloc(len(theta)), dev.malloc(13 * len(theta))
dummies = []
if '--dummy-before' in sys.argv:
    from tabcorr_amd import TabCorr
    for k in range(int(sys.argv[sys.argv.index('--dummy-before') + 1])):
        tb = synthetic.synthetic_table(3, 1, (2, ), 'auto', seed=k)
        d = TabCorr.from_arrays(tb['gal_type'], tb['tpcf_matrix'], tb['tpcf_shape'], tb['attrs'])
        d.to_device(); dummies.append(d)
t = interp.tabcorr_list[0]; ht = t.to_device().handle
if '--dummy-after' in sys.argv:
    from tabcorr_amd import TabCorr
    for k in range(int(sys.argv[sys.argv.index('--dummy-after') + 1])):
        tb = synthetic.synthetic_table(3, 1, (2, ), 'auto', seed=k)
        d = TabCorr.from_arrays(tb['gal_type'], tb['tpcf_matrix'], tb['tpcf_shape'], tb['attrs'])
        d.to_device(); dummies.append(d)
def last(hh):
    v = [ctypes.c_int() for _ in range(4)]
    lib.tc_table_last_launch(hh, *[ctypes.byref(q) for q in v]); return tuple(q.value for q in v)
_lib.check(lib.tc_table_set_option(ht, b'fused', 2))
_lib.check(lib.tc_table_set_option(ht, b'fused_min_draws', 1))
for n in (256, 4096, 10000):
    s = sustained(lambda: _lib.check(lib.tc_predict_zheng07_batch_device(ht, d_theta, 5, n, 10, 0, d_ngal, d_xi)),
                  lambda: _lib.check(lib.tc_table_synchronize(ht)), seconds=0.2)
    print('interp' if make_interp else 'alone ', n, '%.1f us' % (s * 1e6), last(ht), flush=True)
