"""Developer measurement: serialised contraction kernel time against the batch size around
10^4 draws -- batches whose draw tiles divide evenly among the 2048 waves (no wave changes
tile in the middle of its share) against those that do not."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic, _lib

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                              table['attrs'])
dev = halotab.to_device()
lib = dev.lib
_lib.check(lib.tc_table_set_option(dev.handle, b'pipeline', 0))
n_max = 20480
theta = _lib.contiguous(synthetic.zheng07_draws(n_max, seed=1))
d = ctypes.c_void_p()
_lib.check(lib.tc_device_malloc(ctypes.byref(d), theta.nbytes + n_max * 20 * 8))
_lib.check(lib.tc_memcpy_h2d(d, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))
ngal = ctypes.c_void_p(d.value + theta.nbytes)
xi = ctypes.c_void_p(d.value + theta.nbytes + n_max * 8)
for n_draws in (8192, 9984, 10000, 10016, 10240, 10400, 12288, 16384, 20480, 8192, 10000):
    for _ in range(3000):
        _lib.check(lib.tc_predict_zheng07_batch_device(dev.handle, d, 5, n_draws, 10, 0, ngal, xi))
    _lib.check(lib.tc_table_synchronize(dev.handle))
    _lib.check(lib.tc_table_timer_begin(dev.handle, 1))
    for _ in range(1000):
        _lib.check(lib.tc_predict_zheng07_batch_device(dev.handle, d, 5, n_draws, 10, 0, ngal, xi))
    ms = ctypes.c_float()
    _lib.check(lib.tc_table_timer_end(dev.handle, ctypes.byref(ms)))
    count, kernel_ms = ctypes.c_int(), ctypes.c_float()
    _lib.check(lib.tc_table_kernel_time(dev.handle, ctypes.byref(count), ctypes.byref(kernel_ms)))
    tiles = (n_draws + 31) // 32
    flop = n_draws * (2 * 19 * 5050 + 3 * 5050)
    print('%6d draws  %4d tiles  %.3f waves per tile  kernel %.2f us  %.2f us per 10^4 draws  frac %.3f' % (
        n_draws, tiles, 2048 / tiles, kernel_ms.value * 1e3, kernel_ms.value * 1e3 * 1e4 / n_draws,
        flop / (kernel_ms.value * 1e-3) / 78.6e12))
