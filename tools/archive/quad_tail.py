"""Developer measurement: which waves of contract_quad_kernel finish last?  Per-wave stamps
(tc_table_set_option "trace") grouped by XCD, by whether the wave's share crosses a draw-tile
boundary (two runs) and by winner / loser of its SIMD."""
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
n_units, n_tiles, n_waves = 325, 313, 2048
total = n_units * n_tiles
for rep in range(3):
    halotab.predict_batch(theta)
    nw = ctypes.c_int64()
    _lib.check(lib.tc_debug_wave_trace(dev.handle, None, 0, ctypes.byref(nw)))
    w = np.zeros((nw.value, 6), dtype=np.uint64)
    _lib.check(lib.tc_debug_wave_trace(dev.handle, w.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), nw.value, ctypes.byref(nw)))
    t = (w[:, :5].astype(np.int64) - int(w[:, 0].min())) / 100.0
    wave = np.arange(n_waves)
    block = wave // 4
    xcd = block % 8
    local = (block // 8) * 4 + wave % 4
    share = xcd * (n_waves // 8) + local
    begin = share * total // n_waves
    end = (share + 1) * total // n_waves
    crosses = (begin // n_units) != ((end - 1) // n_units)
    done = t[:, 2]
    loser = done > 25.0
    print('rep %d: span %.2f; losers %d; main done of losers: median %.2f p90 %.2f max %.2f' % (
        rep, t[:, 4].max(), loser.sum(), np.median(done[loser]), np.percentile(done[loser], 90), done[loser].max()))
    for flag in (False, True):
        sel = loser & (crosses == flag)
        print('   share crosses a tile boundary %-5s: %4d waves, main done median %.2f p90 %.2f max %.2f; main loop median %.2f' % (
            flag, sel.sum(), np.median(done[sel]), np.percentile(done[sel], 90), done[sel].max(),
            np.median((t[:, 2] - t[:, 1])[sel])))
    print('   by XCD (losers): ' + '  '.join('%d: %.2f/%.2f' % (x, np.median(done[loser & (xcd == x)]), done[loser & (xcd == x)].max()) for x in range(8)))
    late = np.argsort(done)[-12:]
    print('   latest waves: ' + ' '.join('w%d(x%d,%s,%.1f)' % (i, xcd[i], 'X' if crosses[i] else '-', done[i]) for i in late))
    winners = ~loser
    print('   winners: %d, main done median %.2f max %.2f; crossing winners median %.2f' % (
        winners.sum(), np.median(done[winners]), done[winners].max(), np.median(done[winners & crosses]) if (winners & crosses).any() else 0))
