#!/usr/bin/env python3
"""Mode cross, a table of up to 16 rows (the reference's AbacusSummit table, 13 r values): the
register form (predict_cross_small_kernel) against the 32-row chunk form with group records
and deferred pairs (launch.hip: choose_cross_fused), device-resident pipelined calls, per batch
size.  On the GPU box: python tools/r05_cross_wide.py  ->  gpurun_out/r05_cross_wide.log"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench_legs                                           # noqa: E402
from tabcorr_amd import Interpolator, _lib, synthetic       # noqa: E402


def main():
    lib = _lib.load()
    dev = bench_legs.Device(lib, _lib)
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    tab = interp.tabcorr_list[0]
    handle = tab.to_device().handle
    rng = np.random.default_rng(0)
    lines = []
    for n in (256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 10000, 20000, 40000):
        theta = synthetic.zheng07_draws(n, seed=1)
        theta[:, 0] = rng.uniform(12.5, 13.3, n)
        theta[:, 3] = rng.uniform(13.6, 14.4, n)
        d_theta, d_ngal, d_xi = dev.upload(theta), dev.malloc(n), dev.malloc(n * 13)
        row = {}
        out = {}
        for name, value in (('registers', 0), ('chunks', 1)):
            _lib.check(lib.tc_table_set_option(handle, b'cross_wide_min_draws', value))
            row[name] = bench_legs.sustained(
                lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                    handle, d_theta, 5, n, 10, 0, d_ngal, d_xi)),
                lambda: _lib.check(lib.tc_table_synchronize(handle)), seconds=0.3) * 1e6
            out[name] = dev.download(d_xi, n * 13)
        _lib.check(lib.tc_table_set_option(handle, b'cross_wide_min_draws', 4096))
        default = bench_legs.sustained(
            lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n, 10, 0, d_ngal, d_xi)),
            lambda: _lib.check(lib.tc_table_synchronize(handle)), seconds=0.3) * 1e6
        differ = float(np.max(np.abs(out['chunks'] - out['registers']) /
                              np.maximum(np.abs(out['registers']), 1e-300)))
        lines.append('%6d draws: registers %7.2f us, chunks %7.2f us, default %7.2f us; forms '
                     'differ by %.1e' % (n, row['registers'], row['chunks'], default, differ))
        print(lines[-1], flush=True)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(REPO, 'gpurun_out', 'r05_cross_wide.log'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
