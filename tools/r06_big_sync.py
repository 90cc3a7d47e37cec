import sys, time, numpy as np
sys.path.insert(0, '.')
from tabcorr_amd import TabCorr, synthetic, _lib
lib = _lib.load()
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
h = halotab.to_device().handle
for n in (100000, 200000, 400000):
    theta = synthetic.zheng07_draws(n, seed=1)
    ngal, xi = np.empty(n), np.empty((n, 19))
    def call():
        _lib.check(lib.tc_predict_zheng07_batch(h, _lib.as_double_p(theta), 5, n, 10, 0, _lib.as_double_p(ngal), _lib.as_double_p(xi)))
    for stagger in (10, 0, 10, 0):
        _lib.check(lib.tc_table_set_option(h, b'sync_stagger', stagger))
        for _ in range(3): call()
        t0 = time.perf_counter()
        for _ in range(20): call()
        print(n, 'sync_stagger', stagger, '%.1f us per call' % ((time.perf_counter() - t0) / 20 * 1e6), flush=True)
