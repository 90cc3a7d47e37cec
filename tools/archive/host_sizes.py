"""Developer measurement: host-buffer predict_batch() latency against batch size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
sizes = [int(v) for v in sys.argv[1:]] or [1, 64, 256, 1000, 4000, 10000, 40000]
for n in sizes:
    theta = synthetic.zheng07_draws(n, seed=1)
    for _ in range(8):
        halotab.predict_batch(theta)
    reps = 200 if n <= 4000 else 40
    t0 = time.perf_counter()
    for _ in range(reps):
        halotab.predict_batch(theta)
    dt = (time.perf_counter() - t0) / reps
    print('%6d draws: %8.1f us per call  %.3g calls/s' % (n, dt * 1e6, n / dt))
