"""Developer measurement: host-buffer API (PCIe-inclusive) and per-call latency."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Zheng07Model, synthetic

table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
theta = synthetic.zheng07_draws(10000, seed=1)
halotab.predict_batch(theta)
for n in [1, 10, 100, 1000, 10000, 100000]:
    th = synthetic.zheng07_draws(n, seed=2)
    for _ in range(4):       # buffers grow and pages fault in on the first calls
        halotab.predict_batch(th)
    reps = max(5, min(200, 200000 // n))
    t0 = time.perf_counter()
    for _ in range(reps):
        halotab.predict_batch(th)
    dt = (time.perf_counter() - t0) / reps
    print('predict_batch(%6d draws), host arrays in/out: %9.1f us  -> %.3g calls/s' % (n, dt * 1e6, n / dt))
model = Zheng07Model()
halotab.predict(model)
t0 = time.perf_counter()
for i in range(2000):
    model.param_dict['logMmin'] = 12.0 + 1e-4 * i
    halotab.predict(model)
dt = (time.perf_counter() - t0) / 2000
print('predict(model) scalar API: %.1f us per call -> %.3g calls/s' % (dt * 1e6, 1 / dt))
t0 = time.perf_counter()
for i in range(2000):
    model.param_dict['logMmin'] = 12.0 + 1e-4 * i
    halotab.predict(model, separate_gal_type=True)
dt = (time.perf_counter() - t0) / 2000
print('predict(model, separate_gal_type=True): %.1f us per call' % (dt * 1e6))
