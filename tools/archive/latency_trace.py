"""Developer: kernel timeline of un-batched predict() calls (run under rocprofv3 --kernel-trace)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, synthetic
table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
theta = synthetic.zheng07_draws(1, seed=1)
for _ in range(300):
    halotab.predict_batch(theta)
