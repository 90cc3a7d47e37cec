"""Developer check: device / pinned allocations are released with the handles."""
import ctypes, gc, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Interpolator, synthetic

hip = ctypes.CDLL('/opt/rocm/lib/libamdhip64.so')
def free_bytes():
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return free.value

def cycle():
    table = synthetic.synthetic_table(20, 1, (19, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'], table['attrs'])
    theta = synthetic.zheng07_draws(3000, seed=1)
    halotab.predict_batch(theta)
    halotab.predict_batch(theta[:1])
    halotab.predict_batch(theta, separate_gal_type=True)
    halotab.chi2_batch(theta, np.ones(19), np.eye(19))
    halotab.mean_occupation_batch(theta)
    tables, keys, points = synthetic.synthetic_interpolator((4, 4), 10, 1, (19, ), 'auto', seed=7)
    interp = Interpolator([TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs']) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    x = np.stack([np.full(100, 0.5 * (xp[0] + xp[-1])) for xp in interp.xp], axis=-1)
    interp.predict_batch(theta[:100], x)
    del halotab, interp
    gc.collect()

cycle()
before = free_bytes()
for _ in range(30):
    cycle()
after = free_bytes()
print('free before %.1f MB, after 30 cycles %.1f MB, difference %.2f MB' % (before / 2**20, after / 2**20, (before - after) / 2**20))
assert before - after < 64 * 2**20
print('ok')
