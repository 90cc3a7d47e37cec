"""Developer fuzzing: random table / interpolator shapes, batch sizes and options through
the HIP path against the NumPy oracle (float64: 1e-10; float32: 1e-5 of the scale)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from tabcorr_amd import TabCorr, Interpolator, synthetic
from oracle import tabcorr_oracle as oracle

def make(t, **kw):
    return TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs'], **kw)

def close(a, b, tol, what):
    scale = np.max(np.abs(b)) if np.size(b) else 1.0
    err = np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(scale, 1e-300) if np.size(b) else 0.0
    assert err < tol, (what, err)
    return err

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    n_prim = int(rng.integers(1, 40)); n_sec = int(rng.choice([1, 1, 2, 3]))
    n_r = int(rng.integers(1, 75)); mode = str(rng.choice(['auto', 'cross']))
    shape = (n_r, ) if rng.random() < 0.7 or n_r < 4 else (2, n_r // 2)
    n_draws = int(rng.choice([1, 2, 5, 63, 64, 65, 129, 700]))
    separate = bool(rng.random() < 0.4); modulate = bool(rng.random() < 0.3)
    f32 = bool(rng.random() < 0.25); n_gauss = int(rng.choice([10, 10, 10, 3, 17]))
    tol = 1e-5 if f32 else 1e-10
    kw = dict(compute_dtype='float32') if f32 else {}
    theta = synthetic.zheng07_draws(n_draws, seed=1000 + trial)
    if rng.random() < 0.7:
        table = synthetic.synthetic_table(n_prim, n_sec, shape, mode, seed=trial)
        got = make(table, **kw).predict_batch(theta, separate_gal_type=separate, n_gauss_prim=n_gauss,
                                               modulate_with_cenocc=modulate)
        sel = slice(0, min(n_draws, 6))
        want = oracle.predict_zheng07_batch(table, theta[sel], separate_gal_type=separate,
                                            n_gauss_prim=n_gauss, modulate_with_cenocc=modulate)
        kind = 'table'
    else:
        grid = tuple(int(v) for v in rng.integers(4, 6, size=int(rng.integers(1, 3))))
        tables, keys, points = synthetic.synthetic_interpolator(grid, min(n_prim, 14), 1, shape, mode, seed=trial)
        interp = Interpolator([make(t, **kw) for t in tables], {k: points[:, d] for d, k in enumerate(keys)})
        x = np.stack([rng.uniform(xp[0], xp[-1], size=n_draws) for xp in interp.xp], axis=-1)
        got = interp.predict_batch(theta, x, separate_gal_type=separate, n_gauss_prim=n_gauss,
                                   modulate_with_cenocc=modulate)
        sel = slice(0, min(n_draws, 3))
        setup = oracle.interpolator_setup(tables, points)
        want = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[sel], x[sel], separate_gal_type=separate,
                                                         n_gauss_prim=n_gauss, modulate_with_cenocc=modulate)
        kind = 'interp%s' % (grid, )
    if separate:
        for key in want[1]:
            close(got[1][key][sel], want[1][key], tol, (trial, kind, key))
            close(got[0][key if key in got[0] else list(got[0])[0]][sel], want[0][key if key in want[0] else list(want[0])[0]], 1e-10, (trial, 'ngal'))
    else:
        e = close(got[1][sel], want[1], tol, (trial, kind, 'xi'))
        if not f32 and e > worst:
            worst = e
            print('trial %d: %s G=%d R=%s mode=%s draws=%d n_gauss=%d modulate=%s: %.3g' % (
                trial, kind, 2 * n_prim * n_sec, shape, mode, n_draws, n_gauss, modulate, e))
        close(got[0][sel], want[0], 1e-10, (trial, kind, 'ngal'))
print('all trials passed; worst float64 deviation %.3g' % worst)
