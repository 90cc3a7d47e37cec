"""predict_fused_kernel (one launch per batch: occupation -> quadratic form -> results inside a
workgroup) against the golden vectors, the oracle and the three-kernel path, through every
entry point that may take it.  Needs an MI355X."""

import numpy as np
import pytest

from util import load_golden, table_from_golden, assert_rel

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def make_tabcorr(table, **kwargs):
    from tabcorr_amd import TabCorr
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                               table['tpcf_shape'], table['attrs'], **kwargs)


def set_option(halotab, name, value):
    from tabcorr_amd import _lib
    lib = _lib.load()
    _lib.check(lib.tc_table_set_option(halotab.to_device().handle, name.encode(), value))


def force_fused(halotab, on=True):
    set_option(halotab, 'fused', 2 if on else 0)
    set_option(halotab, 'fused_min_draws', 1)
    set_option(halotab, 'single_draw', 0)      # (small batches: the batched path, not the
                                               # one-launch path of un-batched calls)


def last_launch(halotab):
    """(workgroups, waves per workgroup, slabs, LDS bytes) of the last contraction launch."""
    import ctypes
    from tabcorr_amd import _lib
    lib = _lib.load()
    values = [ctypes.c_int() for _ in range(4)]
    _lib.check(lib.tc_table_last_launch(halotab.to_device().handle, *[ctypes.byref(v) for v in values]))
    return tuple(v.value for v in values)


def fused_ran(halotab, waves=(8, 16)):
    """Was the last launch predict_fused_kernel (8 waves per workgroup -- of 64 draws, or of 32
    for batches below 8192 draws and for tables of 105-208 bins --, 16 when forced for larger
    tables; no slabs of partial sums)?"""
    launch = last_launch(halotab)
    return launch[1] in waves and launch[2] == 0


def test_fused_matches_golden():
    data = load_golden('synthetic_cfg2')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    ngal, xi = halotab.predict_batch(data['theta'])
    workgroups, waves, slabs, lds = last_launch(halotab)
    assert waves == 8 and slabs == 0 and workgroups == (len(data['theta']) + 31) // 32
    assert_rel(ngal, data['ngal'], RTOL, 'ngal')
    assert_rel(xi, data['xi'], RTOL, 'xi')
    # the three-kernel path gives the same to rounding
    force_fused(halotab, False)
    ngal3, xi3 = halotab.predict_batch(data['theta'])
    assert last_launch(halotab)[2] > 0
    assert_rel(ngal, ngal3, 1e-13)
    assert_rel(xi, xi3, 1e-12)


@pytest.mark.parametrize('n_prim, n_sec, n_r, n_gauss, n_draws', [
    (7, 1, 1, 10, 1),          # 14 bins (padded to 16 rows), one r value, one draw
    (5, 2, 3, 10, 63),         # 20 bins
    (13, 1, 4, 10, 64),        # 26 bins -> 28 rows
    (10, 1, 5, 7, 65),         # any n_gauss, two r sub-tiles
    (25, 1, 9, 10, 200),
    (12, 2, 13, 3, 129),
    (50, 1, 19, 10, 1000),     # BASELINE configs[1]'s table
    (33, 1, 20, 10, 333),
    (60, 1, 17, 10, 150),      # 120 bins: 88 KB of LDS, one workgroup per CU
    (3, 1, 2, 100, 40),
])
def test_fused_matches_oracle(n_prim, n_sec, n_r, n_gauss, n_draws):
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim + n_r)
    theta = synthetic.zheng07_draws(n_draws, seed=n_draws)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    ngal, xi = halotab.predict_batch(theta, n_gauss_prim=n_gauss)
    assert fused_ran(halotab), 'the fused kernel did not run'
    expect = oracle.predict_zheng07_batch(table, theta, n_gauss_prim=n_gauss)
    assert_rel(ngal, expect[0], RTOL, 'ngal')
    assert_rel(xi, expect[1], RTOL, 'xi')


def test_fused_likelihood_and_async():
    from tabcorr_amd import pinned_array, pinned_empty, synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = make_tabcorr(table)
    set_option(halotab, 'fused_min_draws', 1)       # fused = 1: pipelined calls only
    set_option(halotab, 'single_draw', 0)
    theta = pinned_array(synthetic.zheng07_draws(777, seed=3))
    expect = oracle.predict_zheng07_batch(table, theta)
    ngal, xi = pinned_empty(777), pinned_empty((777, 19))
    got = halotab.predict_batch_async(theta, out=(ngal, xi)).wait()
    assert fused_ran(halotab), 'the fused kernel did not run'
    assert_rel(got[0], expect[0], RTOL)
    assert_rel(got[1], expect[1], RTOL)
    rng = np.random.default_rng(0)
    vector = expect[1][0] * 1.1
    a = rng.normal(size=(19, 19))
    precision = a @ a.T / np.mean(vector)**2
    want = np.einsum('bi,ij,bj->b', expect[1] - vector, precision, expect[1] - vector)
    n_chi, chi2 = halotab.chi2_batch_async(theta, vector, precision).wait()
    assert fused_ran(halotab)
    assert_rel(n_chi, expect[0], RTOL)
    assert_rel(chi2, want, 1e-9)
    # the synchronous calls run alone on their lane: three kernels unless forced
    sync = halotab.chi2_batch(theta, vector, precision)
    assert last_launch(halotab)[2] > 0
    assert_rel(sync[1], chi2, 1e-11)
    force_fused(halotab)
    forced = halotab.chi2_batch(theta, vector, precision)
    assert fused_ran(halotab)
    assert_rel(forced[1], chi2, 1e-13)
    # many calls in flight, different sizes
    sizes = [64, 1, 300, 65, 2048, 127]
    thetas = [pinned_array(synthetic.zheng07_draws(size, seed=70 + i))
              for i, size in enumerate(sizes)]
    outs = [(pinned_empty(size), pinned_empty((size, 19))) for size in sizes]
    pending = [halotab.predict_batch_async(t, out=o) for t, o in zip(thetas, outs)]
    for index in reversed(range(len(sizes))):
        got = pending[index].wait()
        want = oracle.predict_zheng07_batch(table, thetas[index][:5])
        assert_rel(got[0][:5], want[0], RTOL)
        assert_rel(got[1][:5], want[1], RTOL)
        full = oracle.predict_zheng07_batch(table, thetas[index][-3:])
        assert_rel(got[1][-3:], full[1], RTOL)


def test_fused_degenerate_parameters():
    """NaN / infinite / tied parameters: the same values (NaN for NaN) as the three-kernel path,
    which test_gpu_full_size.py compares with the oracle."""
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(20, 1, (6, ), 'auto', seed=2)
    theta = synthetic.zheng07_draws(200, seed=5)
    theta[3, 0] = np.nan
    theta[10, 1] = 0.0
    theta[11, 1] = np.inf
    theta[20, 2] = np.nan
    theta[30, 3] = np.nan
    theta[40, 4] = np.nan
    theta[50, 3] = -np.inf
    theta[60, 2] = np.inf
    theta[70, 0] = np.inf
    theta[80, 0] = -np.inf
    theta[90, 4] = 0.0
    theta[100, 4] = -1.0
    theta[110, 3] = -400.0
    theta[120] = [11.0, 0.0, 11.0, 13.0, 1.0]
    halotab = make_tabcorr(table)
    force_fused(halotab, False)
    with np.errstate(all='ignore'):
        ngal3, xi3 = halotab.predict_batch(theta)
        force_fused(halotab)
        ngal, xi = halotab.predict_batch(theta)
    assert fused_ran(halotab)
    assert np.array_equal(np.isnan(ngal), np.isnan(ngal3))
    assert np.array_equal(np.isnan(xi), np.isnan(xi3))
    assert np.array_equal(np.isinf(xi), np.isinf(xi3))
    good = np.isfinite(xi3)
    assert_rel(xi[good], xi3[good], 1e-12)
    good = np.isfinite(ngal3)
    assert_rel(ngal[good], ngal3[good], 1e-13)


def test_fused_is_not_taken_where_it_does_not_apply():
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(10, 2, (25, ), 'auto', seed=1)    # two r tiles
    theta = synthetic.zheng07_draws(100, seed=1)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    halotab.predict_batch(theta)
    assert last_launch(halotab)[2] > 0
    table = synthetic.synthetic_table(10, 2, (5, ), 'auto', seed=1)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    halotab.predict_batch(theta)
    assert fused_ran(halotab)
    # (Heaviside assembly bias with another n_gauss_prim than the reference's default keeps the
    # three kernels)
    decorated = np.hstack([theta, np.full((100, 2), 0.3)])
    halotab.predict_batch(decorated, assembias=True, n_gauss_prim=7)
    assert last_launch(halotab)[2] > 0


@pytest.mark.parametrize('draws', [64, 32])
@pytest.mark.parametrize('name', ['leauthaud11_bolplanck_wp', 'leauthaud11_synthetic'])
def test_fused_leauthaud11(name, draws):
    """The Leauthaud11 family (Newton inverse of the stellar-to-halo mass relation per central
    node) in the one-launch form: fixtures recorded from the reference with a duck model,
    with and without modulate_with_cenocc; more draws than a tile (jittered fixture draws)
    against the oracle, total and separated by galaxy type, any n_gauss_prim; the fused
    likelihood; NaN parameters."""
    from oracle import tabcorr_oracle as oracle
    data = load_golden(name)
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    set_option(halotab, 'fused_draws', draws)     # (workgroups of 64 draws, or of one 32-draw tile)
    theta = data['theta']
    n_r = int(np.prod(data['xi'].shape[1:]))
    rng = np.random.default_rng(len(theta))
    many = theta[rng.integers(0, len(theta), 140)].copy()
    many[:, :12] *= 1.0 + 0.003 * rng.normal(size=(140, 12))
    for modulate, suffix in ((True, ''), (False, '_nomodulate')):
        ngal, xi = halotab.predict_batch(theta, family='leauthaud11',
                                         modulate_with_cenocc=modulate)
        assert fused_ran(halotab), 'the fused kernel did not run'
        assert_rel(ngal, data['ngal' + suffix], RTOL, 'ngal' + suffix)
        assert_rel(xi, data['xi' + suffix], RTOL, 'xi' + suffix)
        # more than two tiles: oracle, and the three-kernel path to rounding
        ngal, xi = halotab.predict_batch(many, family='leauthaud11',
                                         modulate_with_cenocc=modulate)
        assert last_launch(halotab)[:3] == ((140 + draws - 1) // draws, 8, 0)
        want = oracle.predict_leauthaud11_batch(table, many, modulate_with_cenocc=modulate)
        assert_rel(ngal, want[0], RTOL, 'ngal, 140 draws')
        assert_rel(xi, want[1], RTOL, 'xi, 140 draws')
        force_fused(halotab, False)
        ngal3, xi3 = halotab.predict_batch(many, family='leauthaud11',
                                           modulate_with_cenocc=modulate)
        assert last_launch(halotab)[2] > 0
        force_fused(halotab)
        assert_rel(ngal, ngal3, 1e-13)
        assert_rel(xi, xi3, 1e-12)
        # separated by galaxy type
        ngal_s, xi_s = halotab.predict_batch(many[:70], family='leauthaud11',
                                             modulate_with_cenocc=modulate,
                                             separate_gal_type=True)
        assert fused_ran(halotab)
        want = oracle.predict_leauthaud11_batch(table, many[:70], separate_gal_type=True,
                                                modulate_with_cenocc=modulate)
        for key in want[0]:
            assert_rel(ngal_s[key], want[0][key], RTOL, 'ngal ' + key)
        for key in want[1]:
            assert_rel(xi_s[key], want[1][key], RTOL, 'xi ' + key, floor=1e-13)
    # any n_gauss_prim
    for n_gauss in (3, 16):
        ngal, xi = halotab.predict_batch(many[:65], family='leauthaud11', n_gauss_prim=n_gauss,
                                         modulate_with_cenocc=True)
        assert fused_ran(halotab)
        want = oracle.predict_leauthaud11_batch(table, many[:65], n_gauss_prim=n_gauss)
        assert_rel(ngal, want[0], RTOL)
        assert_rel(xi, want[1], RTOL)
    # the fused likelihood (tables with up to 20 r values)
    expect = data['xi'].reshape(len(theta), n_r)
    vector = expect[0] * 1.05
    a = rng.normal(size=(n_r, n_r))
    precision = a @ a.T / np.mean(np.abs(vector))**2
    delta = expect - vector
    n_chi, chi2 = halotab.chi2_batch(theta, vector, precision, family='leauthaud11',
                                     modulate_with_cenocc=True)
    assert fused_ran(halotab)
    assert_rel(n_chi, data['ngal'], RTOL)
    assert_rel(chi2, np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-9)
    # NaN parameters reject the draw and no other
    bad = many.copy()
    bad[1, 5] = np.nan        # scatter
    bad[66, 6] = np.nan       # alphasat
    ngal, xi = halotab.predict_batch(bad, family='leauthaud11', modulate_with_cenocc=True)
    assert fused_ran(halotab)
    assert np.isnan(ngal[1]) and np.isnan(ngal[66]) and np.all(np.isnan(xi[[1, 66]]))
    keep = np.setdiff1d(np.arange(140), [1, 66])
    want = oracle.predict_leauthaud11_batch(table, many[keep], modulate_with_cenocc=True)
    assert_rel(xi[keep], want[1], RTOL)


def test_fused_at_the_benchmarked_batch_size():
    """10^4 draws of BASELINE configs[1] through the asynchronous entry points with default
    options (the one-launch path is taken by itself), against the oracle on draws from both
    ends and the middle of the batch and against the three-kernel path on all of them."""
    from tabcorr_amd import pinned_array, pinned_empty, synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    n = 10000
    theta = pinned_array(synthetic.zheng07_draws(n, seed=1))
    halotab = make_tabcorr(table)
    ngal, xi = pinned_empty(n), pinned_empty((n, 19))
    pending = [halotab.predict_batch_async(theta, out=(ngal, xi)) for _ in range(3)]
    for item in pending:
        item.wait()
    assert fused_ran(halotab), 'the fused kernel did not run'
    index = np.r_[0:40, 4990:5030, 9960:10000]
    expect = oracle.predict_zheng07_batch(table, theta[index])
    assert_rel(ngal[index], expect[0], RTOL)
    assert_rel(xi[index], expect[1], RTOL)
    set_option(halotab, 'sync_chunks', -1)             # (the serial path: ...
    set_option(halotab, 'fused_spread', 0)             # ... without the latency form ...
    ngal3, xi3 = halotab.predict_batch(theta)          # ... synchronous, three kernels)
    assert last_launch(halotab)[2] > 0
    assert_rel(ngal, ngal3, 1e-13)
    assert_rel(xi, xi3, 1e-12)
    rng = np.random.default_rng(0)
    vector = expect[1][0] * 1.1
    a = rng.normal(size=(19, 19))
    precision = a @ a.T / np.mean(vector)**2
    want = np.einsum('bi,ij,bj->b', xi3 - vector, precision, xi3 - vector)
    n_chi, chi2 = halotab.chi2_batch_async(theta, vector, precision).wait()
    assert fused_ran(halotab)
    assert_rel(n_chi, ngal3, 1e-13)
    assert_rel(chi2, want, 1e-9)
    # separated by galaxy type, asynchronously: one launch as well
    n_sep, x_sep = halotab.predict_batch_async(theta, separate_gal_type=True).wait()
    assert fused_ran(halotab)
    n_sep3, x_sep3 = halotab.predict_batch(theta, separate_gal_type=True)
    assert last_launch(halotab)[2] > 0
    scale = np.max(np.abs(xi3))
    for key in x_sep3:
        np.testing.assert_allclose(x_sep[key], x_sep3[key], rtol=1e-11, atol=1e-13 * scale)
    for key in n_sep3:
        assert_rel(n_sep[key], n_sep3[key], 1e-13)
    np.testing.assert_allclose(sum(x_sep.values()), xi3, rtol=1e-11, atol=1e-13 * scale)


@pytest.mark.parametrize('modulate, assembias', [(True, False), (False, True), (True, True)])
def test_fused_decorated_variants(modulate, assembias):
    """modulate_with_cenocc / Heaviside assembly bias (two percentile bins, strengths beyond
    [-1, 1] clipped) through the one-launch path, against the oracle."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    rng = np.random.default_rng(11)
    table = synthetic.synthetic_table(11, 2, (9, ), 'auto', seed=21)
    n_draws = 150
    theta = synthetic.zheng07_draws(n_draws, seed=8)
    strengths = rng.uniform(-1.2, 1.2, (n_draws, 2))
    expect = oracle.predict_zheng07_batch(table, theta, modulate_with_cenocc=modulate,
                                          assembias=strengths if assembias else None)
    batch = np.hstack([theta, strengths]) if assembias else theta
    halotab = make_tabcorr(table)
    force_fused(halotab)
    ngal, xi = halotab.predict_batch(batch, modulate_with_cenocc=modulate, assembias=assembias)
    assert fused_ran(halotab), 'the fused kernel did not run'
    assert_rel(ngal, expect[0], RTOL, 'ngal')
    assert_rel(xi, expect[1], RTOL, 'xi')
    # any other n_gauss_prim: the three kernels serve the decorated variants
    halotab.predict_batch(batch, modulate_with_cenocc=modulate, assembias=assembias,
                          n_gauss_prim=7)
    assert last_launch(halotab)[2] > 0


def test_a_draws_result_does_not_depend_on_its_place_in_the_batch():
    """Size-independent property at the benchmarked size: reversing / re-batching 10^4 draws
    permutes the results bit for bit (every lane does the same arithmetic whatever tile it
    sits in; the split of the triangle over the waves is fixed)."""
    from tabcorr_amd import pinned_array, synthetic
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    n = 10000
    theta = synthetic.zheng07_draws(n, seed=21)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    set_option(halotab, 'fused_draws', 64)     # (one shape of workgroup for every batch size)
    # (with the moment expansions of csrc/series.h ON -- the default: a lane adds the terms its
    # OWN draw asks for, whatever the narrowest sigma_logM / largest M0 of its wavefront is)
    set_option(halotab, 'series', 3)
    ngal, xi = halotab.predict_batch(theta)
    assert fused_ran(halotab, (8, ))
    ngal_r, xi_r = halotab.predict_batch(theta[::-1].copy())
    assert np.array_equal(ngal_r[::-1], ngal)
    assert np.array_equal(xi_r[::-1], xi)
    # a batch of another size, other neighbours in the tile
    ngal_p, xi_p = halotab.predict_batch(theta[3000:3777])
    assert np.array_equal(ngal_p, ngal[3000:3777])
    assert np.array_equal(xi_p, xi[3000:3777])
    # draws whose neighbours force every path of the wave: a narrow sigma_logM (node loop for
    # that lane, expansion for the others), alpha outside [0, 4], draws to fix up
    mixed = theta[:640].copy()
    mixed[5::64, 1] = 1e-3
    mixed[9::64, 4] = 5.5
    mixed[17::64, 0] = np.inf
    mixed[33::64, 3] = np.nan
    ngal_m, xi_m = halotab.predict_batch(mixed)
    untouched = np.ones(640, dtype=bool)
    for first in (5, 9, 17, 33):
        untouched[first::64] = False
    assert np.array_equal(ngal_m[untouched], ngal[:640][untouched])
    assert np.array_equal(xi_m[untouched], xi[:640][untouched])
    # ... and the three-kernel path (occ_zheng07_kernel)
    force_fused(halotab, False)
    ngal_3, xi_3 = halotab.predict_batch(theta[:640])
    ngal_3m, xi_3m = halotab.predict_batch(mixed)
    assert np.array_equal(ngal_3m[untouched], ngal_3[untouched])
    assert np.array_equal(xi_3m[untouched], xi_3[untouched])
    occ = halotab.mean_occupation_batch(theta[:640])
    occ_m = halotab.mean_occupation_batch(mixed)
    assert np.array_equal(occ_m[untouched], occ[untouched])
    # the expansions against the node loops: rounding
    set_option(halotab, 'series', 0)
    force_fused(halotab)
    set_option(halotab, 'fused_draws', 64)
    ngal_n, xi_n = halotab.predict_batch(theta)
    assert fused_ran(halotab, (8, ))
    assert_rel(ngal, ngal_n, 1e-14)
    assert_rel(xi, xi_n, 1e-13)


@pytest.mark.parametrize('form', ['occupation', 'three kernels', 'one launch'])
def test_infinite_log_m_min_next_to_regular_draws(form):
    """a draw with logMmin = +-inf (or huge) and a regular sigma_logM in a wave of
    regular draws takes the moment expansion, whose recurrence overflows there; the result is
    the plateau's -+m_0 (N_cen = 0 or 1) as in the node loop and the reference, not NaN."""
    import warnings
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = make_tabcorr(table)
    set_option(halotab, 'series', 3)
    theta = synthetic.zheng07_draws(128, seed=31)
    theta[:, 1] = np.random.default_rng(2).uniform(0.3, 0.8, 128)
    special = {3: np.inf, 40: -np.inf, 70: 1e15, 101: -1e15, 127: 1e300}
    for row, value in special.items():
        theta[row, 0] = value
    rows = sorted(special) + [0, 1, 64]
    with np.errstate(all='ignore'), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        expect = oracle.predict_zheng07_batch(table, theta[rows])
    if form == 'occupation':
        occupation = halotab.mean_occupation_batch(theta)
        with np.errstate(all='ignore'):
            want = np.array([oracle.mean_occupation(table, oracle.Zheng07(theta[r]))
                             for r in rows])
        assert np.all(np.isfinite(occupation[rows]))
        assert_rel(occupation[rows], want, RTOL, 'occupation')
        return
    force_fused(halotab, form == 'one launch')
    if form == 'one launch':
        set_option(halotab, 'fused_draws', 64)
    ngal, xi = halotab.predict_batch(theta)
    assert fused_ran(halotab, (8, )) == (form == 'one launch')
    # (logMmin = +inf: no centrals, satellites only; -inf: every halo hosts a central)
    assert np.all(np.isfinite(ngal[rows])) and np.all(np.isfinite(xi[rows]))
    assert_rel(ngal[rows], expect[0], RTOL, 'ngal')
    assert_rel(xi[rows], expect[1], RTOL, 'xi')


def test_fused_matches_the_references_own_tables():
    """The reference's example table (docs/examples/bolplanck_wp.hdf5: 60 bins, 19 r_p bins) and
    the other fixtures recorded by running the reference, through the one-launch path: any
    n_gauss_prim, modulate_with_cenocc, one r value, tables without prim_haloprop_dist_index,
    assembly bias on the 200-bin table (one workgroup per CU: forced)."""
    data = load_golden('bolplanck_wp')
    halotab = make_tabcorr(table_from_golden(data))
    force_fused(halotab)
    for suffix, kwargs in (('', {}), ('_ng1', {'n_gauss_prim': 1}),
                           ('_ng100', {'n_gauss_prim': 100}),
                           ('_modulate', {'modulate_with_cenocc': True})):
        ngal, xi = halotab.predict_batch(data['theta'], **kwargs)
        assert fused_ran(halotab), suffix
        assert_rel(ngal, data['ngal' + suffix], RTOL, 'ngal' + suffix)
        assert_rel(xi, data['xi' + suffix], RTOL, 'xi' + suffix)
    for name, prefixes in (('synthetic_r1', ('', )), ('synthetic_small_auto', ('', 'legacy_'))):
        data = load_golden(name)
        for prefix in prefixes:
            table = table_from_golden(data)
            if prefix == 'legacy_':      # (a table without prim_haloprop_dist_index)
                names = [n for n in table['gal_type'].dtype.names
                         if n != 'prim_haloprop_dist_index']
                table = dict(table, gal_type=table['gal_type'][names])
            halotab = make_tabcorr(table)
            force_fused(halotab)
            ngal, xi = halotab.predict_batch(data['theta'])
            assert fused_ran(halotab), name + prefix
            assert_rel(ngal, data[prefix + 'ngal'], RTOL, name)
            assert_rel(xi, data[prefix + 'xi'], RTOL, name)
    data = load_golden('synthetic_cfg3')
    halotab = make_tabcorr(table_from_golden(data))
    force_fused(halotab)
    ngal, xi = halotab.predict_batch(np.hstack([data['theta'], data['assembias']]),
                                     assembias=True)
    assert fused_ran(halotab)
    assert_rel(ngal, data['ngal'], RTOL)
    assert_rel(xi, data['xi'], RTOL)
    ngal, xi = halotab.predict_batch(data['theta'])
    assert_rel(xi, data['plain_xi'], RTOL)


@pytest.mark.parametrize('n_prim, n_sec, n_r, n_draws, kwargs', [
    (50, 1, 19, 1000, {}),                                  # BASELINE configs[1]'s table
    (7, 1, 3, 65, {}),                                      # 7 + 7 bins: padded blocks
    (13, 2, 12, 129, {'modulate_with_cenocc': True}),
    (9, 1, 20, 64, {'n_gauss_prim': 5}),
    (6, 2, 5, 200, {'assembias': True}),
])
def test_fused_separate_gal_type(n_prim, n_sec, n_r, n_draws, kwargs):
    """cen-cen / cen-sat / sat-sat through the one-launch path (wave 0 of a tile the cen-cen
    triangle, waves 1 and 2 the halves of the cen-sat rectangle, wave 3 the sat-sat triangle)
    against the oracle; the components add up to the total."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    rng = np.random.default_rng(n_draws)
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim)
    theta = synthetic.zheng07_draws(n_draws, seed=n_draws)
    kwargs = dict(kwargs)
    strengths = rng.uniform(-1.2, 1.2, (n_draws, 2)) if kwargs.pop('assembias', False) else None
    expect = oracle.predict_zheng07_batch(
        table, theta, separate_gal_type=True, n_gauss_prim=kwargs.get('n_gauss_prim', 10),
        modulate_with_cenocc=kwargs.get('modulate_with_cenocc', False), assembias=strengths)
    batch = theta if strengths is None else np.hstack([theta, strengths])
    if strengths is not None:
        kwargs['assembias'] = True
    halotab = make_tabcorr(table)
    force_fused(halotab)
    ngal, xi = halotab.predict_batch(batch, separate_gal_type=True, **kwargs)
    assert fused_ran(halotab), 'the fused kernel did not run'
    assert list(xi.keys()) == list(expect[1].keys())
    scale = max(np.max(np.abs(v)) for v in expect[1].values())
    for key in expect[0]:
        assert_rel(ngal[key], expect[0][key], RTOL, key)
    for key in expect[1]:
        np.testing.assert_allclose(xi[key], expect[1][key], rtol=RTOL, atol=1e-13 * scale,
                                   err_msg=key)
    total_ngal, total_xi = halotab.predict_batch(batch, **kwargs)
    assert_rel(sum(ngal.values()), total_ngal, 1e-12)
    np.testing.assert_allclose(sum(xi.values()), total_xi, rtol=1e-11, atol=1e-13 * scale)


def test_fused_separate_matches_golden():
    data = load_golden('bolplanck_wp')
    halotab = make_tabcorr(table_from_golden(data))
    force_fused(halotab)
    ngal, xi = halotab.predict_batch(data['theta'], separate_gal_type=True)
    assert fused_ran(halotab)
    for key in ngal:
        assert_rel(ngal[key], data['ngal_sep_' + key], RTOL, key)
    for key in xi:
        assert_rel(xi[key], data['xi_sep_' + key], RTOL, key)
    # NaN parameters poison every component, as the reference's arithmetic does
    theta = np.array(data['theta'][:8])
    theta[3, 0] = np.nan
    with np.errstate(all='ignore'):
        ngal, xi = halotab.predict_batch(theta, separate_gal_type=True)
        force_fused(halotab, False)
        ngal3, xi3 = halotab.predict_batch(theta, separate_gal_type=True)
    for key in xi:
        assert np.array_equal(np.isnan(xi[key]), np.isnan(xi3[key])), key
        assert np.all(np.isnan(xi[key][3]))


@pytest.mark.parametrize('n_prim, n_sec, n_r, n_draws, kwargs', [
    (50, 1, 19, 1000, {}),                                    # forced on BASELINE configs[1]'s table
    (50, 2, 19, 300, {}),                                     # 200 bins: BASELINE configs[2]'s table
    (50, 2, 19, 200, {'assembias': True}),
    (50, 2, 7, 130, {'modulate_with_cenocc': True}),
    (60, 2, 19, 100, {}),                                     # 240 bins: 158 KB of LDS
    (31, 2, 13, 65, {'n_gauss_prim': 4}),                     # 124 bins
    (7, 1, 3, 65, {}),                                        # forced: fewer block rows than parts
])
def test_fused_sixteen_waves(n_prim, n_sec, n_r, n_draws, kwargs):
    """One workgroup of 16 waves per CU (eight parts of the units per 32-draw tile, up to 160 KB
    of LDS: tables of 105 ... 248 bins, where two 8-wave workgroups do not fit a CU), total and
    separated by galaxy type, against the oracle; forced on small tables too."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    rng = np.random.default_rng(n_draws)
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim)
    theta = synthetic.zheng07_draws(n_draws, seed=n_draws)
    kwargs = dict(kwargs)
    strengths = rng.uniform(-1.2, 1.2, (n_draws, 2)) if kwargs.pop('assembias', False) else None
    halotab = make_tabcorr(table)
    force_fused(halotab)
    set_option(halotab, 'fused_waves', 16)
    batch = theta if strengths is None else np.hstack([theta, strengths])
    flags = dict(kwargs, assembias=True) if strengths is not None else kwargs
    for separate in (False, True):
        expect = oracle.predict_zheng07_batch(
            table, theta, separate_gal_type=separate, assembias=strengths, **kwargs)
        ngal, xi = halotab.predict_batch(batch, separate_gal_type=separate, **flags)
        assert fused_ran(halotab, (16, )), 'the 16-wave kernel did not run'
        if separate:
            for key in expect[0]:
                assert_rel(ngal[key], expect[0][key], RTOL, 'ngal ' + key)
            for key in expect[1]:
                assert_rel(xi[key], expect[1][key], RTOL, 'xi ' + key, floor=1e-13)
        else:
            assert_rel(ngal, expect[0], RTOL, 'ngal')
            assert_rel(xi, expect[1], RTOL, 'xi')
            total = xi
    # the fused likelihood
    if n_r <= 20 and strengths is None:
        vector = total[0] * 1.1
        a = rng.normal(size=(n_r, n_r))
        precision = a @ a.T / np.mean(vector)**2
        delta = total - vector
        n_chi, chi2 = halotab.chi2_batch(theta, vector, precision, **kwargs)
        assert fused_ran(halotab, (16, ))
        assert_rel(chi2, np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-9)


@pytest.mark.parametrize('n_prim, n_sec, n_r, n_draws, kwargs', [
    (50, 1, 19, 1000, {}),                                  # BASELINE configs[1]'s table
    (30, 1, 19, 333, {}),                                   # the reference's example table size
    (7, 1, 1, 1, {}),
    (5, 2, 3, 31, {}),
    (13, 1, 4, 32, {}),
    (10, 1, 5, 33, {'modulate_with_cenocc': True}),
    (25, 1, 9, 200, {}),
    (26, 2, 20, 97, {'modulate_with_cenocc': True}),        # 104 bins
    (50, 2, 19, 100, {}),                                   # 200 bins: eight waves per workgroup
    (52, 2, 7, 65, {"modulate_with_cenocc": True}),         # 208 bins
    (28, 2, 12, 40, {}),                                    # 112 bins
])
def test_fused_workgroups_of_32_draws(n_prim, n_sec, n_r, n_draws, kwargs):
    """predict_fused_kernel<..., W = 8, DL = 32>: lanes = (draw, half of a bin's nodes) in the
    occupation phase, one 32-draw tile per workgroup -- total, separated by galaxy type and the
    fused likelihood against the oracle; equal to the 64-draw form to rounding; NaN / tied
    parameters as the 64-draw form; a draw's result does not depend on its place in the batch."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    rng = np.random.default_rng(n_draws)
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim + 1)
    theta = synthetic.zheng07_draws(n_draws, seed=n_draws + 3)
    halotab = make_tabcorr(table)
    force_fused(halotab)
    set_option(halotab, 'fused_draws', 32)
    for separate in (False, True):
        expect = oracle.predict_zheng07_batch(table, theta, separate_gal_type=separate, **kwargs)
        ngal, xi = halotab.predict_batch(theta, separate_gal_type=separate, **kwargs)
        launch = last_launch(halotab)
        wide = 2 * n_prim * n_sec > 104
        assert launch[:3] == ((n_draws + 31) // 32, 8, 0), 'the 32-draw kernel did not run'
        if separate:
            for key in expect[0]:
                assert_rel(ngal[key], expect[0][key], RTOL, 'ngal ' + key)
            for key in expect[1]:
                assert_rel(xi[key], expect[1][key], RTOL, 'xi ' + key, floor=1e-13)
        else:
            assert_rel(ngal, expect[0], RTOL, 'ngal')
            assert_rel(xi, expect[1], RTOL, 'xi')
            total = (ngal, xi)
    set_option(halotab, 'fused_draws', 64)
    ngal64, xi64 = halotab.predict_batch(theta, **kwargs)
    assert last_launch(halotab)[:2] == ((n_draws + 63) // 64, 16 if wide else 8)
    assert_rel(total[0], ngal64, 1e-13)
    assert_rel(total[1], xi64, 1e-12)
    set_option(halotab, 'fused_draws', 32)
    # the fused likelihood
    vector = total[1][0] * 1.1
    a = rng.normal(size=(n_r, n_r))
    precision = a @ a.T / np.mean(vector)**2
    delta = total[1] - vector
    n_chi, chi2 = halotab.chi2_batch(theta, vector, precision, **kwargs)
    assert last_launch(halotab)[:2] == ((n_draws + 31) // 32, 8)
    assert_rel(n_chi, total[0], 1e-13)
    assert_rel(chi2, np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-9)
    # reversed / re-batched: bit for bit
    ngal_r, xi_r = halotab.predict_batch(theta[::-1].copy(), **kwargs)
    assert np.array_equal(ngal_r[::-1], total[0]) and np.array_equal(xi_r[::-1], total[1])
    if n_draws > 40:
        ngal_p, xi_p = halotab.predict_batch(theta[7:40], **kwargs)
        assert np.array_equal(xi_p, total[1][7:40])
    # parameters the node loop cannot represent: the same NaN / inf pattern as the 64-draw form
    bad = np.tile(theta[:1], (70, 1)) if n_draws < 70 else theta[:70].copy()
    bad[3, 0] = np.nan                         # logMmin
    bad[5, 3] = np.nan                         # logM1
    bad[9, 1] = 0.0                            # sigma_logM = 0: a step function
    bad[34, 4] = np.nan                        # alpha
    bad[40, 2] = np.nan                        # logM0: no satellites
    bad[41, 3] = -300.0                        # M1 = 0
    got = halotab.predict_batch(bad, **kwargs)
    set_option(halotab, 'fused_draws', 64)
    want = halotab.predict_batch(bad, **kwargs)
    assert np.array_equal(np.isnan(got[0]), np.isnan(want[0]))
    assert np.array_equal(np.isnan(got[1]), np.isnan(want[1]))
    assert np.array_equal(np.isinf(got[1]), np.isinf(want[1]))
    good = np.isfinite(want[1])
    assert_rel(got[1][good], want[1][good], 1e-12)


def test_workgroups_of_32_draws_are_taken_below_8192_draws():
    from tabcorr_amd import pinned_array, pinned_empty, synthetic
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = make_tabcorr(table)
    for n, draws in ((1200, 32), (4096, 32), (8191, 32), (8192, 64), (10000, 64)):
        theta = pinned_array(synthetic.zheng07_draws(n, seed=1))
        out = (pinned_empty(n), pinned_empty((n, 19)))
        halotab.predict_batch_async(theta, out=out).wait()
        launch = last_launch(halotab)
        assert launch[:3] == ((n + draws - 1) // draws, 8, 0), (n, launch)


@pytest.mark.parametrize('n_prim, n_sec, n_r, n_draws, modulate', [
    (50, 2, 19, 200, False),        # BASELINE configs[2]'s table: 200 bins
    (20, 2, 6, 70, True),           # 80 bins
    (13, 2, 3, 33, False),
])
def test_fused_32_draws_with_assembly_bias(n_prim, n_sec, n_r, n_draws, modulate):
    """Heaviside assembly bias (median split) in the halves of the 32-draw workgroups, total and
    separated by galaxy type, against the oracle; strengths beyond [-1, 1] are clipped."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    rng = np.random.default_rng(n_draws)
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim)
    theta = synthetic.zheng07_draws(n_draws, seed=n_draws)
    strengths = rng.uniform(-1.2, 1.2, (n_draws, 2))
    halotab = make_tabcorr(table)
    force_fused(halotab)
    set_option(halotab, 'fused_draws', 32)
    batch = np.hstack([theta, strengths])
    for separate in (False, True):
        expect = oracle.predict_zheng07_batch(table, theta, separate_gal_type=separate,
                                              assembias=strengths, modulate_with_cenocc=modulate)
        ngal, xi = halotab.predict_batch(batch, separate_gal_type=separate, assembias=True,
                                         modulate_with_cenocc=modulate)
        assert last_launch(halotab)[:3] == ((n_draws + 31) // 32, 8, 0)
        if separate:
            for key in expect[0]:
                assert_rel(ngal[key], expect[0][key], RTOL, 'ngal ' + key)
            for key in expect[1]:
                assert_rel(xi[key], expect[1][key], RTOL, 'xi ' + key, floor=1e-13)
        else:
            assert_rel(ngal, expect[0], RTOL, 'ngal')
            assert_rel(xi, expect[1], RTOL, 'xi')
    bad = batch.copy()
    bad[2, 5] = np.nan                  # a NaN strength: the centrals of that draw
    bad[3, 3] = -300.0                  # M1 = 0 with a decorated satellite occupation
    got = halotab.predict_batch(bad, assembias=True, modulate_with_cenocc=modulate)
    force_fused(halotab, False)
    want = halotab.predict_batch(bad, assembias=True, modulate_with_cenocc=modulate)
    assert np.array_equal(np.isnan(got[1]), np.isnan(want[1]))
    good = np.isfinite(want[1])
    assert_rel(got[1][good], want[1][good], 1e-12)


@pytest.mark.parametrize('how', ['asked', 'by itself'])
@pytest.mark.parametrize('n_prim, n_r', [(40, 8), (100, 19), (50, 19)])
def test_autotune_picks_the_fastest_form(n_prim, n_r, how):
    """Option "autotune" (TabCorr.autotune): the measured choice between three kernels and the
    one-launch forms is never far behind the best forced form -- the built-in estimate, fitted
    on a handful of shapes, is up to 40 % behind on others (tools/r04_autotune.py) -- and the
    results do not depend on it.  'by itself' (VERDICT r04 item 5): nobody calls autotune();
    with option "autotune_after" = 256 (opt-in since round 6: ADVICE r05) the 256th pipelined
    call with these flags measures."""
    import ctypes
    import time
    from tabcorr_amd import synthetic, _lib
    from oracle import tabcorr_oracle as oracle
    lib = _lib.load()
    table = synthetic.synthetic_table(n_prim, 1, (n_r, ), 'auto', seed=n_prim)
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    theta = synthetic.zheng07_draws(20000, seed=2)
    pointers = [ctypes.c_void_p() for _ in range(3)]
    for ptr, count in zip(pointers, (theta.size, 20000, 20000 * n_r)):
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
    d_theta, d_ngal, d_xi = pointers
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p), theta.nbytes))

    def us_per_call(n, seconds=0.12):
        def call():
            _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, n, 10, 0, d_ngal,
                                                           d_xi))
        for _ in range(200):
            call()
        _lib.check(lib.tc_table_synchronize(handle))
        count, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            for _ in range(20):
                call()
            count += 20
        _lib.check(lib.tc_table_synchronize(handle))
        return (time.perf_counter() - t0) / count * 1e6

    if how == 'asked':
        result = halotab.autotune()
    else:
        assert halotab.autotune(measure=False) is None
        # (off by default since round 6: the loop of calls below must not measure ...)
        for _ in range(300):
            _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, 3000, 10, 0, d_ngal,
                                                           d_xi))
        _lib.check(lib.tc_table_synchronize(handle))
        assert halotab.autotune(measure=False) is None, 'autotune ran although nobody asked'
        # (... until the caller opts in)
        set_option(halotab, 'autotune_after', 256)
        for _ in range(300):
            _lib.check(lib.tc_predict_zheng07_batch_device(handle, d_theta, 5, 3000, 10, 0, d_ngal,
                                                           d_xi))
        _lib.check(lib.tc_table_synchronize(handle))
        result = halotab.autotune(measure=False)
        assert result is not None, 'the 256th pipelined call did not measure'
    assert list(result['sizes']) == [256 << i for i in range(9)]
    assert set(result['forms']) <= {0, 32, 64} and np.all(result['us_per_call'][:, 0] > 0)
    sizes = (3000, 12000)          # (device-bound; below ~1000 draws the host thread binds)
    tuned = {n: min(us_per_call(n), us_per_call(n)) for n in sizes}
    ngal, xi = halotab.predict_batch_async(theta[:3000]).wait()
    expect = oracle.predict_zheng07_batch(table, theta[:8])
    assert_rel(ngal[:8], expect[0], RTOL)
    assert_rel(xi[:8], expect[1], RTOL)
    set_option(halotab, 'autotune', -1)
    set_option(halotab, 'fused_min_draws', 1)
    set_option(halotab, 'fused_max_draws', 1 << 30)
    for n in sizes:
        forced = []
        for fused, shape in ((0, 0), (2, 64), (2, 32)):
            set_option(halotab, 'fused', fused)
            set_option(halotab, 'fused_draws', shape)
            forced.append(us_per_call(n))
        assert tuned[n] <= 1.2 * min(forced), (n, tuned[n], forced)
    for ptr in pointers:
        lib.tc_device_free(ptr)
    with pytest.raises(ValueError):
        set_option(halotab, 'autotune', 64)         # not a combination of predict flags


def test_satellites_by_expansion_with_deferred_pairs():
    """predict_fused_kernel<..., SATDEFER>: undecorated Zheng07 on a table whose expansions
    serve every bin (64-draw workgroups): the satellites' expansion from the bins' records for
    the draws the shortest expansion serves, the other (bin, draw) pairs after the wave's bins
    -- against the oracle, the same kernel with the node loops in place, separated by galaxy
    type, with draws that need every exit, and wherever a draw sits in the batch."""
    import warnings
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    n = 1500
    theta = synthetic.zheng07_draws(n, seed=77)
    rng = np.random.default_rng(3)
    theta[:200, 2] = rng.uniform(12.9, 14.8, 200)       # M0 inside / above the upper bins
    theta[200:260, 2] = 9.0                             # ... below all of them
    theta[300, 4] = 5.5                                 # alpha outside [0, 4]: node loop
    theta[301, 4] = -0.5
    theta[302, 2] = np.inf
    theta[303, 2] = -np.inf
    theta[304, 3] = np.nan
    theta[305, 1] = 1e-4
    theta[306, 0] = np.inf
    halotab = make_tabcorr(table)
    force_fused(halotab)
    set_option(halotab, 'fused_draws', 64)
    set_option(halotab, 'series', 1)
    results = {}
    for defer in (2, 1, 0):
        set_option(halotab, 'fused_defer', defer)
        for separate in (False, True):
            results[defer, separate] = halotab.predict_batch(theta, separate_gal_type=separate)
            assert fused_ran(halotab, (8, ))
    with warnings.catch_warnings(), np.errstate(all='ignore'):
        warnings.simplefilter('ignore')
        expect = oracle.predict_zheng07_batch(table, theta)
        expect_sep = oracle.predict_zheng07_batch(table, theta, separate_gal_type=True)
    # (2, the default: the centrals by their records as well, those no expansion serves -- the
    # step-like sigma_logM, the draws to fix up -- among the deferred pairs; 1: the satellites only)
    for defer in (1, 2):
        ngal, xi = results[defer, False]
        good = np.isfinite(expect[0])
        assert np.array_equal(np.isnan(ngal), np.isnan(expect[0]))
        assert_rel(ngal[good], expect[0][good], 1e-12)
        assert_rel(xi[good], expect[1][good], 1e-11)
    ngal, xi = results[2, False]
    finite = np.isfinite(expect[0])
    assert np.array_equal(np.isnan(ngal), np.isnan(expect[0]))
    assert np.array_equal(np.isinf(ngal), np.isinf(expect[0]))
    assert_rel(ngal[finite], expect[0][finite], 1e-12)
    assert_rel(xi[finite], expect[1][finite], 1e-11)
    assert_rel(ngal[finite], results[0, False][0][finite], 1e-13)
    assert_rel(xi[finite], results[0, False][1][finite], 1e-12)
    for key in expect_sep[0]:
        good = np.isfinite(expect_sep[0][key]) & finite
        assert_rel(results[2, True][0][key][good], expect_sep[0][key][good], 1e-12)
    for key in expect_sep[1]:
        assert_rel(results[2, True][1][key][finite], expect_sep[1][key][finite], 1e-11,
                   floor=1e-13)
    # the same bits wherever the draw sits, whatever its neighbours defer
    set_option(halotab, 'fused_defer', 2)
    order = rng.permutation(n)
    ngal_p, xi_p = halotab.predict_batch(theta[order][:777])
    assert np.array_equal(ngal_p, ngal[order][:777], equal_nan=True)
    assert np.array_equal(xi_p, xi[order][:777], equal_nan=True)


# ---- the latency form: 40 draws per workgroup, one workgroup per CU (round 6) ----------------

def force_latency_form(halotab, on=True):
    set_option(halotab, 'fused_draws', 40 if on else 0)
    set_option(halotab, 'single_draw', 0 if on else 1)
    set_option(halotab, 'sync_chunks', 1 if on else 0)     # (host-array calls in one piece)


def latency_form_ran(halotab, n_draws):
    workgroups, waves, slabs, lds = last_launch(halotab)
    return workgroups == (n_draws + 39) // 40 and waves == 8 and slabs == 0


@pytest.mark.parametrize('n_prim, n_r, n_draws, options', [
    (50, 19, 10000, {}),                 # BASELINE configs[1]: both galaxy types from records
    (50, 19, 41, {'fused_defer': 1}),    # the satellites' records, centrals by their node loops
    (50, 19, 777, {'series': 0}),        # node loops in place
    (30, 19, 1000, {}),                  # the reference's example shape (bins 0.15 dex wide)
    (50, 3, 39, {}),                     # one r sub-tile
    (50, 7, 40, {}),                     # two
    (50, 12, 1, {}),                     # three
    (50, 16, 250, {}),                   # four
    (104, 19, 300, {}),                  # 208 bins: 86 KB of LDS, one workgroup per CU anyway
    (23, 9, 130, {}),                    # a ragged triangle (46 bins: the last block half full)
])
def test_latency_form_matches_oracle_and_three_kernels(n_prim, n_r, n_draws, options):
    """predict_fused_kernel with 40 draws per workgroup (v_mfma_f64_4x4x4, every wave an eighth
    of the units for all 40 draws): the oracle's values on a sample, the three-kernel path's on
    every draw; tabcorr/tabcorr.py:580-650."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(n_prim, 1, (n_r, ), 'auto', seed=n_prim + n_r)
    theta = synthetic.zheng07_draws(n_draws, seed=3)
    halotab = make_tabcorr(table)
    force_fused(halotab, False)
    ngal3, xi3 = halotab.predict_batch(theta)
    force_fused(halotab)
    force_latency_form(halotab)
    for name, value in options.items():
        set_option(halotab, name, value)
    ngal, xi = halotab.predict_batch(theta)
    assert latency_form_ran(halotab, n_draws), last_launch(halotab)
    assert_rel(ngal, ngal3, 1e-13)
    assert_rel(xi, xi3, 1e-12)
    index = np.unique(np.r_[0:min(n_draws, 24), max(0, n_draws - 24):n_draws])
    expect = oracle.predict_zheng07_batch(table, theta[index])
    assert_rel(ngal[index], expect[0], RTOL)
    assert_rel(xi[index], expect[1], RTOL)


def test_latency_form_is_what_a_call_alone_on_the_chip_takes():
    """Default options, BASELINE configs[1]: the synchronous predict_batch(theta) of 10^4 draws in
    ordinary NumPy arrays runs as 40-draw workgroups -- 250 of them, one
    per CU -- and so does a device-pointer call on a handle with one lane; pipelined calls keep
    the 64-draw throughput form; smaller batches the three kernels.  Same values as the
    throughput form to rounding, the oracle's on a sample, a draw's bits independent of its
    place in the batch, degenerate parameters as everywhere else; the fused likelihood."""
    from tabcorr_amd import synthetic, pinned_array
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    n = 10000
    theta = synthetic.zheng07_draws(n, seed=1)
    halotab = make_tabcorr(table)
    ngal, xi = halotab.predict_batch(theta)
    workgroups, waves, slabs, lds = last_launch(halotab)
    assert waves == 8 and slabs == 0, last_launch(halotab)
    assert workgroups in (250, 127, 124), last_launch(halotab)     # (one piece, or two chunks)
    index = np.r_[0:40, 4990:5030, 9960:10000]
    expect = oracle.predict_zheng07_batch(table, theta[index])
    assert_rel(ngal[index], expect[0], RTOL)
    assert_rel(xi[index], expect[1], RTOL)
    # the throughput form (pipelined, asynchronous) on the same draws
    ngal_t, xi_t = halotab.predict_batch_async(pinned_array(theta)).wait()
    assert last_launch(halotab)[:3] == (157, 8, 0), last_launch(halotab)
    assert_rel(ngal, ngal_t, 1e-14)
    assert_rel(xi, xi_t, 1e-12)
    # small batches that run alone: the three kernels spread them over the chip
    halotab.predict_batch(theta[:1500])
    assert last_launch(halotab)[2] > 0, last_launch(halotab)
    # a draw's bits: reversed, and in another batch
    force_latency_form(halotab)
    ngal, xi = halotab.predict_batch(theta)
    assert latency_form_ran(halotab, n)
    ngal_r, xi_r = halotab.predict_batch(theta[::-1].copy())
    assert np.array_equal(ngal_r[::-1], ngal) and np.array_equal(xi_r[::-1], xi)
    ngal_p, xi_p = halotab.predict_batch(theta[3001:3778])
    assert np.array_equal(ngal_p, ngal[3001:3778]) and np.array_equal(xi_p, xi[3001:3778])
    # degenerate parameters next to regular draws: as the three-kernel path has them
    mixed = theta[:400].copy()
    mixed[5::40, 1] = 1e-3
    mixed[9::40, 4] = 5.5
    mixed[17::40, 0] = np.inf
    mixed[21::40, 0] = -np.inf
    mixed[33::40, 3] = np.nan
    mixed[34::40, 2] = -np.inf
    mixed[35::40, 1] = 0.0
    with np.errstate(all='ignore'):
        ngal_m, xi_m = halotab.predict_batch(mixed)
        assert latency_form_ran(halotab, 400)
        force_latency_form(halotab, False)
        force_fused(halotab, False)
        ngal_3, xi_3 = halotab.predict_batch(mixed)
    assert np.array_equal(np.isnan(ngal_m), np.isnan(ngal_3))
    assert np.array_equal(np.isnan(xi_m), np.isnan(xi_3))
    assert np.array_equal(np.isinf(xi_m), np.isinf(xi_3))
    good = np.isfinite(xi_3)
    assert_rel(xi_m[good], xi_3[good], 1e-12)
    # (a workgroup with such a draw adds its waves' row shares one sum per matrix instruction
    # instead of four: the regular draws beside it keep their bits)
    untouched = np.ones(400, dtype=bool)
    for first in (5, 9, 17, 21, 33, 34, 35):
        untouched[first::40] = False
    assert np.array_equal(xi_m[untouched], xi[:400][untouched])
    assert np.array_equal(ngal_m[untouched], ngal[:400][untouched])
    # the fused likelihood
    force_fused(halotab)
    force_latency_form(halotab)
    rng = np.random.default_rng(0)
    vector = expect[1][0] * 1.1
    a = rng.normal(size=(19, 19))
    precision = a @ a.T / np.mean(vector)**2
    want = np.einsum('bi,ij,bj->b', xi - vector, precision, xi - vector)
    n_chi, chi2 = halotab.chi2_batch(theta, vector, precision)
    assert latency_form_ran(halotab, n)
    assert np.array_equal(n_chi, ngal)
    assert_rel(chi2, want, 1e-9)


def test_latency_form_with_entries_that_are_not_finite():
    """A matrix with infinite / NaN entries: the latency form adds its row shares one sum per
    instruction (0 x inf would reach the sums of other draws otherwise) -- the same NaN / inf
    pattern as the three kernels, the same bits for r values the entries do not touch."""
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    clean = make_tabcorr(table)
    force_fused(clean)
    force_latency_form(clean)
    theta = synthetic.zheng07_draws(400, seed=1)
    theta[7::40, 3] = 16.5         # (hardly any satellites: densities of 0 beside the entries)
    ngal_c, xi_c = clean.predict_batch(theta)
    assert latency_form_ran(clean, 400)
    matrix = np.array(table['tpcf_matrix'], dtype=np.float64, copy=True)
    flat = matrix.reshape(19, -1)
    flat[3, 17] = np.inf
    flat[3, flat.shape[1] - 5] = np.inf
    flat[11, 1234] = np.nan
    broken = dict(table, tpcf_matrix=matrix)
    halotab = make_tabcorr(broken)
    with np.errstate(all='ignore'):
        force_fused(halotab, False)
        ngal_3, xi_3 = halotab.predict_batch(theta)
        assert last_launch(halotab)[2] > 0
        force_fused(halotab)
        force_latency_form(halotab)
        ngal, xi = halotab.predict_batch(theta)
    assert latency_form_ran(halotab, 400)
    assert np.array_equal(ngal, ngal_c)
    assert np.array_equal(np.isnan(xi), np.isnan(xi_3))
    assert np.array_equal(np.isinf(xi), np.isinf(xi_3))
    others = [r for r in range(19) if r not in (3, 11)]
    assert np.array_equal(xi[:, others], xi_c[:, others])
    good = np.isfinite(xi_3)
    assert_rel(xi[good], xi_3[good], 1e-12)
