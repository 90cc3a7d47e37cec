"""predict_cross_fused_kernel: tables with mode = 'cross' (one column per halo bin) and
interpolators over such tables, one launch per batch -- occupations once per group of bins,
every member's mean occupation straight into the row sums of all tables, spline weights and
normalisation in the same workgroup -- against the oracle, the three-kernel path and the
reference's own AbacusSummit fixture.  Needs an MI355X."""

import os

import numpy as np
import pytest

from util import REPO, assert_rel

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def make_tabcorr(table, **kwargs):
    from tabcorr_amd import TabCorr
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                               table['tpcf_shape'], table['attrs'], **kwargs)


def set_option(handle, name, value):
    from tabcorr_amd import _lib
    lib = _lib.load()
    _lib.check(lib.tc_table_set_option(handle, name.encode(), value))


def last_launch(handle):
    import ctypes
    from tabcorr_amd import _lib
    lib = _lib.load()
    values = [ctypes.c_int() for _ in range(4)]
    _lib.check(lib.tc_table_last_launch(handle, *[ctypes.byref(v) for v in values]))
    return tuple(v.value for v in values)


def force(handle, on):
    set_option(handle, 'fused', 2 if on else 0)
    set_option(handle, 'fused_min_draws', 1)
    set_option(handle, 'single_draw', 0)


def cross_ran(handle, n_draws):
    """One launch of eight-wave workgroups, one per 64 draws -- or, for tables of up to 16 rows,
    up to eight per tile (each a share of the groups) --, no slabs of partial sums."""
    workgroups, waves, slabs, _ = last_launch(handle)
    tiles = (n_draws + 63) // 64
    return waves == 8 and slabs == 0 and workgroups % tiles == 0 and 1 <= workgroups // tiles <= 8


def compare(got, expect, separate, rtol, what, floor=1e-14):
    if separate:
        for key in expect[0]:
            assert_rel(got[0][key], expect[0][key], rtol, 'ngal %s %s' % (key, what))
        for key in expect[1]:
            assert_rel(got[1][key], expect[1][key], rtol, 'xi %s %s' % (key, what),
                       floor=max(floor, 1e-13))
    else:
        assert_rel(got[0], expect[0], rtol, 'ngal ' + what)
        assert_rel(got[1], expect[1], rtol, 'xi ' + what, floor=floor)


@pytest.mark.parametrize('n_prim, n_sec, n_r, n_draws, kwargs', [
    (20, 1, 13, 200, {}),                                   # 14 rows -> 16 (2 per wave)
    (20, 2, 13, 333, {'assembias': True}),
    (9, 3, 19, 64, {'modulate_with_cenocc': True}),         # 20 rows -> 32
    (30, 2, 40, 129, {}),                                   # 41 rows -> 48
    (12, 2, 60, 65, {'assembias': True, 'modulate_with_cenocc': True}),   # 61 rows -> 64
    (10, 2, 100, 70, {}),                                   # 101 rows -> 128
    (40, 2, 1, 1000, {}),                                   # one r value
    (3, 1, 5, 1, {}),                                       # one draw
])
def test_cross_table_against_oracle_and_three_kernels(n_prim, n_sec, n_r, n_draws, kwargs):
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'cross', seed=n_prim + n_r)
    theta = synthetic.zheng07_draws(n_draws, seed=n_r)
    kwargs = dict(kwargs)
    oracle_kwargs = dict(kwargs)
    batch = theta
    if kwargs.get('assembias'):
        strengths = np.random.default_rng(n_r).uniform(-1.2, 1.2, (n_draws, 2))
        batch = np.hstack([theta, strengths])
        oracle_kwargs['assembias'] = strengths
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    for separate in (False, True):
        if separate and n_r > 60:
            continue        # (two components of 100 r values do not fit the LDS: three kernels)
        expect = oracle.predict_zheng07_batch(table, theta, separate_gal_type=separate,
                                              **oracle_kwargs)
        force(handle, True)
        got = halotab.predict_batch(batch, separate_gal_type=separate, **kwargs)
        assert cross_ran(handle, n_draws), last_launch(handle)
        force(handle, False)
        three = halotab.predict_batch(batch, separate_gal_type=separate, **kwargs)
        assert last_launch(handle)[2] > 0
        compare(got, expect, separate, RTOL, 'vs oracle')
        compare(got, three, separate, 1e-12, 'vs three kernels')


def test_cross_degenerate_parameters_and_ragged_tables():
    """NaN / infinite / tied parameters give what the three-kernel path gives (which
    test_gpu_full_size.py compares with the oracle), NaN for NaN; shuffled rows with missing
    bins (ragged groups whose members are not adjacent)."""
    from tabcorr_amd import synthetic
    from test_gpu_grouped import degenerate_draws, ragged_table
    table = ragged_table(14, 2, 9, seed=4, mode='cross')
    theta = degenerate_draws(synthetic.zheng07_draws(150, seed=8))
    strengths = np.random.default_rng(1).uniform(-1.2, 1.2, (150, 2))
    strengths[30, 0] = np.nan
    strengths[31, 1] = np.nan
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    for kwargs, batch in (({}, theta), ({'modulate_with_cenocc': True}, theta),
                          ({'assembias': True}, np.hstack([theta, strengths]))):
        for separate in (False, True):
            with np.errstate(all='ignore'):
                force(handle, True)
                got = halotab.predict_batch(batch, separate_gal_type=separate, **kwargs)
                assert cross_ran(handle, 150)
                force(handle, False)
                want = halotab.predict_batch(batch, separate_gal_type=separate, **kwargs)
            pairs = ([(got[0], want[0]), (got[1], want[1])] if not separate else
                     [(got[i][key], want[i][key]) for i in (0, 1) for key in want[i]])
            for g, w in pairs:
                assert np.array_equal(np.isnan(g), np.isnan(w)), (kwargs, separate)
                assert np.array_equal(np.isinf(g), np.isinf(w)), (kwargs, separate)
                good = np.isfinite(w)
                assert_rel(g[good], w[good], 1e-12, floor=1e-13)


def test_cross_likelihood_async_and_threshold():
    from tabcorr_amd import pinned_array, pinned_empty, synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(25, 2, (13, ), 'cross', seed=2)
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    set_option(handle, 'single_draw', 0)
    n = 6500
    theta = pinned_array(synthetic.zheng07_draws(n, seed=3))
    index = np.r_[0:3, 4000:4003, n - 3:n]
    expect = oracle.predict_zheng07_batch(table, theta[index])
    ngal, xi = pinned_empty(n), pinned_empty((n, 13))
    got = halotab.predict_batch_async(theta, out=(ngal, xi)).wait()
    assert cross_ran(handle, n), last_launch(handle)
    assert_rel(got[0][index], expect[0], RTOL)
    assert_rel(got[1][index], expect[1], RTOL)
    rng = np.random.default_rng(0)
    vector = expect[1][0] * 1.1
    a = rng.normal(size=(13, 13))
    precision = a @ a.T / np.mean(vector)**2
    want = np.einsum('bi,ij,bj->b', expect[1] - vector, precision, expect[1] - vector)
    n_chi, chi2 = halotab.chi2_batch_async(theta, vector, precision).wait()
    assert cross_ran(handle, n)
    assert_rel(n_chi[index], expect[0], RTOL)
    assert_rel(chi2[index], want, 1e-9)
    # a medium batch: several workgroups per tile of 64 draws (each a share of the groups, the
    # last to arrive adds the shares in order); identical from call to call
    got_small = halotab.predict_batch_async(theta[:2000], out=(ngal[:2000], xi[:2000])).wait()
    launch = last_launch(handle)
    assert cross_ran(handle, 2000) and launch[0] > 32, launch
    small = oracle.predict_zheng07_batch(table, theta[:6])
    assert_rel(got_small[1][:6], small[1], RTOL)
    again = halotab.predict_batch_async(theta[:2000]).wait()
    assert np.array_equal(again[1], got_small[1][:2000]) and np.array_equal(again[0],
                                                                           got_small[0][:2000])
    # tiny batches and calls that run alone keep the three kernels
    halotab.predict_batch_async(theta[:100], out=(ngal[:100], xi[:100])).wait()
    assert last_launch(handle)[2] > 0
    set_option(handle, 'sync_chunks', -1)       # (the serial path)
    halotab.predict_batch(theta)
    assert last_launch(handle)[2] > 0
    sync = halotab.chi2_batch(theta, vector, precision)
    assert_rel(sync[1][index], want, 1e-9)


def synthetic_cross_interpolator(shape, n_prim, n_sec, n_r, seed, vary_n_h=False):
    from tabcorr_amd import Interpolator, synthetic
    tables, keys, points = synthetic.synthetic_interpolator(shape, n_prim, n_sec, (n_r, ),
                                                            'cross', seed=seed)
    if vary_n_h:     # (tables of different cosmologies: their halo mass functions differ)
        rng = np.random.default_rng(seed)
        for table in tables:
            table['gal_type'] = table['gal_type'].copy()
            table['gal_type']['n_h'] *= rng.uniform(0.8, 1.25, len(table['gal_type']))
    interp = Interpolator([make_tabcorr(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    return tables, points, interp


@pytest.mark.parametrize('shape, n_prim, n_sec, n_r, vary_n_h', [
    ((4, ), 20, 2, 13, False),            # 56 rows: the shape of the reference's fixture
    ((4, ), 16, 1, 7, True),              # 32 rows, every table its own n_h
    ((4, 4), 10, 2, 3, True),             # 16 tables x 4 rows
    ((4, 4), 8, 1, 6, False),             # 16 tables x 7 rows = 112 -> 128
    ((5, ), 12, 2, 11, False),            # 60 rows
])
def test_cross_interpolator(shape, n_prim, n_sec, n_r, vary_n_h):
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    tables, points, interp = synthetic_cross_interpolator(shape, n_prim, n_sec, n_r,
                                                          seed=n_prim, vary_n_h=vary_n_h)
    n_draws = 150
    theta = synthetic.zheng07_draws(n_draws, seed=n_r)
    rng = np.random.default_rng(n_r)
    x = np.stack([rng.uniform(xp[0] - 0.1 * (xp[-1] - xp[0]), xp[-1] + 0.1 * (xp[-1] - xp[0]),
                              n_draws) for xp in interp.xp], axis=-1)
    x[3] = [xp[-1] for xp in interp.xp]                 # the right edge (interpolator.py:319-321)
    x[4] = [xp[0] for xp in interp.xp]
    device = interp.to_device()
    handle = device.tables[0].handle
    setup = oracle.interpolator_setup(tables, points)
    for separate in (False, True):
        expect = oracle.interpolator_predict_zheng07_batch(
            tables, setup, theta, x, separate_gal_type=separate, extrapolate=True)
        force(handle, True)
        got = interp.predict_batch(theta, x, separate_gal_type=separate, extrapolate=True)
        assert cross_ran(handle, n_draws), last_launch(handle)
        force(handle, False)
        three = interp.predict_batch(theta, x, separate_gal_type=separate, extrapolate=True)
        assert not cross_ran(handle, n_draws)
        compare(got, expect, separate, RTOL, 'vs oracle', floor=1e-12)
        compare(got, three, separate, 1e-11, 'vs three kernels', floor=1e-12)
    # a NaN coordinate with extrapolate=True: NaN for that draw only (interpolator.py:318-329)
    x_nan = x[:6].copy()
    x_nan[2, 0] = np.nan
    force(handle, True)
    ngal, xi = interp.predict_batch(theta[:6], x_nan, extrapolate=True)
    assert np.isnan(ngal[2]) and np.all(np.isnan(xi[2]))
    keep = [0, 1, 3, 4, 5]
    assert_rel(xi[keep], expect_total(tables, setup, theta[:6], x_nan)[1][keep], RTOL, floor=1e-12)
    with pytest.raises(ValueError):
        interp.predict_batch(theta[:6], x_nan)
    # the likelihood behind the interpolated prediction
    vector = np.full(n_r, np.mean(got[1]['centrals'] if isinstance(got[1], dict) else got[1]))
    precision = np.eye(n_r) / np.mean(vector)**2
    total = oracle.interpolator_predict_zheng07_batch(tables, setup, theta, x, extrapolate=True)
    want = np.einsum('bi,ij,bj->b', total[1] - vector, precision, total[1] - vector)
    n_chi, chi2 = interp.chi2_batch(theta, x, vector, precision, extrapolate=True)
    assert cross_ran(handle, n_draws)
    assert_rel(n_chi, total[0], RTOL)
    assert_rel(chi2, want, 1e-9)


def expect_total(tables, setup, theta, x):
    from oracle import tabcorr_oracle as oracle
    with np.errstate(all='ignore'):
        return oracle.interpolator_predict_zheng07_batch(tables, setup, theta, x,
                                                         extrapolate=True)


def test_cross_reference_fixture_interpolator():
    """The reference's own AbacusSummit interpolator (tests/AbacusSummit/.../ds_efficient.hdf5:
    four cross tables of 1104 bins, 13 r values -- 56 rows): one launch per batch against the
    golden values recorded from the reference and against the three kernels."""
    from tabcorr_amd import Interpolator, synthetic
    from util import load_golden
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    device = interp.to_device()
    handle = device.tables[0].handle
    rng = np.random.default_rng(5)
    n = 700
    theta = synthetic.zheng07_draws(n, seed=9)
    theta[:, 0] = rng.uniform(12.5, 13.3, n)
    theta[:, 3] = rng.uniform(13.6, 14.4, n)
    x = np.stack([rng.uniform(xp[0], xp[-1], n) for xp in interp.xp], axis=-1)
    for separate in (False, True):
        force(handle, True)
        got = interp.predict_batch(theta, x, separate_gal_type=separate)
        assert cross_ran(handle, n), last_launch(handle)
        force(handle, False)
        three = interp.predict_batch(theta, x, separate_gal_type=separate)
        compare(got, three, separate, 1e-11, 'vs three kernels', floor=1e-12)
    golden = load_golden('ds_efficient')
    force(handle, True)
    got = interp.predict_batch(golden['theta'], golden['x'])
    assert cross_ran(handle, len(golden['theta']))
    assert_rel(got[0], golden['ngal'], RTOL)
    assert_rel(got[1], golden['xi'], RTOL, floor=1e-12)
    ngal, xi = interp.predict_batch(golden['theta'], golden['x'], separate_gal_type=True)
    assert cross_ran(handle, len(golden['theta']))
    for key in ('centrals', 'satellites'):
        assert_rel(ngal[key], golden['ngal_sep_' + key], RTOL)
        assert_rel(xi[key], golden['xi_sep_' + key], RTOL, floor=1e-12)
    theta_out = np.repeat(golden['theta'][:1], len(golden['x_out']), axis=0)
    ngal, xi = interp.predict_batch(theta_out, golden['x_out'], extrapolate=True)
    assert_rel(ngal, golden['ngal_out'], RTOL)
    assert_rel(xi, golden['xi_out'], RTOL, floor=1e-12)
    # the first table by itself (14 rows)
    halotab = interp.tabcorr_list[0]
    handle0 = halotab.to_device().handle
    force(handle0, True)
    ngal, xi = halotab.predict_batch(golden['theta'])
    assert cross_ran(handle0, len(golden['theta']))
    assert_rel(ngal, golden['table0_ngal'], RTOL)
    assert_rel(xi, golden['table0_xi'], RTOL)
    ngal, xi = halotab.predict_batch(golden['theta'], separate_gal_type=True)
    assert cross_ran(handle0, len(golden['theta']))
    for key in ('centrals', 'satellites'):
        assert_rel(ngal[key], golden['table0_ngal_sep_' + key], RTOL)
        assert_rel(xi[key], golden['table0_xi_sep_' + key], RTOL, floor=1e-13)


def test_tables_of_few_rows_in_the_chunk_form():
    """Up to 16 rows: large undecorated batches leave the register form for the 32-row chunk
    form (matrix pipe, group records, deferred pairs -- launch.hip: choose_cross_fused).  Forced
    here for small batches: against the oracle, the register form, the reference's fixture, and
    a draw's bits wherever it sits in the batch."""
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    from util import load_golden
    table = synthetic.synthetic_table(40, 2, (13, ), 'cross', seed=7)
    n = 700
    theta = synthetic.zheng07_draws(n, seed=3)
    theta[5, 1] = 1e-3              # (a step: no expansion serves it)
    theta[6, 2] = 15.5              # (M0 above every bin)
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    set_option(handle, 'series', 3)
    force(handle, True)
    narrow_launch = None
    for separate in (False, True):
        expect = oracle.predict_zheng07_batch(table, theta, separate_gal_type=separate)
        set_option(handle, 'cross_wide_min_draws', 0)
        narrow = halotab.predict_batch(theta, separate_gal_type=separate)
        narrow_launch = narrow_launch if separate else last_launch(handle)
        set_option(handle, 'cross_wide_min_draws', 1)
        wide = halotab.predict_batch(theta, separate_gal_type=separate)
        assert cross_ran(handle, n)
        assert separate or last_launch(handle)[3] != narrow_launch[3]
        compare(wide, expect, separate, RTOL, 'vs oracle')
        compare(wide, narrow, separate, 1e-12, 'vs the register form')
    order = np.random.default_rng(1).permutation(n)
    ngal, xi = halotab.predict_batch(theta)
    ngal_p, xi_p = halotab.predict_batch(theta[order][:333])
    assert np.array_equal(xi_p, xi[order][:333]) and np.array_equal(ngal_p, ngal[order][:333])
    # decorated calls keep the register form
    strengths = np.random.default_rng(2).uniform(-1, 1, (n, 2))
    halotab.predict_batch(np.hstack([theta, strengths]), assembias=True)
    assert last_launch(handle)[3] == narrow_launch[3]
    # the reference's AbacusSummit table (13 r values)
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    golden = load_golden('ds_efficient')
    first = interp.tabcorr_list[0]
    handle0 = first.to_device().handle
    force(handle0, True)
    set_option(handle0, 'cross_wide_min_draws', 1)
    ngal, xi = first.predict_batch(golden['theta'])
    assert_rel(ngal, golden['table0_ngal'], RTOL)
    assert_rel(xi, golden['table0_xi'], RTOL)
