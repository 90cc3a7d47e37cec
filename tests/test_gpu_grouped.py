"""Bins that share their quadrature nodes (the secondary-percentile bins of one mass bin,
tabcorr/tabcorr.py:186-205, :548-549): the occupation kernels evaluate every node once per
GROUP of such bins (kernels.hip.h: occ_group_zheng07, option "grouped").  Per bin that is the
same arithmetic in the same order as the per-bin loop -- checked bit for bit --, the sums over
bins may differ in their last bits.  Needs an MI355X."""

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


def set_option(halotab, name, value):
    from tabcorr_amd import _lib
    lib = _lib.load()
    _lib.check(lib.tc_table_set_option(halotab.to_device().handle, name.encode(), value))


def last_launch(halotab):
    import ctypes
    from tabcorr_amd import _lib
    lib = _lib.load()
    values = [ctypes.c_int() for _ in range(4)]
    _lib.check(lib.tc_table_last_launch(halotab.to_device().handle,
                                        *[ctypes.byref(v) for v in values]))
    return tuple(v.value for v in values)


def ragged_table(n_prim, n_sec, n_r, seed, mode='auto'):
    """A table whose rows are shuffled within each galaxy type and of which some are missing
    (the reference drops empty bins: tabcorr.py:226-227), so that the groups are ragged and
    their members are not adjacent."""
    from tabcorr_amd import synthetic
    rng = np.random.default_rng(seed)
    full = synthetic.synthetic_gal_type(n_prim, n_sec, seed=seed)
    half = len(full) // 2
    keep_cen = np.sort(rng.permutation(half)[:max(2, int(0.8 * half))])
    keep_sat = half + np.sort(rng.permutation(half)[:max(2, int(0.7 * half))])
    rows = np.concatenate([rng.permutation(keep_cen), rng.permutation(keep_sat)])
    gal_type = full[rows]
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), mode, seed=seed)
    table['gal_type'] = gal_type
    table['tpcf_matrix'] = synthetic.synthetic_tpcf_matrix(len(gal_type), n_r, mode=mode,
                                                           seed=seed + 1)
    return table


def degenerate_draws(theta):
    theta = theta.copy()
    theta[1, 0] = np.nan
    theta[2, 1] = 0.0
    theta[3, 1] = np.nan
    theta[4, 2] = np.nan
    theta[5, 3] = np.nan
    theta[6, 4] = np.nan
    theta[7, 3] = -400.0
    theta[8, 2] = 400.0
    theta[9, 0] = np.inf
    theta[10, 0] = -np.inf
    theta[11, :5] = [12.0, 0.0, 11.0, 13.0, 1.0]      # a step exactly on a bin edge
    theta[12, 2] = 15.5
    return theta


@pytest.mark.parametrize('variant', ['plain', 'modulate', 'assembias', 'assembias+modulate'])
@pytest.mark.parametrize('n_prim, n_sec, ragged', [(12, 2, False), (9, 3, False), (15, 2, True),
                                                   (7, 4, True), (30, 1, False)])
def test_grouped_occupations_equal_the_per_bin_loop_bit_for_bit(variant, n_prim, n_sec, ragged):
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = (ragged_table(n_prim, n_sec, 4, seed=n_prim) if ragged
             else synthetic.synthetic_table(n_prim, n_sec, (4, ), 'auto', seed=n_prim))
    n_draws = 200
    theta = degenerate_draws(synthetic.zheng07_draws(n_draws, seed=n_sec))
    kwargs = {}
    if 'modulate' in variant:
        kwargs['modulate_with_cenocc'] = True
    batch = theta
    strengths = None
    if 'assembias' in variant:
        rng = np.random.default_rng(5)
        strengths = rng.uniform(-1.2, 1.2, (n_draws, 2))
        strengths[20, 0] = np.nan
        strengths[21, 1] = np.nan
        batch = np.hstack([theta, strengths])
        kwargs['assembias'] = True
    halotab = make_tabcorr(table)
    with np.errstate(all='ignore'):
        grouped = halotab.mean_occupation_batch(batch, **kwargs)
        set_option(halotab, 'grouped', 0)
        per_bin = halotab.mean_occupation_batch(batch, **kwargs)
    assert grouped.tobytes() == per_bin.tobytes()
    # ... and the oracle on the regular draws
    oracle_kwargs = dict(kwargs)
    if strengths is not None:
        oracle_kwargs['assembias'] = strengths
    index = np.r_[0, 13:20, 22:40]
    for i in index:
        model = oracle.Zheng07(theta[i], kwargs.get('modulate_with_cenocc', False),
                               None if strengths is None else strengths[i])
        assert_rel(grouped[i], oracle.mean_occupation(table, model), RTOL, 'draw %d' % i)


@pytest.mark.parametrize('n_prim, n_sec, n_r, ragged, kwargs', [
    (50, 2, 19, False, {}),                          # BASELINE configs[2]'s table: 200 bins
    (50, 2, 19, False, {'assembias': True}),
    (25, 2, 7, False, {}),                           # 100 bins: 64-draw workgroups
    (25, 2, 7, False, {'assembias': True, 'modulate_with_cenocc': True}),
    (17, 3, 5, True, {'modulate_with_cenocc': True}),
    (11, 4, 3, True, {'assembias': True}),
])
def test_grouped_predictions(n_prim, n_sec, n_r, ragged, kwargs):
    """Three kernels, one launch with 64-draw and with 32-draw workgroups, total and separated by
    galaxy type: grouped against the per-bin loop (rounding of the sums over bins only) and
    against the oracle."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = (ragged_table(n_prim, n_sec, n_r, seed=n_prim) if ragged
             else synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim))
    n_draws = 333
    theta = synthetic.zheng07_draws(n_draws, seed=n_prim)
    kwargs = dict(kwargs)
    batch, strengths = theta, None
    if kwargs.get('assembias'):
        strengths = np.random.default_rng(n_r).uniform(-1.2, 1.2, (n_draws, 2))
        batch = np.hstack([theta, strengths])
    oracle_kwargs = dict(kwargs)
    if strengths is not None:
        oracle_kwargs['assembias'] = strengths
    n_bins = len(table['gal_type'])
    for separate in (False, True):
        expect = oracle.predict_zheng07_batch(table, theta, separate_gal_type=separate,
                                              **oracle_kwargs)
        results = {}
        for form in ('three kernels', 'one launch, 64 draws', 'one launch, 32 draws'):
            for grouped in (1, 0):
                halotab = make_tabcorr(table)
                set_option(halotab, 'grouped', grouped)
                set_option(halotab, 'single_draw', 0)
                set_option(halotab, 'fused_min_draws', 1)
                set_option(halotab, 'fused', 0 if form == 'three kernels' else 2)
                if form != 'three kernels':
                    set_option(halotab, 'fused_draws', 64 if '64' in form else 32)
                results[form, grouped] = halotab.predict_batch(
                    batch, separate_gal_type=separate, **kwargs)
                launch = last_launch(halotab)
                if form == 'three kernels' or (n_bins > 104 and '64' in form):
                    continue       # (tables beyond 104 bins have no 64-draw workgroups)
                assert launch[2] == 0, (form, launch)
                assert launch[0] == (n_draws + (63 if '64' in form else 31)) // (
                    64 if '64' in form else 32), (form, launch)
        for (form, grouped), (ngal, xi) in results.items():
            what = '%s, grouped=%d, separate=%s' % (form, grouped, separate)
            reference = results[form, 0]
            if separate:
                for key in expect[0]:
                    assert_rel(ngal[key], expect[0][key], RTOL, 'ngal ' + key + ' ' + what)
                    assert_rel(ngal[key], reference[0][key], 1e-13, what)
                for key in expect[1]:
                    assert_rel(xi[key], expect[1][key], RTOL, 'xi ' + key + ' ' + what,
                               floor=1e-13)
                    assert_rel(xi[key], reference[1][key], 1e-12, what, floor=1e-13)
            else:
                assert_rel(ngal, expect[0], RTOL, 'ngal ' + what)
                assert_rel(xi, expect[1], RTOL, 'xi ' + what)
                assert_rel(ngal, reference[0], 1e-13, what)
                assert_rel(xi, reference[1], 1e-12, what)


def test_grouped_real_abacus_table_in_mode_cross():
    """The reference's AbacusSummit fixture (tests/AbacusSummit/.../ds_efficient.hdf5: mode
    cross, 1104 bins = 560 groups of one or two percentile bins, real
    prim_haloprop_dist_index per bin): occupations bit for bit, predictions against the oracle."""
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    halotab = Interpolator.read(os.path.join(REPO, 'tests', 'golden',
                                             'ds_efficient.hdf5')).tabcorr_list[0]
    table = {'gal_type': halotab.gal_type.as_array(), 'tpcf_matrix': halotab.tpcf_matrix,
             'tpcf_shape': halotab.tpcf_shape, 'attrs': halotab.attrs}
    rng = np.random.default_rng(0)
    theta = synthetic.zheng07_draws(300, seed=2)
    theta[:, 0] = rng.uniform(12.5, 13.3, 300)          # (a galaxy sample this table resolves)
    theta[:, 3] = rng.uniform(13.6, 14.4, 300)
    strengths = rng.uniform(-1, 1, (300, 2))
    for kwargs, batch in (({}, theta), ({'assembias': True}, np.hstack([theta, strengths]))):
        set_option(halotab, 'grouped', 1)
        grouped = halotab.mean_occupation_batch(batch, **kwargs)
        ngal, xi = halotab.predict_batch(batch, **kwargs)
        set_option(halotab, 'grouped', 0)
        per_bin = halotab.mean_occupation_batch(batch, **kwargs)
        ngal0, xi0 = halotab.predict_batch(batch, **kwargs)
        assert grouped.tobytes() == per_bin.tobytes()
        assert_rel(ngal, ngal0, 1e-13)
        assert_rel(xi, xi0, 1e-12)
        oracle_kwargs = {'assembias': strengths[:40]} if kwargs else {}
        expect = oracle.predict_zheng07_batch(table, theta[:40], **oracle_kwargs)
        assert_rel(ngal[:40], expect[0], RTOL)
        assert_rel(xi[:40], expect[1], RTOL)


@pytest.mark.parametrize('shape, n_prim, n_sec, n_r, n_draws, separate', [
    ((4, 4), 12, 1, 7, 700, False),
    ((4, 4, 4), 6, 2, 19, 300, False),       # the database's grid shape, 64 tables
    ((5, ), 9, 1, 5, 4000, True),            # tables not a multiple of 8, separated
])
def test_interpolator_table_synchronous_schedule(shape, n_prim, n_sec, n_r, n_draws, separate):
    """hostmath.h: kQuadTableSync -- the waves of an XCD walk its tables one after the other
    (chosen for interpolators whose matrices do not fit the L2s side by side; forced here)
    against the oracle and the table-major order."""
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    tables, keys, points = synthetic.synthetic_interpolator(shape, n_prim, n_sec, (n_r, ), 'auto',
                                                            seed=n_prim)
    interp = Interpolator([make_tabcorr(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    theta = synthetic.zheng07_draws(n_draws, seed=n_r)
    rng = np.random.default_rng(n_draws)
    x = np.stack([rng.uniform(xp[0], xp[-1], n_draws) for xp in interp.xp], axis=-1)
    first = interp.to_device().tables[0]
    results = {}
    for order in (4, 1):
        from tabcorr_amd import _lib
        _lib.check(_lib.load().tc_table_set_option(first.handle, b'quad_order', order))
        results[order] = interp.predict_batch(theta, x, separate_gal_type=separate)
    setup = oracle.interpolator_setup(tables, points)
    index = np.r_[0:6, n_draws - 6:n_draws]
    expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[index], x[index],
                                                       separate_gal_type=separate)
    if separate:
        for key in expect[1]:
            assert_rel(results[4][1][key][index], expect[1][key], RTOL, key, floor=1e-12)
            assert_rel(results[4][1][key], results[1][1][key], 1e-11, key, floor=1e-12)
        for key in expect[0]:
            assert_rel(results[4][0][key][index], expect[0][key], RTOL, key)
    else:
        assert_rel(results[4][0][index], expect[0], RTOL)
        assert_rel(results[4][1][index], expect[1], RTOL, floor=1e-12)
        assert_rel(results[4][1], results[1][1], 1e-11, floor=1e-12)
