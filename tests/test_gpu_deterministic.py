"""Reproducible bits on request: option "deterministic" /
TabCorr.set_deterministic.  With level 2 a draw's (ngal, xi) depends on the draw alone -- one
kernel form per (table, flags) for every entry point and every batch size --, and by default
nothing that is measured at run time (timing, number of calls so far) enters the choice of a
form.  The reference is deterministic (tabcorr/tabcorr.py:580-683).  Needs an MI355X."""

import ctypes
import os

import numpy as np
import pytest

from util import assert_rel

pytestmark = pytest.mark.gpu

RTOL = 1e-10
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')


def make_tabcorr(table, **kwargs):
    from tabcorr_amd import TabCorr
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                               table['tpcf_shape'], table['attrs'], **kwargs)


def equal(a, b):
    if isinstance(a, dict):
        return all(np.array_equal(a[key], b[key], equal_nan=True) for key in a)
    return np.array_equal(a, b, equal_nan=True)


def pick(result, index):
    ngal, xi = result
    if isinstance(xi, dict):
        return ({k: v[index] for k, v in ngal.items()} if isinstance(ngal, dict) else ngal[index],
                {k: v[index] for k, v in xi.items()})
    return ngal[index], xi[index]


CASES = {
    # BASELINE configs[1]'s table: 8 waves x 64 draws, records + deferred pairs
    'cfg2': dict(table=lambda s: s.synthetic_table(50, 1, (19, ), 'auto', seed=0), kwargs={}),
    # the reference's example shape (G = 60), separated by galaxy type
    'g60 separated': dict(table=lambda s: s.synthetic_table(30, 1, (19, ), 'auto', seed=1),
                          kwargs={'separate_gal_type': True}),
    # BASELINE configs[2]: assembly bias, 200 bins (32-draw workgroups: the wide form)
    'cfg3': dict(table=lambda s: s.synthetic_table(50, 2, (19, ), 'auto', seed=2),
                 kwargs={'separate_gal_type': True, 'assembias': True}),
    # mode cross (a Delta Sigma table): the one-launch kernel of mode cross
    'cross': dict(table=lambda s: s.synthetic_table(40, 1, (13, ), 'cross', seed=3), kwargs={}),
}


@pytest.mark.parametrize('case', list(CASES))
def test_a_draws_bits_depend_on_the_draw_alone(case):
    """Batch sizes 1, 63, 777, 4096 and 10^4, draws at arbitrary places of their batch, through
    predict_batch (host arrays), predict_batch_async, the un-batched predict(model) and 600
    pipelined device-pointer calls: bit for bit what the 10^4-draw batch gave for that draw, and
    the oracle's values to 1e-10."""
    from tabcorr_amd import synthetic, _lib, Zheng07Model
    from oracle import tabcorr_oracle as oracle
    lib = _lib.load()
    spec = CASES[case]
    table = spec['table'](synthetic)
    kwargs = spec['kwargs']
    assembias = kwargs.get('assembias', False)
    halotab = make_tabcorr(table)
    assert halotab.set_deterministic(True) or kwargs, 'no batch-invariant form for this table'
    handle = halotab.to_device().handle
    from tabcorr_amd.tabcorr import _flags
    flags = _flags(kwargs.get('separate_gal_type', False), False, assembias, 'zheng07')
    out = ctypes.c_int(0)
    _lib.check(lib.tc_table_batch_invariant(handle, 10, flags, ctypes.byref(out)))
    assert out.value == 1, 'no batch-invariant form for these flags'
    theta = synthetic.zheng07_draws(10000, seed=11)
    if assembias:
        rng = np.random.default_rng(12)
        theta = np.column_stack([theta, rng.uniform(-1, 1, (10000, 2))])
    whole = halotab.predict_batch(theta, **kwargs)
    rng = np.random.default_rng(13)
    for n in (1, 63, 777, 4096):
        index = rng.choice(10000, n, replace=False)
        got = halotab.predict_batch(theta[index], **kwargs)
        assert equal(got[0], pick(whole, index)[0]), (case, n, 'ngal')
        assert equal(got[1], pick(whole, index)[1]), (case, n, 'xi')
        pending = halotab.predict_batch_async(theta[index], **kwargs)
        got = pending.wait()
        assert equal(got[0], pick(whole, index)[0]), (case, n, 'async ngal')
        assert equal(got[1], pick(whole, index)[1]), (case, n, 'async xi')
    # against the oracle (a sample: the oracle takes a few ms per draw)
    sample = rng.choice(10000, 48, replace=False)
    expect = oracle.predict_zheng07_batch(
        table, theta[sample, :5], separate_gal_type=kwargs.get('separate_gal_type', False),
        assembias=theta[sample, 5:] if assembias else None)
    got = pick(whole, sample)
    if isinstance(got[1], dict):
        scale = np.max(np.abs(sum(expect[1].values())))
        for key in got[1]:
            np.testing.assert_allclose(got[1][key], expect[1][key], rtol=RTOL, atol=1e-12 * scale)
    else:
        assert_rel(got[0], expect[0], RTOL)
        assert_rel(got[1], expect[1], RTOL)
    if not kwargs:
        # the reference's usage: one predict(model) per step
        model = Zheng07Model(redshift=table['attrs']['redshift'])
        for i in (0, 5000, 9999):
            for key, value in zip(KEYS, theta[i]):
                model.param_dict[key] = value
            ngal, xi = halotab.predict(model)
            assert ngal == whole[0][i] and np.array_equal(xi, whole[1][i]), (case, i)
        # 600 pipelined device-pointer calls of changing sizes, results resident in HBM (the
        # calls of a handle complete in any order: every call of a group of eight gets its
        # own place for its results)
        n_r = whole[1].shape[1]
        slots = 8
        pointers = [ctypes.c_void_p() for _ in range(3)]
        for ptr, count in zip(pointers, (theta.size, slots * 10000, slots * 10000 * n_r)):
            _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        d_theta, d_ngal, d_xi = pointers
        _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p),
                                     theta.nbytes))
        ngal = np.empty(slots * 10000)
        xi = np.empty((slots * 10000, n_r))
        sizes = (63, 10000, 777, 4096, 1, 2500)
        group = []
        for call in range(600):
            n = sizes[call % len(sizes)]
            first = (call * 37) % (10000 - n + 1)
            slot = call % slots
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, ctypes.c_void_p(d_theta.value + first * theta.shape[1] * 8),
                theta.shape[1], n, 10, 0, ctypes.c_void_p(d_ngal.value + slot * 10000 * 8),
                ctypes.c_void_p(d_xi.value + slot * 10000 * n_r * 8)))
            group.append((slot, first, n))
            if len(group) == slots:
                _lib.check(lib.tc_table_synchronize(handle))
                _lib.check(lib.tc_memcpy_d2h(ngal.ctypes.data_as(ctypes.c_void_p), d_ngal,
                                             ngal.nbytes))
                _lib.check(lib.tc_memcpy_d2h(xi.ctypes.data_as(ctypes.c_void_p), d_xi, xi.nbytes))
                for slot, first, n in group:
                    begin = slot * 10000
                    assert np.array_equal(ngal[begin:begin + n], whole[0][first:first + n]), call
                    assert np.array_equal(xi[begin:begin + n], whole[1][first:first + n]), call
                group = []
        _lib.check(lib.tc_table_synchronize(handle))
        for ptr in pointers:
            _lib.check(lib.tc_device_free(ptr))


def test_the_reference_interpolator_fixture_is_batch_invariant_on_request():
    """The reference's own AbacusSummit interpolator (mode cross, four tables): with
    set_deterministic a draw gives the same bits alone, in a small batch and in a large one."""
    from tabcorr_amd import Interpolator, synthetic
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    assert interp.set_deterministic(True)
    from util import load_golden
    rng = np.random.default_rng(3)
    golden = load_golden('ds_efficient')
    # (draws around the fixture's own: its haloes are massive)
    theta = golden['theta'][rng.integers(0, len(golden['theta']), 5000)]
    theta = theta + rng.normal(0.0, 0.05, theta.shape)
    theta[:, 1] = np.abs(theta[:, 1]) + 0.05
    x = np.stack([rng.uniform(xp[0], xp[-1], 5000) for xp in interp.xp], axis=-1)
    whole = interp.predict_batch(theta, x)
    assert np.all(np.isfinite(whole[1]))
    for n in (1, 63, 777):
        index = rng.choice(5000, n, replace=False)
        got = interp.predict_batch(theta[index], x[index])
        assert np.array_equal(got[0], whole[0][index]), n
        assert np.array_equal(got[1], whole[1][index]), n


def test_measured_dispatch_is_refused_and_default_is_timing_free():
    """Level 1 and 2 refuse "autotune" / "autotune_after"; the default never measures by itself
    (option "autotune_after" = 0): 600 pipelined calls leave no measured choice behind, and the
    same sequence of calls gives the same bits on a second handle."""
    from tabcorr_amd import synthetic, _lib
    lib = _lib.load()
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(6000, seed=5)
    results = []
    for run in range(2):
        halotab = make_tabcorr(table)
        handle = halotab.to_device().handle
        pending = [halotab.predict_batch_async(theta[:n]) for n in (3000, 6000, 500) * 100]
        got = [p.wait() for p in pending[-3:]]
        assert halotab.autotune(measure=False) is None, 'the library measured by itself'
        results.append([halotab.predict_batch(theta[:n]) for n in (1, 40, 3000, 6000)] + got)
    for a, b in zip(*results):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    halotab.set_deterministic(1)
    for name, value in ((b'autotune', 0), (b'autotune_after', 256)):
        assert lib.tc_table_set_option(handle, name, value) != 0
    _lib.check(lib.tc_table_set_option(handle, b'autotune_after', 0))
    halotab.set_deterministic(False)
    _lib.check(lib.tc_table_set_option(handle, b'autotune_after', 256))


def test_tables_without_a_batch_invariant_form_say_so():
    """set_deterministic(True) returns False for a table no one-launch form serves (float32
    storage): its calls then run as with level 1 -- reproducible from run to run, not
    batch-invariant -- and tc_table_batch_invariant reports it per combination of flags."""
    from tabcorr_amd import synthetic, _lib
    from tabcorr_amd.tabcorr import _flags
    lib = _lib.load()
    table = synthetic.synthetic_table(12, 1, (6, 5), 'auto', seed=3)
    halotab = make_tabcorr(table, compute_dtype='float32')
    assert halotab.set_deterministic(True) is False
    theta = synthetic.zheng07_draws(300, seed=4)
    first = halotab.predict_batch(theta)
    again = halotab.predict_batch(theta)
    assert np.array_equal(first[1], again[1])
    wide = make_tabcorr(synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0))
    assert wide.set_deterministic(True) is True
    out = ctypes.c_int(-1)
    # the fused likelihood of separated components does not exist: no form for that combination
    for separate, expect in ((False, 1), (True, 1)):
        _lib.check(lib.tc_table_batch_invariant(wide.to_device().handle, 10,
                                                _flags(separate, False, False, 'zheng07'),
                                                ctypes.byref(out)))
        assert out.value == expect
    _lib.check(lib.tc_table_batch_invariant(wide.to_device().handle, 7, 0, ctypes.byref(out)))
    assert out.value == 1                    # (any n_gauss_prim for the undecorated family)
    assert wide.set_deterministic(False) is False
