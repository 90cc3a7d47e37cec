"""BASELINE.json configurations at full size on the GPU: spot checks against the CPU
oracle on a few draws plus size-independent properties (components sum to the total,
linearity in the table, interpolation reproduces the tables at grid nodes, batch
splitting / ordering invariance)."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from util import assert_rel  # noqa: E402

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def make(table, **kwargs):
    from tabcorr_amd import TabCorr
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                               table['tpcf_shape'], table['attrs'], **kwargs)


def oracle_check(table, theta, ngal, xi, index, **kwargs):
    from oracle import tabcorr_oracle as oracle
    expect = oracle.predict_zheng07_batch(table, theta[index], **kwargs)
    assert_rel(ngal[index], expect[0], RTOL)
    assert_rel(xi[index], expect[1], RTOL)


def test_config2_batch_of_1e4():
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(10000, seed=1)
    halotab = make(table)
    ngal, xi = halotab.predict_batch(theta)
    assert ngal.shape == (10000, ) and xi.shape == (10000, 19)
    assert np.all(np.isfinite(xi)) and np.all(ngal > 0)
    index = np.r_[0:6, 4093:4099, 9994:10000]
    oracle_check(table, theta, ngal, xi, index)

    ngal_sep, xi_sep = halotab.predict_batch(theta, separate_gal_type=True)
    assert_rel(sum(ngal_sep.values()), ngal, 1e-13)
    assert_rel(sum(xi_sep.values()), xi, 1e-12)

    # deterministic: same bits on a second call.  A draw's position in the batch decides
    # where the equal-share schedule cuts its sums (hostmath.h: QuadSchedule): reordering /
    # splitting changes last bits of xi only; the number densities -- per draw arithmetic, also
    # with the moment expansions of csrc/series.h, which take the terms each draw needs -- not
    ngal2, xi2 = halotab.predict_batch(theta)
    assert np.array_equal(xi, xi2) and np.array_equal(ngal, ngal2)
    perm = np.random.default_rng(0).permutation(len(theta))
    ngal3, xi3 = halotab.predict_batch(theta[perm])
    assert_rel(ngal3, ngal[perm], 1e-14)
    assert_rel(xi3, xi[perm], 1e-13)
    # ... a draw's number densities do not depend on its neighbours, expansions off or on
    from tabcorr_amd import _lib
    _lib.check(_lib.load().tc_table_set_option(halotab.to_device().handle, b'series', 0))
    assert np.array_equal(halotab.predict_batch(theta[perm])[0],
                          halotab.predict_batch(theta)[0][perm])
    _lib.check(_lib.load().tc_table_set_option(halotab.to_device().handle, b'series', 3))
    assert np.array_equal(halotab.predict_batch(theta[perm])[0],
                          halotab.predict_batch(theta)[0][perm])
    ngal4, xi4 = halotab.predict_batch(theta[:777])
    assert_rel(xi4, xi[:777], 1e-13)

    # linearity in the table: xi is linear in tpcf_matrix, ngal independent of it
    doubled = dict(table)
    doubled['tpcf_matrix'] = 2.0 * table['tpcf_matrix']
    ngal5, xi5 = make(doubled).predict_batch(theta)
    assert np.array_equal(ngal5, ngal)
    assert np.array_equal(xi5, 2.0 * xi)      # same decomposition: bit-exact
    ngal6, xi6 = make(doubled).predict_batch(theta[:2000])
    assert_rel(ngal6, ngal[:2000], 1e-14)     # other batch size: other summation order
    assert_rel(xi6, 2.0 * xi[:2000], 1e-13)


def test_config3_separate_gal_type_assembias():
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 2, (19, ), 'auto', seed=3)
    theta = synthetic.zheng07_draws(10000, seed=2)
    rng = np.random.default_rng(4)
    theta7 = np.hstack([theta, rng.uniform(-1, 1, size=(len(theta), 2))])
    halotab = make(table)
    ngal, xi = halotab.predict_batch(theta7, assembias=True)
    ngal_sep, xi_sep = halotab.predict_batch(theta7, assembias=True,
                                             separate_gal_type=True)
    assert list(xi_sep) == ['centrals-centrals', 'centrals-satellites',
                            'satellites-satellites']
    assert_rel(sum(ngal_sep.values()), ngal, 1e-13)
    assert_rel(sum(xi_sep.values()), xi, 1e-12)
    index = np.r_[0:4, 5000:5003, 9997:10000]
    expect = oracle.predict_zheng07_batch(table, theta[index],
                                          assembias=theta7[index, 5:])
    assert_rel(ngal[index], expect[0], RTOL)
    assert_rel(xi[index], expect[1], RTOL)
    expect = oracle.predict_zheng07_batch(
        table, theta[index], assembias=theta7[index, 5:],
        separate_gal_type=True)
    for key in xi_sep:
        assert_rel(xi_sep[key][index], expect[1][key], RTOL, key)
    # zero assembly bias strength = plain Zheng07
    plain = halotab.predict_batch(theta[:500])
    zero = halotab.predict_batch(np.hstack([theta[:500], np.zeros((500, 2))]),
                                 assembias=True)
    assert_rel(zero[1], plain[1], 1e-13)


def test_config4_interpolator_5x5():
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    tables, keys, points = synthetic.synthetic_interpolator(
        (5, 5), 50, 1, (19, ), 'auto', seed=7)
    halotabs = [make(t) for t in tables]
    interp = Interpolator(halotabs, {k: points[:, d] for d, k in
                                     enumerate(keys)})
    n_draws = 12500          # one GPU's share of the 10^5 draws of config 4
    theta = synthetic.zheng07_draws(n_draws, seed=5)
    rng = np.random.default_rng(6)
    x = np.stack([rng.uniform(xp[0], xp[-1], size=n_draws)
                  for xp in interp.xp], axis=-1)
    # the first 25 draws sit exactly on the grid nodes
    x[:25] = points
    ngal, xi = interp.predict_batch(theta, x)
    assert xi.shape == (n_draws, 19) and np.all(np.isfinite(xi))
    for k in [0, 7, 24]:
        single = halotabs[k].predict_batch(theta[k:k + 1])
        assert_rel(ngal[k], single[0][0], 1e-11)
        assert_rel(xi[k], single[1][0], 1e-11)
    setup = oracle.interpolator_setup(tables, points)
    index = [30, 6000, n_draws - 1]
    expect = oracle.interpolator_predict_zheng07_batch(
        tables, setup, theta[index], x[index])
    assert_rel(ngal[index], expect[0], RTOL)
    assert_rel(xi[index], expect[1], RTOL)
    ngal_sep, xi_sep = interp.predict_batch(theta[:3000], x[:3000],
                                            separate_gal_type=True)
    assert_rel(sum(xi_sep.values()), xi[:3000], 1e-11)


# draws the oracle checks in a batch of 10^4: both ends, the boundaries of the 32- and 64-draw
# tiles of the contraction kernels, the middle, and the last (partly filled) tile
BENCH_DRAWS = 10000
BENCH_INDEX = np.r_[0:2, 31:34, 63:66, 4999:5001, 9983:9985, 9998:10000]


def test_config5_rp_pi_table_float64():
    """AbacusSummit-scale table at the batch size bench.py times (BASELINE configs[4] in
    float64): 100 mass bins x {cen, sat}, tpcf_shape (19, 40); R = 760 exercises the r
    tiling, 10^4 draws the schedule of the benchmark."""
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9)
    theta = synthetic.zheng07_draws(BENCH_DRAWS, seed=8)
    halotab = make(table)
    ngal, xi = halotab.predict_batch(theta)
    assert xi.shape == (BENCH_DRAWS, 19, 40)
    oracle_check(table, theta, ngal, xi, BENCH_INDEX)
    ngal_sep, xi_sep = halotab.predict_batch(theta[:512], separate_gal_type=True)
    assert_rel(sum(xi_sep.values()), xi[:512], 1e-12)


def test_many_bins_auto_mode():
    """More bins than one LDS staging can hold at once: the plan cuts the pair
    triangle into segments."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(220, 1, (3, ), 'auto', seed=11)   # G = 440
    theta = synthetic.zheng07_draws(130, seed=12)
    halotab = make(table)
    ngal, xi = halotab.predict_batch(theta)
    expect = oracle.predict_zheng07_batch(table, theta[:3])
    assert_rel(ngal[:3], expect[0], RTOL)
    assert_rel(xi[:3], expect[1], RTOL)
    ngal_sep, xi_sep = halotab.predict_batch(theta, separate_gal_type=True)
    assert_rel(sum(xi_sep.values()), xi, 1e-12)
    expect = oracle.predict_zheng07_batch(table, theta[:2],
                                          separate_gal_type=True)
    for key in xi_sep:
        assert_rel(xi_sep[key][:2], expect[1][key], RTOL, key)


def test_many_bins_cross_mode():
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(552, 1, (13, ), 'cross', seed=13)  # G = 1104
    theta = synthetic.zheng07_draws(70, seed=14)
    halotab = make(table)
    ngal, xi = halotab.predict_batch(theta)
    expect = oracle.predict_zheng07_batch(table, theta[:4])
    assert_rel(ngal[:4], expect[0], RTOL)
    assert_rel(xi[:4], expect[1], RTOL)


# north_star / BASELINE configs[4]: float32 path, stated tolerance 1e-5 relative
RTOL_F32 = 1e-5


@pytest.mark.parametrize('case', [
    dict(n_prim=12, n_sec=1, tpcf_shape=(7, 9), mode='auto', seed=41),
    dict(n_prim=9, n_sec=2, tpcf_shape=(40, ), mode='cross', seed=42),
    dict(n_prim=50, n_sec=1, tpcf_shape=(19, ), mode='auto', seed=0),
    dict(n_prim=70, n_sec=2, tpcf_shape=(3, ), mode='auto', seed=43)])
def test_float32_variant_small(case):
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(**case)
    theta = synthetic.zheng07_draws(200, seed=case['seed'] + 1)
    halotab = make(table, compute_dtype='float32')
    ngal, xi = halotab.predict_batch(theta)
    expect = oracle.predict_zheng07_batch(table, theta[:40])
    assert_rel(ngal[:40], expect[0], 1e-12)          # occupations stay float64
    assert_rel(xi[:40], expect[1], RTOL_F32)
    ngal_sep, xi_sep = halotab.predict_batch(theta, separate_gal_type=True)
    assert_rel(sum(xi_sep.values()), xi, RTOL_F32)
    expect = oracle.predict_zheng07_batch(table, theta[:10],
                                          separate_gal_type=True)
    for key in xi_sep:
        scale = np.max(np.abs(expect[1][key]))
        np.testing.assert_allclose(xi_sep[key][:10], expect[1][key],
                                   rtol=RTOL_F32, atol=RTOL_F32 * scale)
    # the ndarray seam (tabcorr.py:616-621) and ragged batch sizes through the float32 path
    occupation = halotab.mean_occupation_batch(theta[:70])
    ngal_occ, xi_occ = halotab.predict(occupation)
    assert_rel(xi_occ, xi[:70], 1e-6)
    for n_draws in (1, 63, 65):
        ngal_n, xi_n = halotab.predict_batch(theta[:n_draws])
        assert_rel(xi_n, xi[:n_draws], 1e-6)


def test_config5_float32_mfma():
    """BASELINE configs[4] at full size AND at the batch size bench.py times (10^4 draws: 157
    draw tiles of 64, the equal-share schedule of the benchmark): 100 mass bins x {cen, sat},
    tpcf_shape (19, 40), float32 table and accumulation on the matrix cores -- against the
    oracle on draws at both ends, across tile boundaries and in the last tile, and against
    the float64 kernel on all draws."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9)
    theta = synthetic.zheng07_draws(BENCH_DRAWS, seed=8)
    halotab32 = make(table, compute_dtype='float32')
    ngal, xi = halotab32.predict_batch(theta)
    assert xi.shape == (BENCH_DRAWS, 19, 40)
    expect = oracle.predict_zheng07_batch(table, theta[BENCH_INDEX])
    assert_rel(ngal[BENCH_INDEX], expect[0], 1e-12)
    assert_rel(xi[BENCH_INDEX], expect[1], RTOL_F32)
    # against the float64 kernel on all draws
    ngal64, xi64 = make(table).predict_batch(theta)
    assert_rel(ngal, ngal64, 1e-12)
    assert_rel(xi, xi64, RTOL_F32)
    assert np.max(np.abs(xi / xi64 - 1)) < RTOL_F32


def test_configs_1_to_3_at_the_benchmarked_batch_size():
    """The batch bench.py times (10^4 draws per call; 12 500 for the interpolator share)
    against the oracle at both ends, across tile boundaries and in the last tile:
    configs[1] total, configs[2] separated with assembly bias, configs[3] interpolated."""
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    theta = synthetic.zheng07_draws(BENCH_DRAWS, seed=1)
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    ngal, xi = make(table).predict_batch(theta)
    oracle_check(table, theta, ngal, xi, BENCH_INDEX)
    table3 = synthetic.synthetic_table(50, 2, (19, ), 'auto', seed=3)
    theta7 = np.hstack([theta, np.random.default_rng(0).uniform(-1, 1, (BENCH_DRAWS, 2))])
    ngal_sep, xi_sep = make(table3).predict_batch(theta7, separate_gal_type=True,
                                                  assembias=True)
    expect = oracle.predict_zheng07_batch(table3, theta7[BENCH_INDEX, :5],
                                          separate_gal_type=True,
                                          assembias=theta7[BENCH_INDEX, 5:])
    for key in expect[0]:
        assert_rel(ngal_sep[key][BENCH_INDEX], expect[0][key], RTOL, key)
    for key in expect[1]:
        assert_rel(xi_sep[key][BENCH_INDEX], expect[1][key], RTOL, key)
    tables, keys, points = synthetic.synthetic_interpolator((5, 5), 50, 1, (19, ), 'auto',
                                                            seed=7)
    interp = Interpolator([make(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    n4 = 12500
    theta4 = synthetic.zheng07_draws(n4, seed=5)
    rng = np.random.default_rng(6)
    x4 = np.stack([rng.uniform(xp[0], xp[-1], size=n4) for xp in interp.xp], axis=-1)
    ngal4, xi4 = interp.predict_batch(theta4, x4)
    index = np.r_[0:2, 31:33, 6249:6251, n4 - 2:n4]
    setup = oracle.interpolator_setup(tables, points)
    expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta4[index], x4[index])
    assert_rel(ngal4[index], expect[0], RTOL)
    assert_rel(xi4[index], expect[1], RTOL, floor=1e-12)


@pytest.mark.parametrize('shape,tpcf_shape,mode', [((4, 4), (40, ), 'auto'),
                                                   ((5, ), (7, 9), 'cross'),
                                                   ((4, 5), (19, ), 'auto')])
def test_float32_interpolator(shape, tpcf_shape, mode):
    """Interpolation over float32 tables (matrix-core kernel with the table loop) against
    the float64 interpolator and, for a few draws, the oracle."""
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    tables, keys, points = synthetic.synthetic_interpolator(
        shape, 14, 1, tpcf_shape, mode, seed=61)
    param_dict = {key: points[:, d] for d, key in enumerate(keys)}
    interp64 = Interpolator([make(t) for t in tables], param_dict)
    interp32 = Interpolator([make(t, compute_dtype='float32') for t in tables], param_dict)
    rng = np.random.default_rng(62)
    for n_draws in (3, 700):
        theta = synthetic.zheng07_draws(n_draws, seed=63)
        x = np.stack([rng.uniform(xp[0], xp[-1], size=n_draws) for xp in interp64.xp], axis=-1)
        ngal64, xi64 = interp64.predict_batch(theta, x)
        ngal32, xi32 = interp32.predict_batch(theta, x)
        assert_rel(ngal32, ngal64, 1e-12)
        scale = np.max(np.abs(xi64), axis=tuple(range(1, xi64.ndim)), keepdims=True)
        np.testing.assert_allclose(xi32, xi64, rtol=RTOL_F32, atol=RTOL_F32 * scale.max())
        sep64 = interp64.predict_batch(theta, x, separate_gal_type=True)[1]
        sep32 = interp32.predict_batch(theta, x, separate_gal_type=True)[1]
        for key in sep64:
            np.testing.assert_allclose(sep32[key], sep64[key], rtol=RTOL_F32,
                                       atol=RTOL_F32 * np.max(np.abs(sep64[key])))
    # the oracle on 48 draws spread over the batch (first and last tiles included)
    setup = oracle.interpolator_setup(tables, points)
    index = np.r_[0:16, 342:358, 684:700]
    expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[index], x[index])
    assert_rel(ngal32[index], expect[0], 1e-11)
    np.testing.assert_allclose(xi32[index], expect[1], rtol=RTOL_F32,
                               atol=RTOL_F32 * np.max(np.abs(expect[1])))
    # a NaN coordinate with extrapolate=True: np.digitize puts it past the last node, the
    # clamp keeps the segment, the polynomial is NaN (interpolator.py:318-329) -- that draw only
    x_nan = x[:6].copy()
    x_nan[2, 0] = np.nan
    for interp in (interp64, interp32):
        ngal_nan, xi_nan = interp.predict_batch(theta[:6], x_nan, extrapolate=True)
        expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[:6], x_nan,
                                                           extrapolate=True)
        assert np.isnan(expect[0][2]) and np.all(np.isnan(expect[1][2]))
        assert np.isnan(ngal_nan[2]) and np.all(np.isnan(xi_nan[2]))
        keep = [0, 1, 3, 4, 5]
        np.testing.assert_allclose(xi_nan[keep], expect[1][keep], rtol=RTOL_F32,
                                   atol=RTOL_F32 * np.max(np.abs(expect[1][keep])))
    with pytest.raises(ValueError):
        interp64.predict_batch(theta[:6], x_nan)


def test_more_draws_than_one_slab():
    """Batches beyond the internal slab size (2^18 draws) are processed in pieces."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(6, 1, (5, ), 'auto', seed=21)
    n_draws = (1 << 18) + 1000
    theta = synthetic.zheng07_draws(n_draws, seed=22)
    halotab = make(table)
    ngal, xi = halotab.predict_batch(theta)
    assert xi.shape == (n_draws, 5) and np.all(np.isfinite(xi))
    index = np.r_[0:3, (1 << 18) - 2:(1 << 18) + 3, n_draws - 3:n_draws]
    oracle_check(table, theta, ngal, xi, index)
    ngal_sep, xi_sep = halotab.predict_batch(theta, separate_gal_type=True)
    assert_rel(sum(xi_sep.values()), xi, 1e-12)


def test_degenerate_parameters_do_not_fault():
    """Non-finite or absurd parameters give non-finite results, not a device fault."""
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(8, 2, (4, ), 'auto', seed=23)
    halotab = make(table)
    theta = synthetic.zheng07_draws(64, seed=24)
    theta[0, 1] = 0.0            # sigma_logM = 0
    theta[1, 0] = np.nan
    theta[2, 4] = np.inf
    theta[3, 3] = -400.0         # M1 underflows
    theta[4, 2] = 400.0          # M0 overflows
    theta[5] = 0.0
    ngal, xi = halotab.predict_batch(theta)
    good = np.arange(6, 64)
    assert np.all(np.isfinite(xi[good])) and np.all(ngal[good] > 0)
    # the device is still healthy afterwards
    ngal2, xi2 = halotab.predict_batch(theta[good])
    assert_rel(xi2, xi[good], 1e-13)


def _occupation_oracle(table, theta, **kwargs):
    from oracle import tabcorr_oracle as oracle
    out = []
    with np.errstate(all='ignore'):
        for i, t in enumerate(theta):
            assembias = kwargs.get('assembias')
            model = oracle.Zheng07(t[:5], kwargs.get('modulate_with_cenocc', False),
                                   None if assembias is None else assembias[i])
            out.append(oracle.mean_occupation(table, model))
    return np.array(out)


@pytest.mark.parametrize('variant', ['plain', 'modulate', 'assembias'])
def test_degenerate_parameters_match_the_oracle(variant):
    """NaN parameters, sigma_logM = 0, all satellites below M0 and draws without any
    galaxies give what the reference's NumPy arithmetic gives (NaN where it gives NaN:
    an MCMC likelihood relies on that to reject the draw) -- batched and un-batched."""
    import warnings
    from tabcorr_amd import synthetic, Zheng07Model
    from tabcorr_amd.models import ASSEMBIAS_KEYS, ZHENG07_KEYS
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(8, 2, (4, ), 'auto', seed=23)
    halotab = make(table)
    base = np.array([12.4, 0.35, 11.9, 13.3, 1.1])
    rows, names = [], []
    def case(name, **changes):
        theta = base.copy()
        for key, value in changes.items():
            theta[ZHENG07_KEYS.index(key)] = value
        rows.append(theta)
        names.append(name)
    case('regular')
    case('sigma = 0: step function', sigma_logM=0.0)
    case('sigma = -0', sigma_logM=-0.0)
    case('sigma tiny', sigma_logM=1e-12)
    case('NaN logMmin', logMmin=np.nan)
    case('NaN sigma', sigma_logM=np.nan)
    case('NaN logM0: no satellites', logM0=np.nan)
    case('NaN logM1', logM1=np.nan)
    case('NaN alpha', alpha=np.nan)
    case('all NaN', logMmin=np.nan, sigma_logM=np.nan, logM0=np.nan, logM1=np.nan,
         alpha=np.nan)
    case('every satellite bin below M0', logM0=15.5)
    case('NaN alpha with every satellite bin below M0', logM0=15.5, alpha=np.nan)
    case('no galaxies at all: 0 / 0', logMmin=30.0, sigma_logM=0.01, logM0=15.5)
    case('no centrals', logMmin=30.0, sigma_logM=0.01)
    case('logMmin = +inf', logMmin=np.inf)
    case('logMmin = -inf', logMmin=-np.inf)
    case('M1 underflows', logM1=-400.0)
    case('M0 overflows', logM0=400.0)
    theta = np.array(rows)
    kwargs = {}
    if variant == 'modulate':
        kwargs['modulate_with_cenocc'] = True
    if variant == 'assembias':
        strengths = np.tile([[0.6, -0.4]], (len(theta), 1))
        strengths[0] = [1.8, -3.0]              # clipped to [-1, 1]
        strengths[3] = [np.nan, 0.2]            # NaN strength: every central bin NaN
        strengths[6] = [0.2, np.nan]
        kwargs['assembias'] = strengths
    with np.errstate(all='ignore'), warnings.catch_warnings():
        warnings.simplefilter('ignore')
        expect_occ = _occupation_oracle(table, theta, **kwargs)
        expect = oracle.predict_zheng07_batch(table, theta, **kwargs)
        expect_sep = oracle.predict_zheng07_batch(table, theta, separate_gal_type=True,
                                                  **kwargs)
    call = dict(kwargs)
    batch = theta
    if variant == 'assembias':
        call['assembias'] = True
        batch = np.hstack([theta, strengths])
    occupation = halotab.mean_occupation_batch(batch, **call)
    ngal, xi = halotab.predict_batch(batch, **call)
    ngal_sep, xi_sep = halotab.predict_batch(batch, separate_gal_type=True, **call)
    for i, name in enumerate(names):
        what = '%s / %s' % (variant, name)
        assert np.array_equal(np.isnan(occupation[i]), np.isnan(expect_occ[i])), what
        assert_rel(np.nan_to_num(occupation[i]), np.nan_to_num(expect_occ[i]), RTOL, what)
        np.testing.assert_allclose(ngal[i], expect[0][i], rtol=RTOL, err_msg=what)
        np.testing.assert_allclose(xi[i], expect[1][i], rtol=RTOL, err_msg=what)
        for key in ngal_sep:
            np.testing.assert_allclose(ngal_sep[key][i], expect_sep[0][key][i], rtol=RTOL,
                                       err_msg=what)
        for key in xi_sep:
            np.testing.assert_allclose(xi_sep[key][i], expect_sep[1][key][i], rtol=RTOL,
                                       atol=1e-14 * np.nanmax(np.abs(np.nan_to_num(
                                           expect[1][i]))), err_msg=what)
    # the step function really is one, and the rejected draws are NaN
    cen = oracle.is_centrals(table['gal_type'])
    if variant == 'plain':
        assert np.all((occupation[1][cen] >= 0) & (occupation[1][cen] <= 1 + 1e-15))
        assert np.all(np.isnan(xi[4])) and np.isnan(ngal[4])
        assert np.all(np.isnan(xi[12]))          # 0 / 0
        assert ngal[12] == 0.0
    # un-batched calls (one launch for one draw) agree with the batched path bit for bit
    # in their NaN pattern and to rounding elsewhere
    for i, name in enumerate(names):
        model = Zheng07Model(modulate_with_cenocc=variant == 'modulate',
                             sec_haloprop_key='halo_nfw_conc' if variant == 'assembias'
                             else None)
        for k, key in enumerate(ZHENG07_KEYS):
            model.param_dict[key] = theta[i, k]
        if variant == 'assembias':
            model.param_dict[ASSEMBIAS_KEYS[0]] = strengths[i, 0]
            model.param_dict[ASSEMBIAS_KEYS[1]] = strengths[i, 1]
        n1, x1 = halotab.predict(model, check_consistency=False)
        np.testing.assert_allclose(n1, ngal[i], rtol=1e-12, err_msg=name)
        np.testing.assert_allclose(x1, xi[i], rtol=1e-12, err_msg=name)


def test_sigma_zero_tie_is_nan():
    """sigma_logM = 0 and logMmin exactly on a quadrature node: the reference divides
    0 by 0 there (tabcorr.py:556-559 -> erf(nan)), so that bin is NaN."""
    import ctypes
    from tabcorr_amd import synthetic, _lib
    table = synthetic.synthetic_table(6, 1, (3, ), 'auto', seed=29)
    halotab = make(table)
    lib = _lib.load()
    x = np.zeros(10)
    w = np.zeros(10)
    _lib.check(lib.tc_gauss_legendre(10, _lib.as_double_p(x), _lib.as_double_p(w)))
    gal_type = table['gal_type']
    lo, hi = gal_type['log_prim_haloprop_min'][2], gal_type['log_prim_haloprop_max'][2]
    node = np.log10(10.0**(lo + (hi - lo) * x[4]))     # as the library forms it
    theta = np.array([[node, 0.0, 11.9, 13.3, 1.1], [node + 1e-9, 0.0, 11.9, 13.3, 1.1]])
    occupation = halotab.mean_occupation_batch(theta)
    assert np.isnan(occupation[0, 2]) and np.sum(np.isnan(occupation[0])) == 1
    assert not np.any(np.isnan(occupation[1]))
    ngal, xi = halotab.predict_batch(theta)
    assert np.isnan(ngal[0]) and np.all(np.isnan(xi[0]))
    assert np.all(np.isfinite(xi[1]))
    occupation = halotab.mean_occupation_batch(theta, modulate_with_cenocc=True)
    assert np.isnan(occupation[0, 2])


def test_many_batch_sizes_do_not_grow_the_schedule_cache_without_bound():
    """A caller sweeping the batch size: one cached schedule per distinct number of draw tiles,
    capped (the cache is flushed behind a device synchronisation); results stay right across
    the flush and device memory does not keep growing."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(6, 1, (5, ), 'auto', seed=4)
    halotab = make(table)
    theta = synthetic.zheng07_draws(32 * 150, seed=2)
    expect = oracle.predict_zheng07_batch(table, theta[:40])
    for n_tiles in range(2, 150):                       # 148 distinct schedules
        ngal, xi = halotab.predict_batch(theta[:32 * n_tiles])
        if n_tiles % 37 == 0:
            assert_rel(ngal[:40], expect[0], RTOL)
            assert_rel(xi[:40], expect[1], RTOL)
    ngal, xi = halotab.predict_batch(theta[:64])
    assert_rel(ngal[:40], expect[0][:40], RTOL)
    assert_rel(xi[:40], expect[1][:40], RTOL)
