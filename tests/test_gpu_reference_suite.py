"""The reference's own test suite (tests/test_general.py), run against this library.

Same test names, model objects and assertions; the data are the reference's example
files kept under tests/golden/ (the Bolshoi-Planck wp / DeltaSigma tables of the
reference's docs and the four AbacusSummit DeltaSigma tables along log_eta) instead of
its multi-gigabyte database, so the interpolator has one dimension."""
import os

import numpy as np
import pytest
from scipy.interpolate import interp1d

from util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def halotab():
    from tabcorr_amd import TabCorr, hdf5
    if not hdf5.available():
        pytest.skip('libhdf5 not found')
    return {tpcf: TabCorr.read(os.path.join(GOLDEN, 'bolplanck_%s.hdf5' % tpcf))
            for tpcf in ['wp', 'ds']}


@pytest.fixture(scope='module')
def interpolator():
    from tabcorr_amd import Interpolator, hdf5
    if not hdf5.available():
        pytest.skip('libhdf5 not found')
    return Interpolator.read(os.path.join(GOLDEN, 'ds_efficient.hdf5'))


@pytest.fixture
def model():
    from tabcorr_amd import Zheng07Model
    return Zheng07Model(redshift=0.0)


@pytest.fixture
def abacus_model():
    from tabcorr_amd import Zheng07Model
    model = Zheng07Model(redshift=0.5, prim_haloprop_key='halo_m258m')
    model.param_dict['log_eta'] = 0.0
    return model


@pytest.mark.parametrize('tpcf', ['wp', 'ds'])
def test_separate_gal_type(halotab, model, tpcf):
    # tests/test_general.py:8-28: the total clustering is the sum of its components
    ngal, xi = halotab[tpcf].predict(model)
    ngal_sep, xi_sep = halotab[tpcf].predict(model, separate_gal_type=True)
    assert len(ngal_sep) == 2
    assert len(xi_sep) == (2 if tpcf == 'ds' else 3)
    assert np.isclose(ngal, np.sum([n for n in ngal_sep.values()]), atol=0, rtol=1e-6)
    assert np.allclose(xi, np.sum([x for x in xi_sep.values()], axis=0), atol=0, rtol=1e-6)


def test_separate_gal_type_interpolator(interpolator, abacus_model):
    ngal, xi = interpolator.predict(abacus_model)
    ngal_sep, xi_sep = interpolator.predict(abacus_model, separate_gal_type=True)
    assert len(ngal_sep) == 2 and len(xi_sep) == 2
    assert np.isclose(ngal, np.sum([n for n in ngal_sep.values()]), atol=0, rtol=1e-6)
    assert np.allclose(xi, np.sum([x for x in xi_sep.values()], axis=0), atol=0, rtol=1e-6)


def test_n_gauss_prim(halotab, model):
    # tests/test_general.py:31-44: 1 vs 10 quadrature nodes differ, 10 vs 100 do not
    ngal_1, xi_1 = halotab['wp'].predict(model, n_gauss_prim=1)
    ngal_2, xi_2 = halotab['wp'].predict(model, n_gauss_prim=10)
    ngal_3, xi_3 = halotab['wp'].predict(model, n_gauss_prim=100)
    assert not np.isclose(ngal_1, ngal_2, atol=0, rtol=1e-6)
    assert not np.allclose(xi_1, xi_2, atol=0, rtol=1e-6)
    assert np.isclose(ngal_2, ngal_3, atol=0, rtol=1e-6)
    assert np.allclose(xi_2, xi_3, atol=0, rtol=1e-6)


def test_interpolator(interpolator, abacus_model):
    # tests/test_general.py:47-69: the multi-dimensional spline interpolation agrees with
    # scipy's one-dimensional cubic interpolation between the tabulated values
    model = abacus_model
    bins = interpolator.xp[0]
    xi_bins = []
    for x in bins:
        model.param_dict['log_eta'] = x
        xi_bins.append(interpolator.predict(model)[1])
    xi_bins = np.array(xi_bins)
    for x in np.linspace(np.amin(bins), np.amax(bins), 10):
        model.param_dict['log_eta'] = x
        xi_tabcorr = interpolator.predict(model)[1]
        xi_scipy = [interp1d(bins, xi_bins[:, i], kind='cubic')(x)
                    for i in range(len(xi_tabcorr))]
        assert np.allclose(xi_tabcorr, xi_scipy)
