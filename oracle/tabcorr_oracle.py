"""CPU oracle: a NumPy restatement of the reference's ``predict`` path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and there only as the checker / the timed CPU baseline.  Nothing
under ``tabcorr_amd/`` imports it and the product path has no CPU fallback.

Every function cites the reference lines (relative to ``/root/reference``,
TabCorr v1.2.0) whose arithmetic it restates, operation for operation (same
gather -> ``einsum('ij, j')`` sequence), so that it doubles as the "port"
CPU baseline.

Parity status
-------------
* pinned: everything from the Gauss-Legendre bin average of ``mean_occupation``
  onwards (``predict`` total and per-gal-type, the packed index map, spline
  matrices, N-D spline interpolation, ``Interpolator.predict``) is checked in
  ``tests/test_oracle_golden.py`` against ``tests/golden/*.npz``, which were
  recorded by running the unmodified reference (``tests/golden/make_golden.py``).
* PARITY UNPINNED: the Zheng07 occupation functions themselves
  (`zheng07_centrals`, `zheng07_satellites`, `heaviside_assembias`, and the
  Leauthaud et al. 2011 family `Leauthaud11`).  In the
  reference they are halotools callbacks (call sites
  ``tabcorr/tabcorr.py:556-563``); halotools (unpinned dependency,
  ``pyproject.toml:12``) is not available, so they restate Zheng et al. (2007)
  eqs. 1, 3 and Hearin et al. (2016) and are pinned only against the duck-typed
  model of ``tests/golden/make_golden.py``.

Tables are plain dicts: ``gal_type`` (structured array, fields as written at
``tabcorr/tabcorr.py:443-463``), ``tpcf_matrix`` ``(R, P)`` float64,
``tpcf_shape`` (tuple), ``attrs`` (dict).
"""

import itertools
import math

import numpy as np

_erf = np.vectorize(math.erf, otypes=[np.float64])

try:  # scipy is optional; math.erf is the fallback (both are libm-grade).
    from scipy.special import erf as _erf  # noqa: F811
except ImportError:  # pragma: no cover
    pass


# -- occupation functions (halotools callbacks; PARITY UNPINNED) -------------

def zheng07_centrals(prim_haloprop, theta):
    """<N_cen> = 1/2 [1 + erf((log10 M - logMmin) / sigma_logM)].

    Zheng et al. (2007) eq. 1; called at ``tabcorr/tabcorr.py:556-559``.
    """
    return 0.5 * (1.0 + _erf((np.log10(prim_haloprop) - theta[0]) / theta[1]))


def zheng07_satellites(prim_haloprop, theta, modulate_with_cenocc=False):
    """<N_sat> = ((M - M0) / M1)^alpha for M > M0, else 0.

    Zheng et al. (2007) eq. 3; called at ``tabcorr/tabcorr.py:560-563``.
    """
    m0 = 10.0**theta[2]
    m1 = 10.0**theta[3]
    prim_haloprop = np.asarray(prim_haloprop, dtype=np.float64)
    n = np.zeros(prim_haloprop.shape)
    use = prim_haloprop - m0 > 0
    n[use] = ((prim_haloprop[use] - m0) / m1)**theta[4]
    if modulate_with_cenocc:
        n = n * zheng07_centrals(prim_haloprop, theta)
    return n


def heaviside_assembias(baseline, percentile, strength, lower, upper,
                        split=0.5):
    """Heaviside assembly bias (Hearin et al. 2016) around a baseline <N>.

    Bins above the percentile ``split`` receive ``+d``, the others
    ``-d f1 / f2`` (f1 = 1 - split), with ``d = strength * d_max`` and
    ``d_max`` the largest shift that keeps both inside ``[lower, upper]``;
    ``strength`` is limited to [-1, 1].
    """
    f1 = 1.0 - split
    f2 = split
    # halotools' HeavisideAssembias.assembias_strength clips to [-1, 1] (two
    # np.where calls, so a NaN strength stays NaN)
    strength = 1.0 if strength > 1 else strength
    strength = -1.0 if strength < -1 else strength
    if strength >= 0:
        dmax = np.minimum(upper - baseline, (baseline - lower) * f2 / f1)
    else:
        dmax = np.minimum(baseline - lower, (upper - baseline) * f2 / f1)
    d1 = strength * dmax
    return np.where(percentile > split, baseline + d1,
                    baseline - d1 * f1 / f2)


class Zheng07:
    """Callbacks with the signature used at ``tabcorr/tabcorr.py:556-563``."""

    def __init__(self, theta, modulate_with_cenocc=False, assembias=None,
                 split=0.5):
        self.theta = np.asarray(theta, dtype=np.float64)
        self.modulate_with_cenocc = modulate_with_cenocc
        self.assembias = assembias
        self.split = split

    def mean_occupation_centrals(self, prim_haloprop,
                                 sec_haloprop_percentile=None):
        n = zheng07_centrals(prim_haloprop, self.theta)
        if self.assembias is not None:
            n = heaviside_assembias(n, sec_haloprop_percentile,
                                    self.assembias[0], 0.0, 1.0, self.split)
        return n

    def mean_occupation_satellites(self, prim_haloprop,
                                   sec_haloprop_percentile=None):
        n = zheng07_satellites(prim_haloprop, self.theta,
                               self.modulate_with_cenocc)
        if self.assembias is not None:
            n = heaviside_assembias(n, sec_haloprop_percentile,
                                    self.assembias[1], 0.0, np.inf,
                                    self.split)
        return n


# -- Leauthaud et al. (2011) family (halotools callbacks; PARITY UNPINNED) ----------
#
# Restated from the literature, not from halotools' source: Behroozi, Conroy &
# Wechsler (2010) eq. 21 for the stellar-to-halo mass relation and Leauthaud et
# al. (2011) eqs. 8 and 12 for the occupations, with halotools' parameter
# conventions (knee mass 1e12, h = 0.7 inside the stellar-to-halo mass relation and
# h = 0.72 in the satellite terms, scatter sqrt(2) sigma).  halotools
# inverts the relation by cubic-spline interpolation of a 100-point table; here
# the inverse is exact (bisection to the last bit), so agreement with halotools
# is limited by ITS table, and these functions are pinned only against the duck
# model of ``tests/golden/make_golden.py`` (an independent root finder).
# theta: logm0, logm1, beta, delta, gamma, scatter, alphasat, bsat, betasat,
# bcut, betacut, threshold, h of the relation, h of the satellite terms.

def behroozi10_log_halo_mass(log_stellar_mass, theta):
    x = log_stellar_mass + 2.0 * np.log10(theta[12]) - theta[0]
    return (theta[1] + theta[2] * x +
            10.0**(theta[3] * x) / (1.0 + 10.0**(-theta[4] * x)) - 0.5 -
            np.log10(theta[12]))


def behroozi10_log_stellar_mass(log_halo_mass, theta):
    """Inverse of `behroozi10_log_halo_mass` by bisection (the relation is
    increasing for positive beta, delta, gamma)."""
    log_halo_mass = np.atleast_1d(np.asarray(log_halo_mass, dtype=np.float64))
    lo = np.full(log_halo_mass.shape, -60.0)
    hi = np.full(log_halo_mass.shape, 60.0)
    with np.errstate(all='ignore'):
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            below = behroozi10_log_halo_mass(mid, theta) < log_halo_mass
            lo = np.where(below, mid, lo)
            hi = np.where(below, hi, mid)
    return 0.5 * (lo + hi)


class Leauthaud11:
    """Callbacks with the signature used at ``tabcorr/tabcorr.py:556-563``."""

    def __init__(self, theta, modulate_with_cenocc=True):
        self.theta = np.asarray(theta, dtype=np.float64)
        self.modulate_with_cenocc = modulate_with_cenocc

    def mean_occupation_centrals(self, prim_haloprop,
                                 sec_haloprop_percentile=None):
        t = self.theta
        log_mstar = behroozi10_log_stellar_mass(np.log10(prim_haloprop), t)
        return 0.5 * (1.0 - _erf((t[11] - log_mstar) / (np.sqrt(2.0) * t[5])))

    def mean_occupation_satellites(self, prim_haloprop,
                                   sec_haloprop_percentile=None):
        t = self.theta
        prim_haloprop = np.asarray(prim_haloprop, dtype=np.float64)
        knee = 10.0**behroozi10_log_halo_mass(t[11], t) * t[13]
        m_sat = 1e12 * t[7] * (knee / 1e12)**t[8]
        m_cut = 1e12 * t[9] * (knee / 1e12)**t[10]
        n = (np.exp(-m_cut / (prim_haloprop * t[13])) *
             (prim_haloprop * t[13] / m_sat)**t[6])
        if self.modulate_with_cenocc:
            n = n * self.mean_occupation_centrals(prim_haloprop)
        return n


def predict_leauthaud11_batch(table, theta, separate_gal_type=False,
                              n_gauss_prim=10, modulate_with_cenocc=True):
    """`predict` for a batch of Leauthaud11 parameter vectors (14 columns)."""
    theta = np.atleast_2d(theta)
    cache = {}
    results = [predict(table, mean_occupation(
        table, Leauthaud11(t, modulate_with_cenocc), n_gauss_prim),
        separate_gal_type, cache) for t in theta]
    return _stack(results, separate_gal_type)


# -- tabcorr/tabcorr.py ---------------------------------------------------------

def symmetric_matrix_to_array(matrix, check_symmetry=True):
    """Packed lower triangle, ``p = i (i + 1) / 2 + j`` with ``j <= i``.

    Restates ``tabcorr/tabcorr.py:770-806``.
    """
    matrix = np.asarray(matrix)
    if check_symmetry:
        if (matrix.shape[0] != matrix.shape[1] or
                not np.all(matrix == matrix.T)):
            raise ValueError('The matrix you provided is not symmetric.')
    n_dim = matrix.shape[0]
    sel = np.zeros((n_dim**2 + n_dim) // 2, dtype=int)
    for i in range(n_dim):
        sel[(i * (i + 1)) // 2:(i * (i + 1)) // 2 + (i + 1)] = np.arange(
            i * n_dim, i * n_dim + i + 1)
    return matrix.ravel()[sel]


def is_centrals(gal_type):
    """Boolean mask of the ``centrals`` rows (``tabcorr/tabcorr.py:555``; the
    reference relies on astropy comparing a bytes column with a str)."""
    column = gal_type['gal_type']
    return (column == b'centrals') | (column == 'centrals')


def mean_occupation(table, model, n_gauss_prim=10):
    """Gauss-Legendre average of <N> over each primary bin, weighted by the
    power law ``M^(d + 1)`` in log M.  Restates ``tabcorr/tabcorr.py:537-578``
    (the consistency checks at ``:496-535`` live in the host class).
    """
    gal_type = table['gal_type']
    log_min = gal_type['log_prim_haloprop_min']
    log_max = gal_type['log_prim_haloprop_max']
    d_log = log_max - log_min
    x_gauss, w_gauss = np.polynomial.legendre.leggauss(n_gauss_prim)
    x_gauss = (x_gauss + 1) / 2                                    # :546

    prim_haloprop = 10**(log_min + d_log * x_gauss[:, np.newaxis]).T.ravel()
    percentile = np.repeat(gal_type['sec_haloprop_percentile'], n_gauss_prim)
    select = np.repeat(is_centrals(gal_type), n_gauss_prim)        # :550-555

    occupation = np.zeros(len(prim_haloprop))
    occupation[select] = model.mean_occupation_centrals(
        prim_haloprop=prim_haloprop[select],
        sec_haloprop_percentile=percentile[select])
    occupation[~select] = model.mean_occupation_satellites(
        prim_haloprop=prim_haloprop[~select],
        sec_haloprop_percentile=percentile[~select])
    occupation = occupation.reshape((len(gal_type), n_gauss_prim))
    prim_haloprop = prim_haloprop.reshape(occupation.shape)

    if 'prim_haloprop_dist_index' in gal_type.dtype.names:         # :568-574
        n = gal_type['prim_haloprop_dist_index'][:, np.newaxis] + 1
    else:
        n = 0
    return (np.sum(w_gauss * occupation * prim_haloprop**n, axis=-1) /
            np.sum(w_gauss * prim_haloprop**n, axis=-1))           # :576-578


def pair_indices(n_bins):
    """Row/column index and prefactor of every packed pair.

    Restates the cache built at ``tabcorr/tabcorr.py:626-639``.
    """
    index_1 = np.repeat(np.arange(n_bins), n_bins).reshape(n_bins, n_bins)
    index_2 = np.tile(np.arange(n_bins), n_bins).reshape(n_bins, n_bins)
    index_1 = symmetric_matrix_to_array(index_1, check_symmetry=False)
    index_2 = symmetric_matrix_to_array(index_2, check_symmetry=False)
    prefactor = np.where(index_1 == index_2, 1, 2)
    return index_1, index_2, prefactor


def predict(table, occupation, separate_gal_type=False, cache=None):
    """Contract the table with the mean occupations of one model.

    Restates ``tabcorr/tabcorr.py:623-683``.  ``cache`` plays the role of the
    attributes the reference caches on ``self`` (``:626-639``).
    """
    gal_type = table['gal_type']
    matrix = table['tpcf_matrix']
    tpcf_shape = tuple(table['tpcf_shape'])
    mode = table['attrs']['mode']
    ngal = occupation * gal_type['n_h']                            # :623

    if mode == 'auto':
        if cache is None:
            cache = {}
        if 'pairs' not in cache:
            cache['pairs'] = pair_indices(len(gal_type))
        index_1, index_2, prefactor = cache['pairs']
        ngal_sq = prefactor * ngal[index_1] * ngal[index_2]        # :641-642

    if not separate_gal_type:
        if mode == 'auto':
            xi = np.einsum('ij, j', matrix, ngal_sq) / np.sum(ngal_sq)
        else:
            xi = np.einsum('ij, j', matrix, ngal) / np.sum(ngal)   # :646-649
        return np.sum(ngal), xi.reshape(tpcf_shape)

    if mode == 'auto':
        xi = (matrix * ngal_sq) / np.sum(ngal_sq)                  # :653
    else:
        xi = (matrix * ngal) / np.sum(ngal)                        # :655

    ngal_dict = {}
    xi_dict = {}
    # np.unique order: 'centrals' < 'satellites' (:660).
    names = [name.decode() if isinstance(name, bytes) else str(name)
             for name in np.unique(gal_type['gal_type'])]
    column = np.array([name.decode() if isinstance(name, bytes) else str(name)
                       for name in gal_type['gal_type']])
    for name in names:
        ngal_dict[name] = np.sum(ngal[column == name])             # :660-662
    if mode == 'auto':
        for name_1, name_2 in itertools.combinations_with_replacement(
                names, 2):
            mask = symmetric_matrix_to_array(
                np.outer(name_1 == column, name_2 == column) |
                np.outer(name_2 == column, name_1 == column))      # :668-673
            xi_dict['%s-%s' % (name_1, name_2)] = np.sum(
                xi * mask, axis=1).reshape(tpcf_shape)
    else:
        for name in names:
            xi_dict[name] = np.sum(
                xi * (column == name), axis=1).reshape(tpcf_shape)  # :677-681
    return ngal_dict, xi_dict


def predict_zheng07(table, theta, separate_gal_type=False, n_gauss_prim=10,
                    modulate_with_cenocc=False, assembias=None, cache=None):
    """``TabCorr.predict(model)`` for one Zheng07 parameter vector."""
    model = Zheng07(theta, modulate_with_cenocc, assembias)
    occupation = mean_occupation(table, model, n_gauss_prim)
    return predict(table, occupation, separate_gal_type, cache)


def predict_zheng07_batch(table, theta, separate_gal_type=False,
                          n_gauss_prim=10, modulate_with_cenocc=False,
                          assembias=None):
    """Loop `predict_zheng07` over draws, as a user's MCMC loop would
    (``README.md:72-75``).  Returns stacked arrays (dicts of stacked arrays if
    ``separate_gal_type``)."""
    theta = np.atleast_2d(theta)
    cache = {}
    results = [predict_zheng07(
        table, t, separate_gal_type, n_gauss_prim, modulate_with_cenocc,
        None if assembias is None else assembias[i], cache)
        for i, t in enumerate(theta)]
    return _stack(results, separate_gal_type)


def _stack(results, separate_gal_type):
    if not separate_gal_type:
        return (np.array([r[0] for r in results]),
                np.array([r[1] for r in results]))
    return ({key: np.array([r[0][key] for r in results])
             for key in results[0][0]},
            {key: np.array([r[1][key] for r in results])
             for key in results[0][1]})


# -- tabcorr/interpolator.py -------------------------------------------------------

def spline_interpolation_matrix(xp):
    """Matrix ``a`` of shape ``(n - 1, 4, n)`` of a not-a-knot cubic spline.

    Restates ``tabcorr/interpolator.py:219-272``.
    """
    xp = np.asarray(xp, dtype=np.float64)
    if len(xp) < 4:
        raise ValueError('Cannot perform spline interpolation with less than' +
                         ' 4 values.')
    n = len(xp) - 1
    m = np.zeros((4 * n, 4 * n))
    for i in range(n):                                   # values, :247-249
        m[i][i * 4:(i + 1) * 4] = xp[i]**np.arange(4)
        m[i + n][i * 4:(i + 1) * 4] = xp[i + 1]**np.arange(4)
    for i in range(n - 1):                               # C1, C2, :252-258
        m[i + 2 * n][i * 4 + 1:(i + 1) * 4] = (
            np.array([1, 2, 3]) * xp[i + 1]**np.arange(3))
        m[i + 2 * n][(i + 1) * 4 + 1:(i + 2) * 4] = -(
            np.array([1, 2, 3]) * xp[i + 1]**np.arange(3))
        m[i + 3 * n - 1][i * 4 + 2:(i + 1) * 4] = (
            np.array([2, 6]) * xp[i + 1]**np.arange(2))
        m[i + 3 * n - 1][(i + 1) * 4 + 2:(i + 2) * 4] = -(
            np.array([2, 6]) * xp[i + 1]**np.arange(2))
    m[-1][3] = 6 * xp[1]                                 # not-a-knot, :261-264
    m[-1][7] = -6 * xp[1]
    m[-2][-5] = 6 * xp[-2]
    m[-2][-1] = -6 * xp[-2]
    m = np.linalg.inv(m)                                 # :267
    a = np.zeros((4 * n, len(xp)))
    a[:, :-1] = m[:, :n]
    a[:, 1:] += m[:, n:2 * n]
    return a.reshape((n, 4, len(xp)))


def spline_interpolate(x, xp, a, yp, extrapolate=False):
    """Tensor-product spline evaluation along the first ``len(x)`` axes.

    Restates ``tabcorr/interpolator.py:275-331``.
    """
    if not isinstance(xp, list):
        xp = [xp]
    if not isinstance(a, list):
        a = [a]
    x = np.atleast_1d(x)
    for xi, ai, xpi in zip(x, a, xp):
        i_spline = np.digitize(xi, xpi) - 1
        if xi == xpi[-1]:
            i_spline = len(xpi) - 2
        if i_spline < 0 or i_spline >= len(xpi) - 1:
            if not extrapolate:
                raise ValueError(
                    'The x-coordinates are outside of the interpolation ' +
                    'range and extrapolation is turned off.')
            i_spline = min(max(i_spline, 0), len(xpi) - 2)
        yp = np.einsum('ij,j...,i', ai[i_spline], yp, xi**np.arange(4))
    return yp


def interpolator_setup(tables, points):
    """Grid bookkeeping of ``Interpolator.__init__``
    (``tabcorr/interpolator.py:32-70``).

    ``points`` is ``(K, D)``, one row per table.  Returns a dict with ``xp``,
    ``a``, ``order`` (table indices in lexicographic grid order, i.e. the
    sorted ``tabcorr_index`` column), ``unique_index`` / ``unique_inverse``.
    """
    points = np.asarray(points, dtype=np.float64)
    if points.ndim == 1:
        points = points[:, np.newaxis]
    if len(tables) != len(points):
        raise ValueError("The number of TabCorr instances does not match" +
                         " the number of entries in 'param_dict_table'.")
    xp = [np.sort(np.unique(points[:, d])) for d in range(points.shape[1])]
    a = [spline_interpolation_matrix(x) for x in xp]
    if (np.prod([len(x) for x in xp]) != len(points) or
            len(np.unique(points, axis=0)) != len(points)):
        raise ValueError("The 'param_dict_table' does not describe a grid.")
    order = np.lexsort(points.T[::-1])                   # :59-61
    flat = [np.array(t['gal_type'].tolist()).ravel() for t in tables]
    unique = np.unique(flat, axis=0, return_index=True, return_inverse=True)
    return {'xp': xp, 'a': a, 'order': order,
            'unique_index': unique[1],
            'unique_inverse': np.asarray(unique[2]).ravel()}


def interpolator_predict(tables, setup, model, x_model,
                         separate_gal_type=False, n_gauss_prim=10,
                         extrapolate=False):
    """``Interpolator.predict`` (``tabcorr/interpolator.py:179-216``)."""
    occupation = [mean_occupation(tables[i], model, n_gauss_prim)
                  for i in setup['unique_index']]        # :181-184
    results = []
    for k in setup['order']:                             # :188-194
        results.append(predict(
            tables[k], occupation[setup['unique_inverse'][k]],
            separate_gal_type))
    shape = [len(xp) for xp in setup['xp']]
    output = []
    for i in range(2):                                   # :198-214
        if separate_gal_type:
            output.append(dict())
            for key in results[0][i].keys():
                data = np.array([r[i][key] for r in results])
                data = data.reshape(shape + list(data.shape[1:]))
                output[-1][key] = spline_interpolate(
                    x_model, setup['xp'], setup['a'], data,
                    extrapolate=extrapolate)
        else:
            data = np.array([r[i] for r in results])
            data = data.reshape(shape + list(data.shape[1:]))
            output.append(spline_interpolate(
                x_model, setup['xp'], setup['a'], data,
                extrapolate=extrapolate))
    return tuple(output)


def interpolator_predict_zheng07_batch(tables, setup, theta, x,
                                       separate_gal_type=False,
                                       n_gauss_prim=10, extrapolate=False,
                                       modulate_with_cenocc=False):
    theta = np.atleast_2d(theta)
    x = np.atleast_2d(x)
    results = [interpolator_predict(
        tables, setup, Zheng07(t, modulate_with_cenocc), xv,
        separate_gal_type, n_gauss_prim, extrapolate)
        for t, xv in zip(theta, x)]
    return _stack(results, separate_gal_type)
