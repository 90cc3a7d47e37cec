"""Seeded synthetic halo tables and HOD parameter draws.

NumPy only, and free of any GPU or reference dependency, so that the very same
generator is used by ``bench.py``, by the parity tests and by
``tests/golden/make_golden.py`` (which feeds these arrays to the reference
implementation to record golden outputs).

The layout produced here is the one ``TabCorr.tabulate`` leaves behind in the
reference (``tabcorr/tabcorr.py:199-234`` and ``:346-368``): one row per
(gal_type, secondary bin, primary bin) with the primary bin running fastest,
all ``centrals`` rows first and all ``satellites`` rows second, and a
correlation matrix of shape ``(R, P)`` whose columns are the packed lower
triangle ``p = i (i + 1) / 2 + j, j <= i`` in mode ``'auto'``
(``tabcorr/tabcorr.py:770-806``) or the rows themselves in mode ``'cross'``.
"""

import numpy as np

GAL_TYPE_DTYPE = np.dtype([
    ('n_h', '<f8'),
    ('log_prim_haloprop_min', '<f8'),
    ('log_prim_haloprop_max', '<f8'),
    ('sec_haloprop_percentile_min', '<f8'),
    ('sec_haloprop_percentile_max', '<f8'),
    ('prim_haloprop', '<f8'),
    ('sec_haloprop_percentile', '<f8'),
    ('prim_haloprop_dist_index', '<f8'),
    ('gal_type', 'S10')])

# Uniform prior box of the Zheng07 draws: logMmin, sigma_logM, logM0, logM1,
# alpha.
ZHENG07_KEYS = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')
ZHENG07_LOW = np.array([11.5, 0.1, 11.0, 12.5, 0.7])
ZHENG07_HIGH = np.array([13.5, 0.8, 13.0, 14.5, 1.4])


def synthetic_gal_type(n_prim, n_sec=1, seed=0, log_m_min=10.5,
                       log_m_max=15.0):
    """Return the structured ``gal_type`` array of a synthetic table.

    Parameters
    ----------
    n_prim : int
        Number of primary (log mass) bins.
    n_sec : int, optional
        Number of secondary-property percentile bins.
    seed : int, optional
        Seed of the ``prim_haloprop_dist_index`` draw.

    Returns
    -------
    gal_type : numpy.ndarray
        Structured array with ``2 * n_sec * n_prim`` rows.
    """
    rng = np.random.default_rng(seed)
    edges = np.linspace(log_m_min, log_m_max, n_prim + 1)
    sec_edges = np.linspace(-1e-3, 1 + 1e-3, n_sec + 1)
    log_m = 0.5 * (edges[1:] + edges[:-1])
    n_h = (3e-2 * 10**(-0.9 * (log_m - 10.5)) *
           np.exp(-10**(log_m - 14.5)) / n_sec)
    dist_index = rng.uniform(-2.5, -1.5, size=n_sec * n_prim)

    half = np.zeros(n_sec * n_prim, dtype=GAL_TYPE_DTYPE)
    for i_sec in range(n_sec):
        sel = slice(i_sec * n_prim, (i_sec + 1) * n_prim)
        half['n_h'][sel] = n_h
        half['log_prim_haloprop_min'][sel] = edges[:-1]
        half['log_prim_haloprop_max'][sel] = edges[1:]
        half['sec_haloprop_percentile_min'][sel] = sec_edges[i_sec]
        half['sec_haloprop_percentile_max'][sel] = sec_edges[i_sec + 1]
        half['prim_haloprop'][sel] = 10**log_m
        half['sec_haloprop_percentile'][sel] = 0.5 * (
            sec_edges[i_sec] + sec_edges[i_sec + 1])
    half['prim_haloprop_dist_index'] = dist_index

    gal_type = np.concatenate([half, half])
    gal_type['gal_type'][:len(half)] = b'centrals'
    gal_type['gal_type'][len(half):] = b'satellites'
    return gal_type


def synthetic_tpcf_matrix(n_bins, n_r, mode='auto', seed=0, dtype=np.float64):
    """Return a synthetic correlation matrix ``exp(N(2, 1.5^2))``.

    Values are rounded through float32, exactly as a table that went through
    the reference's default ``write`` (``tabcorr/tabcorr.py:448``) and ``read``
    (``tabcorr/tabcorr.py:399``) would be.
    """
    rng = np.random.default_rng(seed)
    n_pairs = n_bins * (n_bins + 1) // 2 if mode == 'auto' else n_bins
    matrix = np.empty((n_r, n_pairs), dtype=np.float32)
    # Row by row keeps the peak memory of the largest configuration low.
    for i in range(n_r):
        matrix[i] = np.exp(rng.normal(2.0, 1.5, size=n_pairs))
    return matrix.astype(dtype)


def synthetic_table(n_prim, n_sec=1, tpcf_shape=(19, ), mode='auto', seed=0,
                    dtype=np.float64, redshift=0.0):
    """Return the ingredients of a synthetic ``TabCorr`` table as a dict.

    Keys: ``gal_type`` (structured array), ``tpcf_matrix`` ``(R, P)``,
    ``tpcf_shape`` (tuple) and ``attrs`` (dict as in
    ``tabcorr/tabcorr.py:356-363``).
    """
    gal_type = synthetic_gal_type(n_prim, n_sec, seed=seed)
    n_r = int(np.prod(tpcf_shape))
    matrix = synthetic_tpcf_matrix(
        len(gal_type), n_r, mode=mode, seed=seed + 1000, dtype=dtype)
    attrs = {'tpcf': 'wp' if len(tpcf_shape) == 1 else 'rp_pi_tpcf',
             'mode': mode, 'simname': 'synthetic', 'redshift': redshift,
             'Num_ptcl_requirement': 300, 'prim_haloprop_key': 'halo_mvir',
             'sec_haloprop_key': 'halo_nfw_conc'}
    return {'gal_type': gal_type, 'tpcf_matrix': matrix,
            'tpcf_shape': tuple(int(s) for s in tpcf_shape), 'attrs': attrs}


def zheng07_draws(n_draws, seed=1):
    """Return ``(n_draws, 5)`` uniform Zheng07 draws in ``ZHENG07_KEYS`` order.
    """
    rng = np.random.default_rng(seed)
    return rng.uniform(ZHENG07_LOW, ZHENG07_HIGH, size=(n_draws, 5))


def interpolator_grid(shape, keys=None, low=None, high=None):
    """Return the regular grid of extra parameters of a synthetic interpolator.

    Returns
    -------
    keys : tuple of str
    axes : list of numpy.ndarray
        Abscissae per dimension.
    points : numpy.ndarray
        ``(K, D)`` array in C order over ``axes`` (last key fastest).
    """
    if keys is None:
        keys = ('log_eta', 'alpha_s', 'alpha_c')[:len(shape)]
    if low is None:
        low = (-0.5, 0.8, 0.0)[:len(shape)]
    if high is None:
        high = (0.5, 1.2, 0.4)[:len(shape)]
    axes = [np.linspace(lo, hi, n) for lo, hi, n in zip(low, high, shape)]
    mesh = np.meshgrid(*axes, indexing='ij')
    points = np.stack([m.ravel() for m in mesh], axis=-1)
    return tuple(keys), axes, points


def synthetic_interpolator(shape, n_prim, n_sec=1, tpcf_shape=(19, ),
                           mode='auto', seed=0, dtype=np.float64):
    """Return tables on a regular grid, as ``scripts/tabulate_snapshot.py``
    would produce for a phase-space parameter grid
    (``scripts/tabulate_snapshot.py:158-165``): one shared ``gal_type`` table
    and a smoothly varying correlation matrix per grid point.

    Returns
    -------
    tables : list of dict
        As returned by `synthetic_table`.
    keys : tuple of str
    points : numpy.ndarray
        ``(K, D)`` parameter values, one row per table.
    """
    keys, axes, points = interpolator_grid(shape)
    base = synthetic_table(n_prim, n_sec, tpcf_shape, mode, seed, dtype)
    rng = np.random.default_rng(seed + 2000)
    slopes = rng.normal(0, 0.3, size=(len(shape), ) + base['tpcf_matrix'].shape)
    curv = rng.normal(0, 0.2, size=(len(shape), ) + base['tpcf_matrix'].shape)
    tables = []
    for point in points:
        factor = np.ones_like(base['tpcf_matrix'])
        for d, x in enumerate(point):
            x0 = 0.5 * (axes[d][0] + axes[d][-1])
            factor = factor * np.exp(slopes[d] * (x - x0) +
                                     curv[d] * (x - x0)**2)
        table = dict(base)
        table['tpcf_matrix'] = (base['tpcf_matrix'] * factor).astype(
            np.float32).astype(dtype)
        tables.append(table)
    return tables, keys, points
