"""Two-point functions for the tabulation step, counted on the GPU.

Mirrors ``tabcorr/corrfunc.py`` of johannesulf/TabCorr v1.2.0 (`wp`,
`s_mu_tpcf`), whose functions wrap the Corrfunc pair counters so that they can be handed to
``TabCorr.tabulate`` in the place of the halotools two-point functions.  Here
the pair counts come from this package's HIP kernel
(``tabcorr_amd/csrc/paircount.hip`` through ``tc_pair_count_rppi``): same
signature, same arithmetic around the counts, no Corrfunc.

`mean_delta_sigma` stands in for ``halotools.mock_observables.mean_delta_sigma``
(the two-point function of the reference's excess-surface-density tables,
``scripts/tabulate_snapshot.py:228-237``): the per-object mass in cylinders is
counted on the GPU (``tc_mass_in_cylinders``), for the halos of all bins in one
pass when `TabCorr.tabulate` calls it.

`compute_tpcf_matrix` is the MI355X-native form of the reference's
``compute_tpcf_matrix`` (``tabcorr/tabcorr.py:846-922``) for ``tpcf = wp``:
instead of one pair count per pair of halo bins from a pool of processes, every
point is labelled with its halo bin and ONE pass over the box fills the counts
of all bin pairs.
"""

import ctypes

import numpy as np

from . import _lib


def _period(period):
    if period is None:
        raise ValueError('A periodic box (period) is required.')
    if isinstance(period, (float, int)):
        period = (period, period, period)
    return _lib.contiguous(np.asarray(period, dtype=np.float64).reshape(3))


def _positions(sample):
    sample = _lib.contiguous(np.asarray(sample, dtype=np.float64))
    if sample.ndim != 2 or sample.shape[1] != 3:
        raise ValueError('positions must have shape (n, 3).')
    return sample


def pair_count_rppi(sample1, rp_bins, pi_max, sample2=None, period=None,
                    n_pi=None):
    """Ordered pair counts ``(n_rp, n_pi)`` (uint64) between ``sample1`` and
    ``sample2`` -- ``None``: of ``sample1`` with itself, every pair counted
    twice -- as ``Corrfunc.theory.DDrppi(autocorr, 1, pi_max, rp_bins, ...,
    periodic=True)`` reports them in ``npairs``
    (``tabcorr/corrfunc.py:62-84``)."""
    lib = _lib.load()
    _lib.require_device()
    sample1 = _positions(sample1)
    rp_bins = _lib.contiguous(np.asarray(rp_bins, dtype=np.float64))
    if n_pi is None:
        n_pi = int(pi_max)
    box = _period(period)
    npairs = np.zeros((len(rp_bins) - 1, n_pi), dtype=np.uint64)
    if sample2 is not None:
        sample2 = _positions(sample2)
    _lib.check(lib.tc_pair_count_rppi(
        _lib.as_double_p(sample1), len(sample1),
        _lib.as_double_p(sample2) if sample2 is not None else None,
        len(sample2) if sample2 is not None else 0, _lib.as_double_p(box),
        _lib.as_double_p(rp_bins), len(rp_bins) - 1, float(pi_max), n_pi,
        npairs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    return npairs


def wp(sample1, rp_bins, pi_max, sample2=None, period=None, do_auto=True,
       do_cross=False):
    """Drop-in for ``tabcorr.corrfunc.wp`` (``tabcorr/corrfunc.py:6-95``,
    itself a stand-in for ``halotools.mock_observables.wp``): the projected
    correlation function of ``sample1`` (``do_auto``) or between ``sample1``
    and ``sample2`` (``do_cross``) in a periodic box.

    Raises
    ------
    ValueError
        If ``do_auto`` and ``do_cross`` have the same value.
    """
    if (do_auto and do_cross) or (not do_auto and not do_cross):
        raise ValueError("'do_auto' and 'do_cross' cannot both be True or " +
                         "False.")
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    box = _period(period)
    if do_auto:
        npairs = pair_count_rppi(sample1, rp_bins, pi_max, None, box)
        n_exp = (len(sample1) * len(sample1) / np.prod(box) * np.pi *
                 np.diff(rp_bins**2) * 2 * pi_max)
    else:
        npairs = pair_count_rppi(sample1, rp_bins, pi_max, sample2, box)
        n_exp = (len(sample1) * len(sample2) / np.prod(box) * np.pi *
                 np.diff(rp_bins**2) * 2 * pi_max)
    npairs = np.sum(npairs, axis=1).astype(np.float64)
    return (npairs / n_exp - 1) * 2 * pi_max


def pair_count_smu(sample1, s_bins, n_mu, sample2=None, period=None):
    """Ordered pair counts ``(n_s, n_mu)`` (uint64) in bins of the separation
    ``s`` and of ``mu = |dz| / s`` on ``[0, 1)``, as
    ``Corrfunc.theory.DDsmu(autocorr, 1, s_bins, 1, n_mu, ..., periodic=True)``
    reports them (``tabcorr/corrfunc.py:141-155``)."""
    lib = _lib.load()
    _lib.require_device()
    sample1 = _positions(sample1)
    s_bins = _lib.contiguous(np.asarray(s_bins, dtype=np.float64))
    box = _period(period)
    npairs = np.zeros((len(s_bins) - 1, int(n_mu)), dtype=np.uint64)
    if sample2 is not None:
        sample2 = _positions(sample2)
    _lib.check(lib.tc_pair_count_smu(
        _lib.as_double_p(sample1), len(sample1),
        _lib.as_double_p(sample2) if sample2 is not None else None,
        len(sample2) if sample2 is not None else 0, _lib.as_double_p(box),
        _lib.as_double_p(s_bins), len(s_bins) - 1, int(n_mu),
        npairs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    return npairs


def s_mu_tpcf(sample1, s_bins, mu_bins, sample2=None, period=None,
              do_auto=True, do_cross=False):
    """Drop-in for ``tabcorr.corrfunc.s_mu_tpcf``
    (``tabcorr/corrfunc.py:98-175``, itself a stand-in for
    ``halotools.mock_observables.s_mu_tpcf``): the redshift-space correlation
    function in bins of ``s`` and ``mu``.

    Raises
    ------
    ValueError
        If ``do_auto`` and ``do_cross`` have the same value or if ``mu_bins``
        are not uniform bins from 0 to 1.
    """
    if (do_auto and do_cross) or (not do_auto and not do_cross):
        raise ValueError("'do_auto' and 'do_cross' cannot both be True or " +
                         "False.")
    mu_bins = np.asarray(mu_bins, dtype=np.float64)
    if not np.all(np.isclose(mu_bins, np.linspace(0, 1, len(mu_bins)))):
        raise ValueError('Bins in mu must be uniform from 0 to 1.')
    s_bins = np.asarray(s_bins, dtype=np.float64)
    box = _period(period)
    n_mu = len(mu_bins) - 1
    if do_auto:
        npairs = pair_count_smu(sample1, s_bins, n_mu, None, box)
        n_exp = (len(sample1) * len(sample1) / np.prod(box) * 4 * np.pi / 3 *
                 np.diff(s_bins**3) / n_mu)
    else:
        npairs = pair_count_smu(sample1, s_bins, n_mu, sample2, box)
        n_exp = (len(sample1) * len(sample2) / np.prod(box) * 4 * np.pi / 3 *
                 np.diff(s_bins**3) / n_mu)
    return npairs.astype(np.float64) / n_exp[:, np.newaxis] - 1


def mass_in_cylinders(galaxies, particles, effective_particle_masses, rp_bins,
                      period=None):
    """Per-object mass of the particles within projected separation
    ``rp_bins[k]`` (``r = sqrt(dx^2 + dy^2) <= rp_bins[k]``, periodic in x and y,
    the line of sight spanning the box): ``(n_galaxies, len(rp_bins))`` -- the
    per-object weighted pair count behind
    ``halotools.mock_observables.mean_delta_sigma``.  ``effective_particle_masses``:
    a scalar or one mass per particle."""
    lib = _lib.load()
    _lib.require_device()
    galaxies = _positions(galaxies)
    particles = _positions(particles)
    rp_bins = _lib.contiguous(np.asarray(rp_bins, dtype=np.float64))
    box = _period(period)
    masses = np.asarray(effective_particle_masses, dtype=np.float64)
    scale = 1.0
    if masses.ndim == 0:
        # equal masses: integer counts on the device, one multiplication here
        scale, masses_p, keep = float(masses), None, None
    else:
        keep = _lib.contiguous(masses.ravel())
        if len(keep) != len(particles):
            raise ValueError('effective_particle_masses must be a scalar or '
                             'have one entry per particle.')
        masses_p = _lib.as_double_p(keep)
    out = np.zeros((len(galaxies), len(rp_bins)))
    _lib.check(lib.tc_mass_in_cylinders(
        _lib.as_double_p(galaxies), len(galaxies), _lib.as_double_p(particles),
        len(particles), masses_p, _lib.as_double_p(box),
        _lib.as_double_p(rp_bins), len(rp_bins), _lib.as_double_p(out)))
    del keep
    return out * scale if scale != 1.0 else out


def delta_sigma_from_mass_in_cylinders(mass_encl, rp_bins):
    """Excess surface density per object from the mass in cylinders, as
    halotools' ``mean_delta_sigma`` forms it (restated from its documentation;
    halotools is not part of the reference tree -- parity unpinned): at the
    area-weighted midpoint ``rp_mid = sqrt((r_lo^2 + r_hi^2) / 2)`` of every
    annulus, ``Sigma(< rp_mid)`` -- the surface density inside the cylinders
    ``mass_encl / (pi rp_bins^2)``, interpolated linearly in ``log rp`` --
    ``log Sigma`` over the radii with mass (0 inside the innermost of them) --
    minus the surface density of the annulus
    ``diff(mass_encl) / (pi diff(rp_bins^2))``: ``(n, len(rp_bins) - 1)``."""
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    mass_encl = np.atleast_2d(np.asarray(mass_encl, dtype=np.float64))
    rp_mids = np.sqrt(0.5 * (rp_bins[:-1]**2 + rp_bins[1:]**2))
    sigma_annulus = np.diff(mass_encl, axis=1) / (np.pi * np.diff(rp_bins**2))
    sigma_inside = mass_encl / (np.pi * rp_bins**2)
    log_rp, log_mid = np.log10(rp_bins), np.log10(rp_mids)
    interpolated = np.zeros_like(sigma_annulus)
    # objects with mass inside every cylinder (almost all): the midpoint of annulus k lies
    # between radii k and k + 1, so the interpolation is one vectorised expression
    full = np.all(sigma_inside > 0, axis=1)
    if np.any(full):
        log_sigma = np.log10(sigma_inside[full])
        slope = ((log_sigma[:, 1:] - log_sigma[:, :-1]) /
                 (log_rp[1:] - log_rp[:-1]))
        interpolated[full] = 10.0**(slope * (log_mid - log_rp[:-1]) +
                                    log_sigma[:, :-1])
    rest = np.nonzero(~full)[0]
    if len(rest):
        # objects with empty inner cylinders: linear interpolation in log-log between the
        # nearest radii WITH mass on either side of the midpoint (numpy.interp over the
        # masked radii, for all of them at once): the value at the last radius with mass
        # beyond it, 0 inside the first one or with fewer than two such radii
        mask = sigma_inside[rest] > 0
        n_radii = mask.shape[1]
        index = np.arange(n_radii)
        with np.errstate(divide='ignore'):
            log_sigma = np.where(mask, np.log10(np.where(mask, sigma_inside[rest], 1.0)),
                                 0.0)
        # midpoint k lies between radii k and k + 1
        left = np.maximum.accumulate(np.where(mask, index, -1), axis=1)[:, :-1]
        right = np.minimum.accumulate(np.where(mask, index, n_radii)[:, ::-1],
                                      axis=1)[:, ::-1][:, 1:]
        rows = np.arange(len(rest))[:, None]
        lo, hi = np.clip(left, 0, n_radii - 1), np.clip(right, 0, n_radii - 1)
        x_lo, x_hi = log_rp[lo], log_rp[hi]
        y_lo, y_hi = log_sigma[rows, lo], log_sigma[rows, hi]
        with np.errstate(divide='ignore', invalid='ignore'):
            slope = (y_hi - y_lo) / (x_hi - x_lo)
            value = 10.0**np.where(right < n_radii,
                                   slope * (log_mid - x_lo) + y_lo, y_lo)
        value[(left < 0) | (np.count_nonzero(mask, axis=1) < 2)[:, None]] = 0.0
        interpolated[rest] = value
    return interpolated - sigma_annulus


def mean_delta_sigma(galaxies, particles, effective_particle_masses, rp_bins,
                     period=None, per_object=False, **ignored):
    """Drop-in for ``halotools.mock_observables.mean_delta_sigma(galaxies,
    particles, effective_particle_masses, rp_bins, period=...)``: the mean
    excess surface density of the particles around the galaxies in annuli
    of ``rp_bins`` (``per_object=True``: one row per galaxy).  halotools'
    tuning arguments (``num_threads``, ``approx_cell*_size``, ``verbose``)
    are accepted and ignored."""
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    mass = mass_in_cylinders(galaxies, particles, effective_particle_masses,
                             rp_bins, period)
    result = delta_sigma_from_mass_in_cylinders(mass, rp_bins)
    return result if per_object else np.mean(result, axis=0)


def compute_tpcf_matrix_ds(mode, pos, period, particles,
                           effective_particle_masses, rp_bins):
    """``compute_tpcf_matrix`` of the reference (``tabcorr/tabcorr.py:846-922``)
    for ``tpcf = mean_delta_sigma`` (mode ``'cross'``, as
    ``scripts/tabulate_snapshot.py:228-237`` tabulates it): the halos of ALL
    bins go through one mass-in-cylinders pass instead of one call per bin;
    empty bins give zeros, as in the reference.  Returns ``(n_rp, G)`` and the
    shape ``(n_rp, )``."""
    if mode != 'cross':
        raise ValueError("mean_delta_sigma tables are tabulated in mode "
                         "'cross'.")
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    sizes = np.array([len(p) for p in pos], dtype=np.int64)
    n_r = len(rp_bins) - 1
    matrix = np.zeros((n_r, len(pos)))
    if sizes.sum() == 0:
        return matrix, (n_r, )
    galaxies = np.concatenate(
        [np.asarray(p, dtype=np.float64).reshape(-1, 3) for p in pos])
    mass = mass_in_cylinders(galaxies, particles, effective_particle_masses,
                             rp_bins, period)
    per_object = delta_sigma_from_mass_in_cylinders(mass, rp_bins)
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    for i in range(len(pos)):
        if sizes[i] > 0:
            matrix[:, i] = np.mean(per_object[offsets[i]:offsets[i + 1]],
                                   axis=0)
    return matrix, (n_r, )


def _labelled_points(pos, sample2):
    n_bins = len(pos)
    sizes = np.array([len(p) for p in pos], dtype=np.int64)
    points = _positions(np.concatenate(
        [np.asarray(p, dtype=np.float64).reshape(-1, 3) for p in pos]))
    label = np.ascontiguousarray(
        np.repeat(np.arange(n_bins, dtype=np.int32), sizes))
    int32_p = ctypes.POINTER(ctypes.c_int32)
    if sample2 is None:
        second = (None, None, 0)
        keep = ()
    else:
        sample2 = _positions(sample2)
        label2 = np.zeros(len(sample2), dtype=np.int32)
        second = (_lib.as_double_p(sample2), label2.ctypes.data_as(int32_p),
                  len(sample2))
        keep = (sample2, label2)
    first = (_lib.as_double_p(points), label.ctypes.data_as(int32_p),
             len(points))
    return first, second, (points, label) + keep


def pair_count_matrix(pos, rp_bins, pi_max, period, sample2=None):
    """Pair counts between all bins in one pass.

    ``pos`` is the list of per-bin position arrays ``TabCorr.tabulate`` builds
    (``tabcorr/tabcorr.py:318-330``).  Returns ``(n_rp, G, G)`` ordered pair
    counts between the points of every pair of bins (``sample2`` given:
    ``(n_rp, G)`` counts of every bin against that sample), summed over
    ``|pi| < pi_max``."""
    lib = _lib.load()
    _lib.require_device()
    n_bins = len(pos)
    rp_bins = _lib.contiguous(np.asarray(rp_bins, dtype=np.float64))
    box = _period(period)
    n_rp = len(rp_bins) - 1
    counts = np.zeros((n_rp, n_bins, n_bins), dtype=np.uint64)
    first, second, keep = _labelled_points(pos, sample2)
    _lib.check(lib.tc_pair_count_rppi_labelled(
        *first, *second, max(n_bins, 1), _lib.as_double_p(box),
        _lib.as_double_p(rp_bins), n_rp, float(pi_max),
        counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    del keep
    if sample2 is not None:
        return counts[:, :, 0]
    return counts


def pair_count_matrix_smu(pos, s_bins, n_mu, period, sample2=None):
    """`pair_count_matrix` in bins of ``s`` and ``mu``: ``(n_s, n_mu, G, G)``
    ordered pair counts (``sample2`` given: ``(n_s, n_mu, G)``)."""
    lib = _lib.load()
    _lib.require_device()
    n_bins = len(pos)
    s_bins = _lib.contiguous(np.asarray(s_bins, dtype=np.float64))
    box = _period(period)
    n_s = len(s_bins) - 1
    counts = np.zeros((n_s, int(n_mu), n_bins, n_bins), dtype=np.uint64)
    first, second, keep = _labelled_points(pos, sample2)
    _lib.check(lib.tc_pair_count_smu_labelled(
        *first, *second, max(n_bins, 1), _lib.as_double_p(box),
        _lib.as_double_p(s_bins), n_s, int(n_mu),
        counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
    del keep
    if sample2 is not None:
        return counts[:, :, :, 0]
    return counts


def compute_tpcf_matrix(mode, pos, period, rp_bins, pi_max, sample2=None):
    """``compute_tpcf_matrix`` of the reference (``tabcorr/tabcorr.py:846-922``)
    for ``tpcf = wp``: the correlation functions between all pairs of bins
    (mode ``'auto'``: ``(n_rp, G, G)``, symmetric) or of every bin with
    ``sample2`` (mode ``'cross'``: ``(n_rp, G)``), and the shape ``(n_rp, )``
    of one of them.  Empty bins give zeros, as in the reference."""
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    box = _period(period)
    sizes = np.array([len(p) for p in pos], dtype=np.float64)
    d_rp_sqr = np.diff(rp_bins**2)
    volume = np.prod(box)
    # (the products in the reference's order, tabcorr/corrfunc.py:72-73, 83-84)
    if mode == 'auto':
        counts = pair_count_matrix(pos, rp_bins, pi_max, box).astype(
            np.float64)
        n_exp = ((sizes[:, None] * sizes[None, :] / volume * np.pi)[None] *
                 d_rp_sqr[:, None, None] * 2 * pi_max)
    elif mode == 'cross':
        if sample2 is None:
            raise ValueError("mode 'cross' needs a second sample.")
        counts = pair_count_matrix(pos, rp_bins, pi_max, box,
                                   sample2=sample2).astype(np.float64)
        n_exp = ((sizes * len(sample2) / volume * np.pi)[None] *
                 d_rp_sqr[:, None] * 2 * pi_max)
    else:
        raise ValueError("mode must be 'auto' or 'cross'.")
    with np.errstate(divide='ignore', invalid='ignore'):
        matrix = (counts / n_exp - 1) * 2 * pi_max
    matrix[n_exp == 0] = 0.0          # bins without points: tabcorr.py:888
    return matrix, (len(rp_bins) - 1, )


def compute_tpcf_matrix_smu(mode, pos, period, s_bins, mu_bins, sample2=None):
    """`compute_tpcf_matrix` for ``tpcf = s_mu_tpcf``: ``(n_s n_mu, G, G)``
    (mode ``'auto'``) or ``(n_s n_mu, G)`` (mode ``'cross'``) and the shape
    ``(n_s, n_mu)``, flattened as the reference's ``xi.ravel()``
    (``tabcorr/tabcorr.py:900-915``)."""
    mu_bins = np.asarray(mu_bins, dtype=np.float64)
    if not np.all(np.isclose(mu_bins, np.linspace(0, 1, len(mu_bins)))):
        raise ValueError('Bins in mu must be uniform from 0 to 1.')
    s_bins = np.asarray(s_bins, dtype=np.float64)
    box = _period(period)
    n_mu = len(mu_bins) - 1
    n_s = len(s_bins) - 1
    sizes = np.array([len(p) for p in pos], dtype=np.float64)
    # (the products in the reference's order, tabcorr/corrfunc.py:151-153, 161-163)
    shell = np.diff(s_bins**3)
    volume = np.prod(box)
    if mode == 'auto':
        counts = pair_count_matrix_smu(pos, s_bins, n_mu, box).astype(
            np.float64)
        n_exp = ((sizes[:, None] * sizes[None, :] / volume * 4 * np.pi /
                  3)[None] * shell[:, None, None] / n_mu)[:, None]
    elif mode == 'cross':
        if sample2 is None:
            raise ValueError("mode 'cross' needs a second sample.")
        counts = pair_count_matrix_smu(pos, s_bins, n_mu, box,
                                       sample2=sample2).astype(np.float64)
        n_exp = ((sizes * len(sample2) / volume * 4 * np.pi / 3)[None] *
                 shell[:, None] / n_mu)[:, None]
    else:
        raise ValueError("mode must be 'auto' or 'cross'.")
    with np.errstate(divide='ignore', invalid='ignore'):
        matrix = counts / n_exp - 1
    matrix[np.broadcast_to(n_exp == 0, matrix.shape)] = 0.0   # empty bins
    return matrix.reshape((n_s * n_mu, ) + matrix.shape[2:]), (n_s, n_mu)


def reference_compute_tpcf_matrix(mode, pos, tpcf, period, tpcf_args,
                                  tpcf_kwargs, num_threads=1, verbose=False):
    """`compute_tpcf_matrix` with the reference's own signature
    (``tabcorr/tabcorr.py:846-848``) for ``tpcf = wp``, ``s_mu_tpcf`` or
    ``mean_delta_sigma``: what
    ``TabCorr.tabulate`` swaps in for the reference's pool of per-pair calls.
    ``tpcf_args`` are ``(rp_bins, pi_max)`` / ``(s_bins, mu_bins)``; mode
    ``'cross'`` takes ``sample2`` from ``tpcf_kwargs`` as the reference's call
    at ``:841-844`` would."""
    if tpcf is s_mu_tpcf:
        return compute_tpcf_matrix_smu(mode, pos, period, tpcf_args[0],
                                       tpcf_args[1],
                                       sample2=tpcf_kwargs.get('sample2'))
    if tpcf is mean_delta_sigma:
        # tpcf_args = (particle positions, particle masses, rp_bins)
        return compute_tpcf_matrix_ds(mode, pos, period, tpcf_args[0],
                                      tpcf_args[1], tpcf_args[2])
    if tpcf is not wp:
        raise ValueError('Only tabcorr_amd.corrfunc.wp, s_mu_tpcf and '
                         'mean_delta_sigma are counted on the GPU.')
    rp_bins, pi_max = tpcf_args[0], tpcf_args[1]
    return compute_tpcf_matrix(mode, pos, period, rp_bins, pi_max,
                               sample2=tpcf_kwargs.get('sample2'))
