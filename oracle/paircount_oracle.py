"""CPU oracle of the pair counts behind ``TabCorr.tabulate`` (brute force, NumPy).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as
``oracle/tabcorr_oracle.py``: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s CPU baseline legs may import it).

What it restates
----------------
``tabcorr/corrfunc.py:6-95`` (``wp``) turns ``Corrfunc.theory.DDrppi`` pair
counts into a projected correlation function, and
``tabcorr/tabcorr.py:846-922`` (``compute_tpcf_matrix``) calls it once per
pair of halo bins.  The wrapper arithmetic (`wp`, `compute_tpcf_matrix_wp`)
follows the reference line by line.

PARITY UNPINNED for the pair counter itself: Corrfunc (``pyproject.toml``:
optional, unpinned; not vendored, not installed here) is a third-party
library.  `pair_count_rppi` restates its documented ``DDrppi`` semantics
(theory/DDrppi of Corrfunc 2.x): ordered pairs (an auto-count holds every
pair twice), periodic box, a pair counts when ``|dz| < pimax`` and
``rp_bins[0]^2 <= dx^2 + dy^2 < rp_bins[-1]^2``, squared bin edges, ``int(pimax)``
line-of-sight bins of equal width, self-pairs only when ``rp_bins[0] == 0``.
Corrfunc shifts whole cells by the box size where this restatement takes the
minimum image of every coordinate difference, so a pair within one rounding
error of a bin edge may land differently; everything else is integer
arithmetic.  The HIP kernel (``tabcorr_amd/csrc/paircount.hip``) is bit-exact
against THIS file.
"""

import numpy as np


def _min_image(d, box):
    half = 0.5 * box
    d = np.where(d > half, d - box, d)
    return np.where(d < -half, d + box, d)


def pair_count_rppi(pos1, pos2, boxsize, rp_bins, pi_max, n_pi=None,
                    label1=None, label2=None, n_labels=0, chunk=512):
    """Ordered pair counts ``(n_rp, n_pi)`` -- or, with labels,
    ``(n_rp, n_labels, n_labels)`` summed over the line of sight -- between
    ``pos1`` and ``pos2`` (``None``: ``pos1`` with itself)."""
    pos1 = np.asarray(pos1, dtype=np.float64).reshape(-1, 3)
    if pos2 is None:
        pos2, label2 = pos1, label1
    pos2 = np.asarray(pos2, dtype=np.float64).reshape(-1, 3)
    boxsize = np.broadcast_to(np.asarray(boxsize, dtype=np.float64), (3, ))
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    n_rp = len(rp_bins) - 1
    if n_pi is None:
        n_pi = int(pi_max)
    edge_sqr = rp_bins * rp_bins
    inv_dpi = float(n_pi) / float(pi_max)
    labelled = n_labels > 0
    if labelled:
        counts = np.zeros(n_rp * n_labels * n_labels, dtype=np.uint64)
        label1 = np.asarray(label1, dtype=np.int64)
        label2 = np.asarray(label2, dtype=np.int64)
        n_pi = 1
    else:
        counts = np.zeros(n_rp * n_pi, dtype=np.uint64)
    for begin in range(0, len(pos1), chunk):
        a = pos1[begin:begin + chunk]
        dz = np.abs(_min_image(a[:, None, 2] - pos2[None, :, 2], boxsize[2]))
        dx = _min_image(a[:, None, 0] - pos2[None, :, 0], boxsize[0])
        dy = _min_image(a[:, None, 1] - pos2[None, :, 1], boxsize[1])
        r_sqr = dx * dx + dy * dy
        keep = (dz < pi_max) & (r_sqr >= edge_sqr[0]) & (r_sqr < edge_sqr[-1])
        rp_bin = np.searchsorted(edge_sqr, r_sqr[keep], side='right') - 1
        if labelled:
            i, j = np.nonzero(keep)
            flat = (rp_bin * n_labels + label1[begin + i]) * n_labels + \
                label2[j]
        else:
            pi_bin = (dz[keep] * inv_dpi).astype(np.int64)
            inside = pi_bin < n_pi
            flat = rp_bin[inside] * n_pi + pi_bin[inside]
        counts += np.bincount(flat, minlength=len(counts)).astype(np.uint64)
    if labelled:
        return counts.reshape(n_rp, n_labels, n_labels)
    return counts.reshape(n_rp, n_pi)


def pair_count_smu(pos1, pos2, boxsize, s_bins, n_mu, chunk=512):
    """Ordered pair counts ``(n_s, n_mu)`` in bins of the separation ``s`` and
    of ``mu = |dz| / s`` on ``[0, 1)`` -- ``Corrfunc.theory.DDsmu`` as called at
    ``tabcorr/corrfunc.py:141-155``: ``s_bins[0]^2 <= s^2 < s_bins[-1]^2`` with
    ``s^2 = (dx^2 + dy^2) + dz^2``, ``mu < 1``, mu bin ``int(mu n_mu)``; a pair at
    zero separation (self-pairs when ``s_bins[0] == 0``) goes to mu bin 0."""
    pos1 = np.asarray(pos1, dtype=np.float64).reshape(-1, 3)
    if pos2 is None:
        pos2 = pos1
    pos2 = np.asarray(pos2, dtype=np.float64).reshape(-1, 3)
    boxsize = np.broadcast_to(np.asarray(boxsize, dtype=np.float64), (3, ))
    s_bins = np.asarray(s_bins, dtype=np.float64)
    n_s = len(s_bins) - 1
    edge_sqr = s_bins * s_bins
    counts = np.zeros(n_s * n_mu, dtype=np.uint64)
    for begin in range(0, len(pos1), chunk):
        a = pos1[begin:begin + chunk]
        dz = np.abs(_min_image(a[:, None, 2] - pos2[None, :, 2], boxsize[2]))
        dx = _min_image(a[:, None, 0] - pos2[None, :, 0], boxsize[0])
        dy = _min_image(a[:, None, 1] - pos2[None, :, 1], boxsize[1])
        s_sqr = (dx * dx + dy * dy) + dz * dz
        keep = (s_sqr >= edge_sqr[0]) & (s_sqr < edge_sqr[-1])
        s_sqr, dz = s_sqr[keep], dz[keep]
        with np.errstate(invalid='ignore', divide='ignore'):
            mu = np.where(s_sqr > 0, dz / np.sqrt(s_sqr), 0.0)
        inside = mu < 1.0
        s_bin = np.searchsorted(edge_sqr, s_sqr[inside], side='right') - 1
        mu_bin = (mu[inside] * float(n_mu)).astype(np.int64)
        ok = mu_bin < n_mu
        counts += np.bincount(s_bin[ok] * n_mu + mu_bin[ok],
                              minlength=len(counts)).astype(np.uint64)
    return counts.reshape(n_s, n_mu)


def s_mu_tpcf(sample1, s_bins, mu_bins, sample2=None, period=None,
              do_auto=True, do_cross=False):
    """``tabcorr/corrfunc.py:98-175`` with `pair_count_smu` in the place of
    ``Corrfunc.theory.DDsmu``."""
    if (do_auto and do_cross) or (not do_auto and not do_cross):
        raise ValueError("'do_auto' and 'do_cross' cannot both be True or " +
                         "False.")
    mu_bins = np.asarray(mu_bins, dtype=np.float64)
    if not np.all(np.isclose(mu_bins, np.linspace(0, 1, len(mu_bins)))):
        raise ValueError('Bins in mu must be uniform from 0 to 1.')  # :135-138
    s_bins = np.asarray(s_bins, dtype=np.float64)
    if isinstance(period, (float, int)):
        period = (period, period, period)
    period = tuple(np.asarray(period, dtype=np.float64))
    n_mu = len(mu_bins) - 1
    if do_auto:                                                    # :146-153
        npairs = pair_count_smu(sample1, None, period, s_bins, n_mu)
        n_exp = (len(sample1) * len(sample1) / np.prod(period) * 4 *
                 np.pi / 3 * np.diff(s_bins**3) / n_mu)
    else:                                                          # :155-163
        npairs = pair_count_smu(sample1, sample2, period, s_bins, n_mu)
        n_exp = (len(sample1) * len(sample2) / np.prod(period) * 4 *
                 np.pi / 3 * np.diff(s_bins**3) / n_mu)
    return npairs.astype(np.float64) / n_exp[:, np.newaxis] - 1     # :165-166


def wp(sample1, rp_bins, pi_max, sample2=None, period=None, do_auto=True,
       do_cross=False):
    """``tabcorr/corrfunc.py:6-95`` with `pair_count_rppi` in the place of
    ``Corrfunc.theory.DDrppi``."""
    if (do_auto and do_cross) or (not do_auto and not do_cross):
        raise ValueError("'do_auto' and 'do_cross' cannot both be True or " +
                         "False.")
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    if isinstance(period, (float, int)):
        period = (period, period, period)
    period = tuple(np.asarray(period, dtype=np.float64))
    if do_auto:                                                    # :67-73
        npairs = pair_count_rppi(sample1, None, period, rp_bins, pi_max)
        n_exp = (len(sample1) * len(sample1) / np.prod(period) * np.pi *
                 np.diff(rp_bins**2) * 2 * pi_max)
    else:                                                          # :76-84
        npairs = pair_count_rppi(sample1, sample2, period, rp_bins, pi_max)
        n_exp = (len(sample1) * len(sample2) / np.prod(period) * np.pi *
                 np.diff(rp_bins**2) * 2 * pi_max)
    npairs = np.sum(npairs, axis=1).astype(np.float64)             # :86-87
    return (npairs / n_exp - 1) * 2 * pi_max                       # :89


def compute_tpcf_matrix_wp(mode, pos, period, rp_bins, pi_max, sample2=None):
    """``tabcorr/tabcorr.py:846-922`` for ``tpcf = wp``: one `wp` call per
    pair of non-empty bins (mode 'auto') or per bin against ``sample2``
    (mode 'cross'), in the reference's own loop structure."""
    n_r = len(rp_bins) - 1
    tasks = [i for i in range(len(pos)) if len(pos[i]) > 0]        # :888
    if mode == 'auto':
        matrix = np.zeros((n_r, len(pos), len(pos)))
        for a, i_1 in enumerate(tasks):
            for i_2 in tasks[a:]:                                  # :890-891
                j_1, j_2 = i_1, i_2
                if len(pos[j_1]) > len(pos[j_2]):                  # :838-839
                    j_1, j_2 = j_2, j_1
                xi = wp(pos[j_1], rp_bins, pi_max,
                        sample2=pos[j_2] if j_1 != j_2 else None,
                        do_auto=(j_1 == j_2), do_cross=(j_1 != j_2),
                        period=period)                             # :840-843
                matrix[:, i_1, i_2] += xi.ravel()                  # :910-912
                matrix[:, i_2, i_1] = matrix[:, i_1, i_2]
    else:
        matrix = np.zeros((n_r, len(pos)))
        for i in tasks:
            matrix[:, i] += wp(pos[i], rp_bins, pi_max, sample2=sample2,
                               do_auto=False, do_cross=True, period=period)
    return matrix, (n_r, )


def compute_tpcf_matrix_smu(mode, pos, period, s_bins, mu_bins, sample2=None):
    """``tabcorr/tabcorr.py:846-922`` for ``tpcf = s_mu_tpcf``
    (``tabcorr/corrfunc.py:98-175``): one call per pair of non-empty bins, the
    ``(n_s, n_mu)`` results flattened as the reference's ``xi.ravel()`` does
    (``:900-915``)."""
    shape = (len(s_bins) - 1, len(mu_bins) - 1)
    tasks = [i for i in range(len(pos)) if len(pos[i]) > 0]        # :888
    if mode == 'auto':
        matrix = np.zeros((shape[0] * shape[1], len(pos), len(pos)))
        for a, i_1 in enumerate(tasks):
            for i_2 in tasks[a:]:                                  # :890-891
                j_1, j_2 = i_1, i_2
                if len(pos[j_1]) > len(pos[j_2]):                  # :838-839
                    j_1, j_2 = j_2, j_1
                xi = s_mu_tpcf(pos[j_1], s_bins, mu_bins,
                               sample2=pos[j_2] if j_1 != j_2 else None,
                               do_auto=(j_1 == j_2), do_cross=(j_1 != j_2),
                               period=period)                      # :840-843
                matrix[:, i_1, i_2] += xi.ravel()                  # :910-912
                matrix[:, i_2, i_1] = matrix[:, i_1, i_2]
    else:
        matrix = np.zeros((shape[0] * shape[1], len(pos)))
        for i in tasks:
            matrix[:, i] += s_mu_tpcf(pos[i], s_bins, mu_bins, sample2=sample2,
                                      do_auto=False, do_cross=True,
                                      period=period).ravel()
    return matrix, shape


# ---- excess surface density (halotools' mean_delta_sigma; PARITY UNPINNED) --------------------
#
# halotools (``pyproject.toml``: unpinned, not vendored, not installed here) provides
# ``mean_delta_sigma(galaxies, particles, effective_particle_masses, rp_bins, period)``, the
# two-point function of the reference's ds tables (``scripts/tabulate_snapshot.py:228-237``;
# called per halo bin at ``tabcorr/tabcorr.py:844``).  Restated from its documentation and
# the method it cites (the mass in cylinders of radius rp_bins[k] around every object, then
# Delta Sigma = Sigma(< R) - Sigma(R) at the area-weighted midpoints of the annuli, Sigma(< R)
# log-interpolated between the cylinder radii).  Nothing in the reference tree pins its
# numbers; the HIP kernel is checked against THIS restatement (exactly, for equal masses).

def mass_in_cylinders(galaxies, particles, masses, rp_bins, period, chunk=256):
    """(n_galaxies, len(rp_bins)) summed particle mass within projected separation
    ``r <= rp_bins[k]`` (compared squared; minimum image in x and y)."""
    galaxies = np.asarray(galaxies, dtype=np.float64).reshape(-1, 3)
    particles = np.asarray(particles, dtype=np.float64).reshape(-1, 3)
    period = np.broadcast_to(np.asarray(period, dtype=np.float64), (3, ))
    masses = np.broadcast_to(np.asarray(masses, dtype=np.float64), (len(particles), ))
    edge_sqr = np.asarray(rp_bins, dtype=np.float64)**2
    out = np.zeros((len(galaxies), len(edge_sqr)))
    for begin in range(0, len(galaxies), chunk):
        a = galaxies[begin:begin + chunk]
        dx = _min_image(a[:, None, 0] - particles[None, :, 0], period[0])
        dy = _min_image(a[:, None, 1] - particles[None, :, 1], period[1])
        r_sqr = dx * dx + dy * dy
        for k, edge in enumerate(edge_sqr):
            out[begin:begin + chunk, k] = np.sum(np.where(r_sqr <= edge, masses[None, :], 0.0),
                                                 axis=1)
    return out


def mean_delta_sigma(galaxies, particles, masses, rp_bins, period, per_object=False):
    rp_bins = np.asarray(rp_bins, dtype=np.float64)
    mass = mass_in_cylinders(galaxies, particles, masses, rp_bins, period)
    rp_mids = np.sqrt(0.5 * (rp_bins[:-1]**2 + rp_bins[1:]**2))
    annulus = (mass[:, 1:] - mass[:, :-1]) / (np.pi * (rp_bins[1:]**2 - rp_bins[:-1]**2))
    inside = mass / (np.pi * rp_bins**2)
    result = np.zeros_like(annulus)
    for g in range(len(mass)):
        have = np.nonzero(inside[g] > 0)[0]
        if len(have) >= 2:
            interp = 10.0**np.interp(np.log10(rp_mids), np.log10(rp_bins[have]),
                                     np.log10(inside[g][have]))
            interp[rp_mids < rp_bins[have[0]]] = 0.0
            result[g] = interp
    result -= annulus
    return result if per_object else result.mean(axis=0)


def compute_tpcf_matrix_ds(pos, period, particles, masses, rp_bins):
    """``tabcorr/tabcorr.py:846-922`` in mode 'cross' for ``tpcf = mean_delta_sigma``: one
    call per non-empty bin (``:844``), columns of empty bins stay zero (``:888, 903``)."""
    n_r = len(rp_bins) - 1
    matrix = np.zeros((n_r, len(pos)))
    for i in range(len(pos)):
        if len(pos[i]) > 0:
            matrix[:, i] += mean_delta_sigma(pos[i], particles, masses, rp_bins, period)
    return matrix, (n_r, )
