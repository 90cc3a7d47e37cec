"""CPU tests of the pair-count oracle (oracle/paircount_oracle.py): known answers on
hand-made point sets, invariants of the counts, and the wrapper arithmetic of
tabcorr/corrfunc.py:6-95 / tabcorr/tabcorr.py:846-922."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
from oracle import paircount_oracle as oracle  # noqa: E402


def test_known_answers_on_hand_made_points():
    box = (10.0, 10.0, 10.0)
    rp_bins = np.array([0.5, 1.5, 3.0])
    # two points 1 apart in x, 0.2 apart in z; a third one across the periodic boundary
    pos = np.array([[1.0, 1.0, 1.0], [2.0, 1.0, 1.2], [9.5, 1.0, 9.9]])
    counts = oracle.pair_count_rppi(pos, None, box, rp_bins, pi_max=2.0, n_pi=2)
    # pairs (ordered): 0-1: rp = 1, dz = 0.2 -> bin (0, 0), twice;
    # 0-2: dx = 1.5 (wrapped), dz = 1.1 (wrapped) -> rp bin 1, pi bin 1, twice;
    # 1-2: dx = 2.5, dz = 1.3 -> rp bin 1, pi bin 1, twice
    expect = np.array([[2, 0], [0, 4]], dtype=np.uint64)
    assert np.array_equal(counts, expect)
    # self pairs only when the first edge is 0
    counts0 = oracle.pair_count_rppi(pos, None, box, np.array([0.0, 1.5, 3.0]), 2.0, 2)
    assert counts0[0, 0] == 2 + 3
    # pimax cuts: dz = 1.1 and 1.3 fall out
    assert oracle.pair_count_rppi(pos, None, box, rp_bins, 1.0, 1).sum() == 2
    # edges are half-open: rp exactly on an inner edge goes up, on the last edge out
    pos = np.array([[0.0, 0.0, 5.0], [1.5, 0.0, 5.0], [3.0, 0.0, 5.0]])
    counts = oracle.pair_count_rppi(pos, None, box, rp_bins, 1.0, 1)
    assert counts[:, 0].tolist() == [0, 4]          # 0-1 and 1-2 at rp = 1.5; 0-2 at 3.0 out


def test_count_invariants():
    rng = np.random.default_rng(5)
    box = np.array([40.0, 50.0, 60.0])
    pos1 = rng.uniform(0, 1, (300, 3)) * box
    pos2 = rng.uniform(0, 1, (200, 3)) * box
    rp_bins = np.logspace(-0.5, 1.0, 8)
    auto = oracle.pair_count_rppi(pos1, None, box, rp_bins, 12.0)
    assert auto.shape == (7, 12) and np.all(auto % 2 == 0)          # every pair twice
    cross = oracle.pair_count_rppi(pos1, pos2, box, rp_bins, 12.0)
    swapped = oracle.pair_count_rppi(pos2, pos1, box, rp_bins, 12.0)
    assert np.array_equal(cross, swapped)
    # a rigid shift through the periodic boundary moves no pair across an edge here
    shifted = (pos1 + np.array([17.0, 33.0, 41.0])) % box
    assert np.abs(oracle.pair_count_rppi(shifted, None, box, rp_bins, 12.0).astype(int)
                  - auto.astype(int)).sum() <= 4
    # labelled counts: the bin-pair blocks are the pair counts of the sub-samples
    label = rng.integers(0, 4, len(pos1))
    matrix = oracle.pair_count_rppi(pos1, None, box, rp_bins, 12.0, label1=label,
                                    n_labels=4)
    assert matrix.shape == (7, 4, 4)
    assert np.array_equal(matrix, matrix.transpose(0, 2, 1))
    assert np.array_equal(matrix.sum(axis=(1, 2)), auto.sum(axis=1))
    for a in range(4):
        for b in range(4):
            sub = oracle.pair_count_rppi(pos1[label == a], None if a == b else pos1[label == b],
                                         box, rp_bins, 12.0)
            assert np.array_equal(sub.sum(axis=1), matrix[:, a, b])


def test_wp_of_a_poisson_sample_is_zero_and_matrix_matches_pairwise_calls():
    rng = np.random.default_rng(11)
    box = 100.0
    pos = rng.uniform(0, box, (3000, 3))
    rp_bins = np.array([2.0, 5.0, 10.0, 20.0])
    wp = oracle.wp(pos, rp_bins, 20.0, period=box)
    # (npairs / n_exp - 1) 2 pi_max with ~1e4 .. 2e5 pairs per bin
    assert np.all(np.abs(wp) < 40.0 * 5.0 / np.sqrt(3000 * 3000 / box**3 * np.pi *
                                                   np.diff(rp_bins**2) * 40.0))
    with pytest.raises(ValueError):
        oracle.wp(pos, rp_bins, 20.0, period=box, do_auto=True, do_cross=True)
    bins = [pos[:100], pos[100:100], pos[100:700], pos[700:1000]]      # one empty bin
    matrix, shape = oracle.compute_tpcf_matrix_wp('auto', bins, box, rp_bins, 20.0)
    assert shape == (3, ) and matrix.shape == (3, 4, 4)
    assert np.all(matrix[:, 1, :] == 0) and np.all(matrix[:, :, 1] == 0)
    assert np.array_equal(matrix, matrix.transpose(0, 2, 1))
    assert np.array_equal(matrix[:, 2, 2], oracle.wp(bins[2], rp_bins, 20.0, period=box))
    assert np.array_equal(matrix[:, 0, 3], oracle.wp(bins[0], rp_bins, 20.0, sample2=bins[3],
                                                    period=box, do_auto=False, do_cross=True))
    cross, _ = oracle.compute_tpcf_matrix_wp('cross', bins, box, rp_bins, 20.0,
                                             sample2=pos[1000:])
    assert cross.shape == (3, 4) and np.all(cross[:, 1] == 0)


def test_s_mu_counts():
    box = (20.0, 20.0, 20.0)
    s_bins = np.array([0.5, 2.0, 5.0])
    # s = 1 along x (mu = 0); s = 5 along z (mu = 1: never counted, and on the last edge);
    # (3, 0, 4) from the first point: s = 5 out; (0.6, 0, 0.8): s = 1, mu = 0.8
    pos = np.array([[1.0, 1.0, 1.0], [2.0, 1.0, 1.0], [1.0, 1.0, 19.5], [1.6, 1.0, 1.8]])
    counts = oracle.pair_count_smu(pos, None, box, s_bins, 5)
    assert counts.shape == (2, 5)
    # pairs: 0-1 (s 1, mu 0) bin (0, 0); 0-3 (s 1, mu .8) bin (0, 4); 1-3: d = (.4, 0, .8),
    # s = .894, mu = .894 -> (0, 4); 0-2: dz = 1.5 wrapped, s = 1.5, mu = 1 -> out;
    # 1-2: d = (1, 0, 1.5), s = 1.80, mu = .83 -> (0, 4); 2-3: d = (.6, 0, 2.3 wrapped): s = 2.38,
    # mu = .968 -> (1, 4)
    expect = np.zeros((2, 5), dtype=np.uint64)
    expect[0, 0] = 2
    expect[0, 4] = 6
    expect[1, 4] = 2
    assert np.array_equal(counts, expect)
    rng = np.random.default_rng(8)
    box = 60.0
    pos1, pos2 = rng.uniform(0, box, (400, 3)), rng.uniform(0, box, (300, 3))
    s_bins = np.logspace(-0.3, 1.2, 7)
    auto = oracle.pair_count_smu(pos1, None, box, s_bins, 10)
    assert np.all(auto % 2 == 0)
    assert np.array_equal(oracle.pair_count_smu(pos1, pos2, box, s_bins, 10),
                          oracle.pair_count_smu(pos2, pos1, box, s_bins, 10))
    # summed over mu the counts are those of spherical shells: a Poisson sample has xi ~ 0
    xi = oracle.s_mu_tpcf(rng.uniform(0, box, (2500, 3)), s_bins, np.linspace(0, 1, 5),
                          period=box)
    assert xi.shape == (6, 4) and np.all(np.abs(xi[3:]) < 0.2)
    with pytest.raises(ValueError, match='uniform'):
        oracle.s_mu_tpcf(pos1, s_bins, np.array([0.0, 0.3, 1.0]), period=box)


def test_s_mu_matrix_matches_pairwise_calls():
    """compute_tpcf_matrix_smu (tabcorr/tabcorr.py:846-922 with tpcf = s_mu_tpcf): symmetric,
    zero rows for empty bins, entries equal to the single-pair calls, flattened as
    xi.ravel()."""
    rng = np.random.default_rng(12)
    box = 50.0
    s_bins = np.array([0.5, 2.0, 6.0, 15.0])
    mu_bins = np.linspace(0, 1, 4)
    points = rng.uniform(0, box, (900, 3))
    pos = [points[:200], points[200:200], points[200:500], points[500:]]
    matrix, shape = oracle.compute_tpcf_matrix_smu('auto', pos, box, s_bins, mu_bins)
    assert shape == (3, 3) and matrix.shape == (9, 4, 4)
    assert np.array_equal(matrix, matrix.transpose(0, 2, 1))
    assert np.all(matrix[:, 1] == 0) and np.all(matrix[:, :, 1] == 0)
    assert np.array_equal(matrix[:, 2, 2],
                          oracle.s_mu_tpcf(pos[2], s_bins, mu_bins, period=box).ravel())
    assert np.array_equal(
        matrix[:, 0, 3],
        oracle.s_mu_tpcf(pos[0], s_bins, mu_bins, sample2=pos[3], period=box,
                         do_auto=False, do_cross=True).ravel())
    other = rng.uniform(0, box, (300, 3))
    cross, shape = oracle.compute_tpcf_matrix_smu('cross', pos, box, s_bins, mu_bins,
                                                  sample2=other)
    assert shape == (3, 3) and cross.shape == (9, 4) and np.all(cross[:, 1] == 0)
    assert np.array_equal(
        cross[:, 3],
        oracle.s_mu_tpcf(pos[3], s_bins, mu_bins, sample2=other, period=box,
                         do_auto=False, do_cross=True).ravel())


# ---- pinned against the reference's own wrappers -------------------------------------------

def _wrapper_fixture():
    from util import load_golden
    data = load_golden('paircount_wrappers')
    offsets = np.concatenate([[0], np.cumsum(data['sizes'])])
    pos = [data['pos'][a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    return data, pos


def test_oracle_wrappers_match_the_reference_fixture():
    """tests/golden/paircount_wrappers.npz was recorded by running the reference's
    tabcorr.corrfunc.wp / s_mu_tpcf and tabcorr.tabcorr.compute_tpcf_matrix (unmodified,
    imported from /root/reference) over a stub Corrfunc.theory with an independent
    brute-force counter (tests/golden/make_golden.py: golden_paircount).  The oracle's
    restatement of the wrapper arithmetic and of the matrix assembly -- task list, swap of
    the samples by size, symmetrisation, empty bin -- must reproduce it."""
    data, pos = _wrapper_fixture()
    period, rp_bins, pi_max = data['period'], data['rp_bins'], float(data['pi_max'])
    s_bins, mu_bins = data['s_bins'], data['mu_bins']
    close = dict(rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(oracle.wp(pos[0], rp_bins, pi_max, period=period),
                               data['wp_auto'], **close)
    np.testing.assert_allclose(
        oracle.wp(pos[0], rp_bins, pi_max, sample2=pos[3], period=period, do_auto=False,
                  do_cross=True), data['wp_cross'], **close)
    np.testing.assert_allclose(
        oracle.wp(np.mod(pos[3], 60.0), rp_bins, pi_max, period=60.0),
        data['wp_auto_scalar_period'], **close)
    np.testing.assert_allclose(oracle.s_mu_tpcf(pos[0], s_bins, mu_bins, period=period),
                               data['smu_auto'], **close)
    np.testing.assert_allclose(
        oracle.s_mu_tpcf(pos[0], s_bins, mu_bins, sample2=pos[3], period=period,
                         do_auto=False, do_cross=True), data['smu_cross'], **close)
    matrix, shape = oracle.compute_tpcf_matrix_wp('auto', pos, period, rp_bins, pi_max)
    assert tuple(shape) == tuple(data['shape_auto_wp'])
    np.testing.assert_allclose(matrix, data['matrix_auto_wp'], **close)
    assert np.all(matrix[:, 2] == 0) and np.all(matrix[:, :, 2] == 0)      # the empty bin
    matrix, shape = oracle.compute_tpcf_matrix_wp('cross', pos, period, rp_bins, pi_max,
                                                  sample2=data['particles'])
    assert tuple(shape) == tuple(data['shape_cross_wp'])
    np.testing.assert_allclose(matrix, data['matrix_cross_wp'], **close)
    matrix, shape = oracle.compute_tpcf_matrix_smu('auto', pos, period, s_bins, mu_bins)
    assert tuple(shape) == tuple(data['shape_auto_smu'])
    np.testing.assert_allclose(matrix, data['matrix_auto_smu'], **close)
    matrix, shape = oracle.compute_tpcf_matrix_smu('cross', pos, period, s_bins, mu_bins,
                                                   sample2=data['particles'])
    assert tuple(shape) == tuple(data['shape_cross_smu'])
    np.testing.assert_allclose(matrix, data['matrix_cross_smu'], **close)


def test_mass_in_cylinders_and_delta_sigma_known_answers():
    box = (20.0, 20.0, 50.0)
    rp_bins = np.array([1.0, 2.0, 4.0])
    galaxies = np.array([[5.0, 5.0, 1.0], [19.5, 10.0, 40.0]])
    # around galaxy 0: one particle at r = 0.5, two at r = 1.5 (any z), one at r = 3;
    # around galaxy 1: one particle across the periodic boundary at r = 1 (edge: r <= 1)
    particles = np.array([[5.5, 5.0, 30.0], [5.0, 6.5, 2.0], [3.5, 5.0, 49.0], [5.0, 2.0, 7.0],
                          [0.5, 10.0, 3.0]])
    mass = oracle.mass_in_cylinders(galaxies, particles, 2.0, rp_bins, box)
    assert mass.tolist() == [[2.0, 6.0, 8.0], [2.0, 2.0, 2.0]]
    weights = np.array([1.0, 10.0, 100.0, 1000.0, 5.0])
    mass = oracle.mass_in_cylinders(galaxies, particles, weights, rp_bins, box)
    assert mass.tolist() == [[1.0, 111.0, 1111.0], [5.0, 5.0, 5.0]]
    # a uniform sheet: Sigma(< R) = Sigma(R), so Delta Sigma = 0 up to shot noise
    rng = np.random.default_rng(0)
    sheet = rng.uniform(0, 1, (40000, 3)) * np.array(box)
    ds = oracle.mean_delta_sigma(rng.uniform(0, 1, (50, 3)) * np.array(box), sheet, 1.0,
                                 np.array([1.0, 2.0, 4.0, 8.0]), box)
    assert np.all(np.abs(ds) < 0.1 * 40000 / 400.0)
    # a point mass M at the centre of a galaxy: Sigma(< R) = M / (pi R^2), Sigma(R) = 0
    ds = oracle.mean_delta_sigma(np.array([[10.0, 10.0, 0.0]]), np.array([[10.0, 10.0, 9.0]]),
                                 3.0, rp_bins, box)
    mids = np.sqrt(0.5 * (rp_bins[:-1]**2 + rp_bins[1:]**2))
    np.testing.assert_allclose(ds, 3.0 / (np.pi * mids**2), rtol=1e-12)
