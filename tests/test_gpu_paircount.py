"""GPU pair counting (tabcorr_amd/csrc/paircount.hip through the C ABI) against the
brute-force oracle (oracle/paircount_oracle.py): integer counts, so the bar is bit-exact.
Seeded point sets: uniform and strongly clustered, cubic and non-cubic boxes, grids with one
cell in a dimension, points on the box faces, auto / cross / labelled counts, and the
tabulation wrappers of tabcorr/corrfunc.py:6-95 and tabcorr/tabcorr.py:846-922."""

import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))

pytestmark = pytest.mark.gpu


def clustered(rng, n, box, n_centres=40, scale=1.5):
    centres = rng.uniform(0, 1, (n_centres, 3)) * box
    pos = centres[rng.integers(0, n_centres, n)] + rng.normal(0, scale, (n, 3))
    return np.mod(pos, box)


@pytest.mark.parametrize('case', ['uniform', 'clustered', 'flat box', 'faces'])
def test_pair_counts_are_exact(case):
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    rng = np.random.default_rng({'uniform': 1, 'clustered': 2, 'flat box': 3, 'faces': 4}[case])
    box = np.array([120.0, 120.0, 120.0])
    rp_bins = np.logspace(-1, np.log10(25.0), 13)
    pi_max = 40.0
    if case == 'uniform':
        pos1, pos2 = rng.uniform(0, 1, (5000, 3)) * box, rng.uniform(0, 1, (3000, 3)) * box
    elif case == 'clustered':
        pos1, pos2 = clustered(rng, 6000, box), clustered(rng, 2500, box)
    elif case == 'flat box':
        # fewer than three cells along y and z: one cell there, minimum image only
        box = np.array([200.0, 60.0, 90.0])
        pos1, pos2 = clustered(rng, 4000, box, 25), rng.uniform(0, 1, (2000, 3)) * box
    else:
        pos1 = rng.uniform(0, 1, (3000, 3)) * box
        pos1[:500, 0] = 0.0
        pos1[500:1000, 1] = box[1]          # exactly on the upper face
        pos1[1000:1500, 2] = 0.0
        pos1[1500:1600] = pos1[1400:1500]   # duplicates: separation exactly 0
        pos2 = pos1[::3] + 0.0
    for a, b in ((pos1, None), (pos1, pos2), (pos2, pos1)):
        got = corrfunc.pair_count_rppi(a, rp_bins, pi_max, b, box)
        expect = oracle.pair_count_rppi(a, b, box, rp_bins, pi_max)
        assert got.dtype == np.uint64 and got.shape == (12, 40)
        assert np.array_equal(got, expect), (case, int(np.abs(
            got.astype(np.int64) - expect.astype(np.int64)).sum()))
    # first edge 0: self pairs and exact duplicates count
    edges0 = np.array([0.0, 0.5, 2.0, 10.0])
    assert np.array_equal(corrfunc.pair_count_rppi(pos1, edges0, 5.0, None, box, n_pi=3),
                          oracle.pair_count_rppi(pos1, None, box, edges0, 5.0, 3))


def test_empty_and_tiny_samples_and_errors():
    from tabcorr_amd import corrfunc
    box = 50.0
    rp_bins = np.array([1.0, 2.0, 4.0])
    empty = np.zeros((0, 3))
    one = np.array([[1.0, 2.0, 3.0]])
    assert corrfunc.pair_count_rppi(empty, rp_bins, 5.0, None, box).sum() == 0
    assert corrfunc.pair_count_rppi(one, rp_bins, 5.0, None, box).sum() == 0
    assert corrfunc.pair_count_rppi(one, rp_bins, 5.0, empty, box).sum() == 0
    two = np.array([[1.0, 2.0, 3.0], [2.5, 2.0, 49.5]])     # rp = 1.5, dz = 3.5 (wrapped)
    assert corrfunc.pair_count_rppi(two, rp_bins, 5.0, None, box)[:, 3].tolist() == [2, 0]
    with pytest.raises(ValueError, match='half the box'):
        corrfunc.pair_count_rppi(two, np.array([1.0, 30.0]), 5.0, None, box)
    with pytest.raises(ValueError, match='outside'):
        corrfunc.pair_count_rppi(two + 60.0, rp_bins, 5.0, None, box)
    with pytest.raises(ValueError, match='increasing'):
        corrfunc.pair_count_rppi(two, np.array([2.0, 1.0]), 5.0, None, box)
    with pytest.raises(ValueError):
        corrfunc.wp(two, rp_bins, 5.0, period=box, do_auto=False, do_cross=False)


def test_wp_and_the_tabulation_matrix():
    """corrfunc.wp and compute_tpcf_matrix against the reference's loop structure run with
    the brute-force counter: identical counts, identical arithmetic -> identical floats."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    rng = np.random.default_rng(77)
    box = 150.0
    rp_bins = np.logspace(-1, 1.3, 11)
    pi_max = 40.0
    halos = clustered(rng, 7000, np.full(3, box), 60, 2.0)
    # halo bins of very different sizes, one of them empty (tabcorr.py:888 skips it)
    cuts = [0, 40, 40, 400, 1500, 3500, 7000]
    pos = [halos[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    matrix, shape = corrfunc.compute_tpcf_matrix('auto', pos, box, rp_bins, pi_max)
    expect, expect_shape = oracle.compute_tpcf_matrix_wp('auto', pos, box, rp_bins, pi_max)
    assert shape == expect_shape == (10, )
    assert np.array_equal(matrix, expect)
    assert np.all(matrix[:, 1] == 0)
    particles = rng.uniform(0, box, (4000, 3))
    cross, _ = corrfunc.compute_tpcf_matrix('cross', pos, box, rp_bins, pi_max,
                                            sample2=particles)
    expect, _ = oracle.compute_tpcf_matrix_wp('cross', pos, box, rp_bins, pi_max,
                                              sample2=particles)
    assert np.array_equal(cross, expect)
    # the reference's own call signature (what TabCorr.tabulate swaps in for its pool)
    swapped, _ = corrfunc.reference_compute_tpcf_matrix(
        'auto', pos, corrfunc.wp, np.full(3, box), (rp_bins, pi_max), {}, num_threads=4)
    assert np.array_equal(swapped, matrix)
    swapped, _ = corrfunc.reference_compute_tpcf_matrix(
        'cross', pos, corrfunc.wp, np.full(3, box), (rp_bins, pi_max),
        {'sample2': particles, 'do_auto': False, 'do_cross': True})
    assert np.array_equal(swapped, cross)
    # the single-pair entry point, as TabCorr.tabulate would call it through the pool
    assert np.array_equal(corrfunc.wp(pos[3], rp_bins, pi_max, period=box), matrix[:, 3, 3])
    assert np.array_equal(
        corrfunc.wp(pos[2], rp_bins, pi_max, sample2=pos[4], period=box, do_auto=False,
                    do_cross=True), matrix[:, 2, 4])


def test_labelled_counts_at_tabulation_scale():
    """2 x 10^5 halos in 100 bins (the shape of BASELINE configs[1]'s table): properties that
    do not need the O(N^2) oracle -- symmetry, even diagonal, the sum over bin pairs equals
    the unlabelled count, invariance under a relabelling -- plus the oracle on a sub-box."""
    from tabcorr_amd import corrfunc
    rng = np.random.default_rng(3)
    box = 250.0
    n = 200000
    pos = clustered(rng, n, np.full(3, box), 3000, 3.0)
    label = rng.integers(0, 100, n)
    rp_bins = np.logspace(-1, np.log10(30.0), 20)
    order = np.argsort(label, kind='stable')
    bins = np.split(pos[order], np.cumsum(np.bincount(label, minlength=100))[:-1])
    counts = corrfunc.pair_count_matrix(bins, rp_bins, 40.0, box)
    assert counts.shape == (19, 100, 100)
    assert np.array_equal(counts, counts.transpose(0, 2, 1))
    diagonal = counts[:, np.arange(100), np.arange(100)]
    assert np.all(diagonal % 2 == 0)
    total = corrfunc.pair_count_rppi(pos, rp_bins, 40.0, None, box).sum(axis=1)
    assert np.array_equal(counts.sum(axis=(1, 2)), total)
    assert total.sum() > 1e7
    # merging bins adds their blocks
    merged = [np.concatenate(bins[2 * k:2 * k + 2]) for k in range(50)]
    coarse = corrfunc.pair_count_matrix(merged, rp_bins, 40.0, box)
    assert np.array_equal(coarse, counts.reshape(19, 50, 2, 50, 2).sum(axis=(2, 4)))


@pytest.mark.parametrize('case', ['uniform', 'clustered', 'flat box'])
def test_s_mu_pair_counts_are_exact(case):
    """DD(s, mu) (tabcorr/corrfunc.py:98-175) against the brute-force oracle, and the
    s_mu_tpcf wrapper built on it."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    rng = np.random.default_rng({'uniform': 11, 'clustered': 12, 'flat box': 13}[case])
    box = np.array([100.0, 100.0, 100.0])
    s_bins = np.logspace(-0.5, np.log10(20.0), 9)
    if case == 'uniform':
        pos1, pos2 = rng.uniform(0, 1, (4000, 3)) * box, rng.uniform(0, 1, (2500, 3)) * box
    elif case == 'clustered':
        pos1, pos2 = clustered(rng, 5000, box), clustered(rng, 2000, box)
    else:
        box = np.array([150.0, 50.0, 55.0])
        pos1, pos2 = clustered(rng, 3000, box, 20), rng.uniform(0, 1, (1500, 3)) * box
        pos1[:200, 2] = pos1[200:400, 2]        # pairs exactly across the line of sight: mu = 0
        pos1[400:500, :2] = pos1[500:600, :2]   # ... and exactly along it: mu = 1, never counted
    for a, b in ((pos1, None), (pos1, pos2)):
        got = corrfunc.pair_count_smu(a, s_bins, 10, b, box)
        expect = oracle.pair_count_smu(a, b, box, s_bins, 10)
        assert np.array_equal(got, expect), (case, int(np.abs(
            got.astype(np.int64) - expect.astype(np.int64)).sum()))
    mu_bins = np.linspace(0, 1, 11)
    assert np.array_equal(corrfunc.s_mu_tpcf(pos1, s_bins, mu_bins, period=box),
                          oracle.s_mu_tpcf(pos1, s_bins, mu_bins, period=box))
    assert np.array_equal(
        corrfunc.s_mu_tpcf(pos1, s_bins, mu_bins, sample2=pos2, period=box, do_auto=False,
                           do_cross=True),
        oracle.s_mu_tpcf(pos1, s_bins, mu_bins, sample2=pos2, period=box, do_auto=False,
                         do_cross=True))
    with pytest.raises(ValueError, match='uniform'):
        corrfunc.s_mu_tpcf(pos1, s_bins, np.array([0.0, 0.2, 1.0]), period=box)
    # first edge 0: self pairs land in mu bin 0
    edges0 = np.array([0.0, 1.0, 4.0])
    assert np.array_equal(corrfunc.pair_count_smu(pos1, edges0, 4, None, box),
                          oracle.pair_count_smu(pos1, None, box, edges0, 4))


def test_labelled_counts_with_private_counters():
    """Many points per (cell, label): the labelled count keeps per-workgroup counters in LDS
    (points sorted by label inside the cells, at most 8 labels per workgroup).  Against the
    brute-force oracle, auto and cross, with a label that has no points at all."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    rng = np.random.default_rng(21)
    box = np.array([100.0, 100.0, 100.0])
    rp_bins = np.logspace(-0.5, np.log10(20.0), 8)
    pos = clustered(rng, 12000, box, 30, 2.5)
    label = rng.choice([0, 1, 3], size=len(pos), p=[0.5, 0.3, 0.2])      # label 2 is empty
    order = np.argsort(label, kind='stable')
    bins = np.split(pos[order], np.cumsum(np.bincount(label, minlength=4))[:-1])
    counts = corrfunc.pair_count_matrix(bins, rp_bins, 40.0, box)
    expect = oracle.pair_count_rppi(pos, None, box, rp_bins, 40.0, label1=label, n_labels=4)
    assert np.array_equal(counts, expect)
    assert counts[:, 2].sum() == 0 and counts[:, :, 2].sum() == 0 and counts.sum() > 1e6
    other = rng.uniform(0, 1, (5000, 3)) * box
    cross = corrfunc.pair_count_matrix(bins, rp_bins, 40.0, box, sample2=other)
    expect = oracle.pair_count_rppi(pos, other, box, rp_bins, 40.0, label1=label,
                                    label2=np.zeros(len(other), dtype=int), n_labels=4)
    assert np.array_equal(cross, expect[:, :, 0])


def test_s_mu_tabulation_matrix():
    """compute_tpcf_matrix for tpcf = s_mu_tpcf (tabcorr/tabcorr.py:846-922 with
    tabcorr/corrfunc.py:98-175): all bin pairs in one labelled pass against the reference's
    loop of per-pair calls run with the brute-force counter.  Counts are integers (exact);
    the float arithmetic behind them is the same sequence of operations -> identical."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    rng = np.random.default_rng(91)
    box = 140.0
    s_bins = np.logspace(-0.5, np.log10(30.0), 9)
    mu_bins = np.linspace(0, 1, 7)
    halos = clustered(rng, 6000, np.full(3, box), 50, 2.0)
    cuts = [0, 30, 30, 300, 1300, 3000, 6000]               # one empty bin
    pos = [halos[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    # global 64-bit atomics (few points per cell and label) ...
    counts = corrfunc.pair_count_matrix_smu(pos, s_bins, 6, box)
    assert counts.shape == (8, 6, 6, 6) and counts.dtype == np.uint64
    for a in (0, 2, 3, 5):
        for b in (2, 4, 5):
            expect = oracle.pair_count_smu(pos[a], pos[b] if a != b else None,
                                           np.full(3, box), s_bins, 6)
            assert np.array_equal(counts[:, :, a, b], expect), (a, b)
    assert np.array_equal(counts, counts.transpose(0, 1, 3, 2))
    assert np.array_equal(counts.sum(axis=(2, 3)),
                          corrfunc.pair_count_smu(halos, s_bins, 6, None, box))
    matrix, shape = corrfunc.compute_tpcf_matrix_smu('auto', pos, box, s_bins, mu_bins)
    expect, expect_shape = oracle.compute_tpcf_matrix_smu('auto', pos, box, s_bins, mu_bins)
    assert shape == expect_shape == (8, 6) and matrix.shape == (48, 6, 6)
    assert np.array_equal(matrix, expect)
    assert np.all(matrix[:, 1] == 0)
    particles = rng.uniform(0, box, (3000, 3))
    cross, _ = corrfunc.compute_tpcf_matrix_smu('cross', pos, box, s_bins, mu_bins,
                                                sample2=particles)
    expect, _ = oracle.compute_tpcf_matrix_smu('cross', pos, box, s_bins, mu_bins,
                                               sample2=particles)
    assert cross.shape == (48, 6) and np.array_equal(cross, expect)
    swapped, swapped_shape = corrfunc.reference_compute_tpcf_matrix(
        'auto', pos, corrfunc.s_mu_tpcf, np.full(3, box), (s_bins, mu_bins), {})
    assert swapped_shape == (8, 6) and np.array_equal(swapped, matrix)
    assert np.array_equal(
        corrfunc.s_mu_tpcf(pos[3], s_bins, mu_bins, period=box).ravel(), matrix[:, 3, 3])
    assert np.array_equal(
        corrfunc.s_mu_tpcf(pos[2], s_bins, mu_bins, sample2=pos[4], period=box,
                           do_auto=False, do_cross=True).ravel(), matrix[:, 2, 4])
    with pytest.raises(ValueError):
        corrfunc.compute_tpcf_matrix_smu('auto', pos, box, s_bins, np.array([0, 0.3, 1.0]))
    # ... and per-workgroup private counters (many points per cell and label)
    dense = clustered(rng, 12000, np.full(3, 100.0), 30, 2.5)
    label = rng.choice([0, 2], size=len(dense), p=[0.6, 0.4])
    order = np.argsort(label, kind='stable')
    bins = np.split(dense[order], np.cumsum(np.bincount(label, minlength=3))[:-1])
    few_s = np.array([0.5, 2.0, 8.0, 20.0])
    counts = corrfunc.pair_count_matrix_smu(bins, few_s, 4, 100.0)
    for a in (0, 2):
        for b in (0, 2):
            expect = oracle.pair_count_smu(bins[a], bins[b] if a != b else None,
                                           np.full(3, 100.0), few_s, 4)
            assert np.array_equal(counts[:, :, a, b], expect), (a, b)
    assert counts[:, :, 1].sum() == 0 and counts.sum() > 1e6


def test_wrappers_match_the_reference_fixture():
    """HIP path against tests/golden/paircount_wrappers.npz, recorded from the REFERENCE's
    tabcorr.corrfunc.wp / s_mu_tpcf / tabcorr.tabcorr.compute_tpcf_matrix over a stub
    Corrfunc counter (make_golden.py: golden_paircount): the drop-in wrappers, and the
    one-pass form of the matrix the tabulation swaps in for the reference's per-pair loop
    (unequal bin sizes, an empty bin, auto and cross)."""
    from tabcorr_amd import corrfunc
    from util import load_golden
    data = load_golden('paircount_wrappers')
    offsets = np.concatenate([[0], np.cumsum(data['sizes'])])
    pos = [data['pos'][a:b] for a, b in zip(offsets[:-1], offsets[1:])]
    period, rp_bins, pi_max = data['period'], data['rp_bins'], float(data['pi_max'])
    s_bins, mu_bins = data['s_bins'], data['mu_bins']
    close = dict(rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(corrfunc.wp(pos[0], rp_bins, pi_max, period=period),
                               data['wp_auto'], **close)
    np.testing.assert_allclose(
        corrfunc.wp(pos[0], rp_bins, pi_max, sample2=pos[3], period=period, do_auto=False,
                    do_cross=True), data['wp_cross'], **close)
    np.testing.assert_allclose(
        corrfunc.wp(np.mod(pos[3], 60.0), rp_bins, pi_max, period=60.0),
        data['wp_auto_scalar_period'], **close)
    np.testing.assert_allclose(corrfunc.s_mu_tpcf(pos[0], s_bins, mu_bins, period=period),
                               data['smu_auto'], **close)
    np.testing.assert_allclose(
        corrfunc.s_mu_tpcf(pos[0], s_bins, mu_bins, sample2=pos[3], period=period,
                           do_auto=False, do_cross=True), data['smu_cross'], **close)
    cross_kwargs = dict(sample2=data['particles'], do_auto=False, do_cross=True)
    for name, tpcf, args in (('wp', corrfunc.wp, (rp_bins, pi_max)),
                             ('smu', corrfunc.s_mu_tpcf, (s_bins, mu_bins))):
        matrix, shape = corrfunc.reference_compute_tpcf_matrix(
            'auto', pos, tpcf, period, args, {})
        assert tuple(shape) == tuple(data['shape_auto_' + name])
        np.testing.assert_allclose(matrix, data['matrix_auto_' + name], **close)
        matrix, shape = corrfunc.reference_compute_tpcf_matrix(
            'cross', pos, tpcf, period, args, cross_kwargs)
        assert tuple(shape) == tuple(data['shape_cross_' + name])
        np.testing.assert_allclose(matrix, data['matrix_cross_' + name], **close)


def test_mass_in_cylinders_and_mean_delta_sigma():
    """SURVEY.md 8f.4, the third estimator of the reference's database
    (scripts/tabulate_snapshot.py:228-237): per-object mass in cylinders on the GPU against
    the brute-force oracle -- exact for equal masses (counts), 1e-12 for per-particle masses
    (the order of the additions differs) -- and the excess surface density built on it, per
    object, averaged, and for all halo bins in one pass."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    rng = np.random.default_rng(12)
    for box, n_gal, n_ptcl in ((np.array([150.0, 150.0, 150.0]), 3000, 40000),
                               (np.array([90.0, 200.0, 60.0]), 1500, 25000)):
        galaxies = clustered(rng, n_gal, box, 30, 4.0)
        particles = np.vstack([clustered(rng, n_ptcl // 2, box, 30, 6.0),
                               rng.uniform(0, 1, (n_ptcl - n_ptcl // 2, 3)) * box])
        galaxies[:5, 0] = 0.0
        galaxies[5:10, 1] = box[1]                    # on the faces of the box
        particles[:50] = galaxies[:50]                # separation exactly 0
        rp_bins = np.logspace(-1, np.log10(0.3 * box.min() / 2), 14)
        got = corrfunc.mass_in_cylinders(galaxies, particles, 2.5e9, rp_bins, box)
        expect = oracle.mass_in_cylinders(galaxies, particles, 2.5e9, rp_bins, box)
        assert got.shape == (n_gal, 14)
        assert np.array_equal(got, expect)
        weights = rng.uniform(0.5, 2.0, n_ptcl)
        got = corrfunc.mass_in_cylinders(galaxies, particles, weights, rp_bins, box)
        expect = oracle.mass_in_cylinders(galaxies, particles, weights, rp_bins, box)
        np.testing.assert_allclose(got, expect, rtol=1e-12, atol=0)
        ds = corrfunc.mean_delta_sigma(galaxies, particles, 2.5e9, rp_bins, period=box,
                                       num_threads=4)
        np.testing.assert_allclose(
            ds, oracle.mean_delta_sigma(galaxies, particles, 2.5e9, rp_bins, box), rtol=1e-12)
        per_object = corrfunc.mean_delta_sigma(galaxies, particles, weights, rp_bins,
                                               period=box, per_object=True)
        np.testing.assert_allclose(
            per_object, oracle.mean_delta_sigma(galaxies, particles, weights, rp_bins, box,
                                                per_object=True), rtol=1e-10, atol=1e-12)
    # all halo bins in one pass, as TabCorr.tabulate calls it (tabcorr.py:322-325, 844)
    bins = [galaxies[:400], galaxies[400:400], galaxies[400:1000], galaxies[1000:]]
    matrix, shape = corrfunc.reference_compute_tpcf_matrix(
        'cross', bins, corrfunc.mean_delta_sigma, box, (particles, 2.5e9, rp_bins), {})
    expect, expect_shape = oracle.compute_tpcf_matrix_ds(bins, box, particles, 2.5e9, rp_bins)
    assert tuple(shape) == tuple(expect_shape) == (13, )
    np.testing.assert_allclose(matrix, expect, rtol=1e-12)
    assert np.all(matrix[:, 1] == 0)
    # errors and edge cases
    with pytest.raises(ValueError, match='half the box'):
        corrfunc.mass_in_cylinders(galaxies, particles, 1.0, np.array([1.0, 46.0]), box)
    with pytest.raises(ValueError, match='one entry per particle'):
        corrfunc.mass_in_cylinders(galaxies, particles, np.ones(3), rp_bins, box)
    with pytest.raises(ValueError, match='outside'):
        corrfunc.mass_in_cylinders(galaxies - 500.0, particles, 1.0, rp_bins, box)
    assert corrfunc.mass_in_cylinders(galaxies[:0], particles, 1.0, rp_bins, box).shape == (0, 14)
    assert np.all(corrfunc.mass_in_cylinders(galaxies, particles[:0], 1.0, rp_bins, box) == 0)


def test_tabulate_swaps_in_the_one_pass_pair_count(monkeypatch):
    """TabCorr.tabulate delegates mock population to the reference package (halotools-bound,
    absent here) and swaps the reference's per-pair process pool (tabcorr.py:846-922) for the
    GPU's one-pass count.  A stand-in `tabcorr` package whose tabulate() drives
    module.compute_tpcf_matrix exactly as the reference does (tabcorr.py:318-344: call,
    symmetric packing) executes that glue end to end: matrix = oracle's, the reference's
    function restored afterwards, and the tabulated object predicts."""
    import types
    from tabcorr_amd import TabCorr, corrfunc, synthetic, symmetric_matrix_to_array
    from oracle import paircount_oracle as oracle

    def reference_compute_tpcf_matrix(*args, **kwargs):
        raise AssertionError('the process-pool loop must have been swapped out')

    module = types.ModuleType('tabcorr.tabcorr')
    module.compute_tpcf_matrix = reference_compute_tpcf_matrix

    class ReferenceTabCorr:
        @classmethod
        def tabulate(cls, halocat, tpcf, *tpcf_args, mode='auto', **kwargs):
            matrix, shape = module.compute_tpcf_matrix(
                mode, halocat['pos'], tpcf, halocat['period'], tpcf_args,
                halocat.get('tpcf_kwargs', {}), num_threads=kwargs.get('num_threads', 1))
            if mode == 'auto':
                matrix = np.array([symmetric_matrix_to_array(m) for m in matrix])
            out = cls()
            out.gal_type, out.tpcf_matrix, out.tpcf_shape = halocat['gal_type'], matrix, shape
            out.attrs, out.tpcf_args, out.tpcf_kwargs = halocat['attrs'], tpcf_args, {}
            return out
    ReferenceTabCorr.__module__ = 'tabcorr.tabcorr'
    package = types.ModuleType('tabcorr')
    package.TabCorr = ReferenceTabCorr
    package.tabcorr = module
    monkeypatch.setitem(sys.modules, 'tabcorr', package)
    monkeypatch.setitem(sys.modules, 'tabcorr.tabcorr', module)

    table = synthetic.synthetic_table(4, 1, (5, ), 'auto', seed=3)      # 8 halo/galaxy bins
    rng = np.random.default_rng(21)
    period = np.array([80.0, 80.0, 80.0])
    pos = [clustered(rng, n, period, 20, 3.0) for n in (300, 120, 260, 90, 400, 150, 60, 220)]
    rp_bins = np.array([0.3, 0.8, 2.0, 5.0, 9.0, 15.0])
    halocat = {'pos': pos, 'period': period, 'gal_type': table['gal_type'],
               'attrs': table['attrs']}
    halotab = TabCorr.tabulate(halocat, corrfunc.wp, rp_bins, 20.0, mode='auto')
    assert module.compute_tpcf_matrix is reference_compute_tpcf_matrix      # restored
    expect, shape = oracle.compute_tpcf_matrix_wp('auto', pos, period, rp_bins, 20.0)
    expect = np.array([symmetric_matrix_to_array(m) for m in expect])
    assert tuple(halotab.tpcf_shape) == tuple(shape) == (5, )
    np.testing.assert_allclose(halotab.tpcf_matrix, expect, rtol=1e-13, atol=1e-13)
    ngal, xi = halotab.predict_batch(synthetic.zheng07_draws(8, seed=4))
    assert xi.shape == (8, 5) and np.all(np.isfinite(xi)) and np.all(ngal > 0)
    # the excess surface density, mode 'cross' (scripts/tabulate_snapshot.py:228-237)
    particles = clustered(rng, 20000, period, 20, 5.0)
    halocat['attrs'] = dict(table['attrs'], mode='cross')
    ds_bins = np.logspace(-0.5, 1.2, 7)
    halotab = TabCorr.tabulate(halocat, corrfunc.mean_delta_sigma, particles, 1.5e10, ds_bins,
                               mode='cross')
    expect, shape = oracle.compute_tpcf_matrix_ds(pos, period, particles, 1.5e10, ds_bins)
    np.testing.assert_allclose(halotab.tpcf_matrix, expect, rtol=1e-12)
    ngal, ds = halotab.predict_batch(synthetic.zheng07_draws(8, seed=5))
    assert ds.shape == (8, 6) and np.all(np.isfinite(ds))
    # any other two-point function stays with the reference's own loop
    with pytest.raises(AssertionError, match='swapped out'):
        TabCorr.tabulate(halocat, lambda *a, **k: None, rp_bins, mode='cross')
    assert module.compute_tpcf_matrix is reference_compute_tpcf_matrix


def _pairs_on_a_bin_edge(n_cases, seed=5):
    """Separations (dx, dy) and a bin edge e such that r^2 = dx dx + dy dy formed with
    separately rounded products and sum equals fl(e e) exactly, while a fused multiply-add
    (dx dx + fl(dy dy) rounded once) falls on the other side of it."""
    import math
    rng = np.random.default_rng(seed)
    cases = []
    scale = 2.0**-30
    while len(cases) < n_cases:
        dx = float(rng.integers(1, 2**30)) * scale * 4.0        # < 4, exact in the box below
        dy = float(rng.integers(1, 2**30)) * scale * 4.0
        separate = dx * dx + dy * dy                             # NumPy / the oracle
        fused = math.fma(dx, dx, dy * dy) if hasattr(math, 'fma') else None
        if fused is None:
            # (Python < 3.13: exact rational arithmetic)
            from fractions import Fraction
            exact = Fraction(dx) * Fraction(dx) + Fraction(dy * dy)
            fused = float(exact)
        if fused == separate:
            continue
        upper = max(separate, fused)
        edge = math.sqrt(upper)
        for candidate in (edge, np.nextafter(edge, 0.0), np.nextafter(edge, 10.0)):
            if candidate * candidate == upper:
                cases.append((dx, dy, float(candidate), separate >= upper))
                break
    return cases


def test_products_and_sums_are_rounded_separately():
    """r_p^2 exactly on a bin edge with the oracle's arithmetic, on the other side of it with a
    fused multiply-add: the pair must land in the oracle's bin (the toolchain contracts
    a * b + c * d by default)."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle as oracle
    period = np.array([64.0, 64.0, 64.0])
    for dx, dy, edge, in_upper_bin in _pairs_on_a_bin_edge(24):
        rp_bins = np.array([0.5 * edge, edge, 1.5 * edge])
        if rp_bins[-1] >= 0.5 * period[0]:
            continue
        pos = np.array([[16.0, 24.0, 8.0], [16.0 + dx, 24.0 + dy, 8.5]])
        assert pos[1, 0] - pos[0, 0] == dx and pos[1, 1] - pos[0, 1] == dy
        expect = oracle.pair_count_rppi(pos, None, period, rp_bins, 4.0)
        assert expect.sum() == 2 and (expect[1].sum() == 2) == in_upper_bin
        got = corrfunc.pair_count_rppi(pos, rp_bins, 4.0, None, period)
        assert np.array_equal(got, expect), (dx, dy, edge)
        labelled = corrfunc.pair_count_matrix([pos[:1], pos[1:]], rp_bins, 4.0, period)
        assert labelled[1 if in_upper_bin else 0, 0, 1] == 1
        assert labelled.sum() == 2
