"""The asynchronous host-to-host entry points (tc_*_async, tc_table_wait, page-locked
buffers): parity with the synchronous path, the golden vectors and the oracle; results land
in the buffers of the right ticket whatever the order of the waits.  Needs an MI355X."""

import ctypes
import os
import threading

import numpy as np
import pytest

from util import load_golden, table_from_golden, interpolator_tables_from_golden, assert_rel

pytestmark = pytest.mark.gpu

RTOL = 1e-10


def make_tabcorr(table, **kwargs):
    from tabcorr_amd import TabCorr
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                               table['tpcf_shape'], table['attrs'], **kwargs)


def test_pinned_arrays():
    from tabcorr_amd import pinned_empty, pinned_array, is_pinned, pin
    a = pinned_empty((7, 3))
    assert a.shape == (7, 3) and a.dtype == np.float64 and a.flags.c_contiguous
    assert is_pinned(a) and is_pinned(a[2:5])
    assert not is_pinned(np.zeros(4))
    b = pinned_array(np.arange(12.0).reshape(3, 4))
    assert is_pinned(b) and np.array_equal(b, np.arange(12.0).reshape(3, 4))
    own = np.zeros((1000, 5))
    with pin(own) as same:
        assert same is own and is_pinned(own)
    assert not is_pinned(own)
    view = a[1]
    del a                      # views keep the allocation alive
    view[:] = 1.0
    assert is_pinned(view)


def test_async_matches_golden_and_sync():
    from tabcorr_amd import pinned_array, pinned_empty
    data = load_golden('synthetic_cfg2')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    theta = pinned_array(data['theta'])
    n = len(theta)
    n_r = int(np.prod(table['tpcf_shape']))
    ngal, xi = pinned_empty(n), pinned_empty((n, n_r))
    pending = halotab.predict_batch_async(theta, out=(ngal, xi))
    got_ngal, got_xi = pending.wait()
    assert pending.done()
    assert np.shares_memory(got_xi, xi) and np.shares_memory(got_ngal, ngal)
    assert_rel(got_ngal, data['ngal'], RTOL, 'ngal')
    assert_rel(got_xi, data['xi'], RTOL, 'xi')
    # out= through the synchronous signature
    ngal[:] = 0
    xi[:] = 0
    again = halotab.predict_batch(theta, out=(ngal, xi))
    assert_rel(again[1], data['xi'], RTOL)
    # ... and into ordinary arrays kept by the caller
    kept = np.zeros(n), np.zeros((n, n_r))
    into_kept = halotab.predict_batch(theta, out=kept)
    assert np.shares_memory(into_kept[1], kept[1])
    assert np.array_equal(kept[1], halotab.predict_batch(theta)[1])
    assert np.array_equal(kept[0], halotab.predict_batch(theta)[0])
    with pytest.raises(ValueError, match='elements'):
        halotab.predict_batch(theta, out=(np.zeros(n), np.zeros((n, n_r + 1))))
    with pytest.raises(ValueError, match='float64'):
        halotab.predict_batch(theta, out=(np.zeros(n), np.zeros((n, n_r), dtype=np.float32)))
    # separated by galaxy type
    ngal2, xi3 = pinned_empty((n, 2)), pinned_empty((n, 3, n_r))
    n_sep, x_sep = halotab.predict_batch_async(theta, separate_gal_type=True,
                                               out=(ngal2, xi3)).wait()
    for key in n_sep:
        assert_rel(n_sep[key], data['ngal_sep_' + key], RTOL, key)
    for key in x_sep:
        assert_rel(x_sep[key], data['xi_sep_' + key], RTOL, key)
    # polling instead of waiting (tc_table_query)
    import time
    polled = halotab.predict_batch_async(theta, out=(ngal, xi))
    deadline = time.time() + 10.0
    while not polled.done():
        assert time.time() < deadline
    assert_rel(polled.wait()[1], data['xi'], RTOL)
    # pageable theta, no out: pooled pinned staging, results copied out
    plain = halotab.predict_batch_async(np.array(data['theta'])).wait()
    assert_rel(plain[0], data['ngal'], RTOL)
    assert_rel(plain[1], data['xi'], RTOL)
    # the likelihood
    rng = np.random.default_rng(0)
    vector = data['xi'][0] * 1.1
    a = rng.normal(size=(n_r, n_r))
    precision = a @ a.T / np.mean(vector)**2
    expect = np.einsum('bi,ij,bj->b', data['xi'] - vector, precision, data['xi'] - vector)
    n_chi, chi2 = halotab.chi2_batch_async(theta, vector, precision).wait()
    assert_rel(n_chi, data['ngal'], RTOL)
    assert_rel(chi2, expect, 1e-9)
    sync = halotab.chi2_batch(theta, vector, precision)
    assert_rel(chi2, sync[1], 1e-12)


def test_tickets_deliver_into_their_own_buffers():
    """Many calls in flight with different draws and batch sizes, waited for in reverse and
    in shuffled order: every ticket's buffers hold that ticket's results (compared with the
    oracle), and more than 64 pending tickets still resolve."""
    from tabcorr_amd import pinned_array, pinned_empty, synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(12, 2, (7, ), 'auto', seed=5)
    halotab = make_tabcorr(table)
    sizes = [1, 64, 65, 700, 3, 1000, 129, 256, 31, 2048, 5, 77]
    thetas = [pinned_array(synthetic.zheng07_draws(size, seed=40 + i))
              for i, size in enumerate(sizes)]
    outs = [(pinned_empty(size), pinned_empty((size, 7))) for size in sizes]
    for ngal, xi in outs:
        ngal[:] = np.nan
        xi[:] = np.nan
    pending = [halotab.predict_batch_async(theta, out=out)
               for theta, out in zip(thetas, outs)]
    order = list(reversed(range(len(sizes))))
    for index in order:
        ngal, xi = pending[index].wait()
        expect = oracle.predict_zheng07_batch(table, thetas[index])
        assert_rel(ngal, expect[0], RTOL, 'ngal of ticket %d' % index)
        assert_rel(xi, expect[1], RTOL, 'xi of ticket %d' % index)
    # second round: shuffled waits, then 100 small calls without waiting in between
    pending = [halotab.predict_batch_async(theta, out=out)
               for theta, out in zip(thetas, outs)]
    for index in np.random.default_rng(1).permutation(len(sizes)):
        ngal, xi = pending[index].wait()
        expect = oracle.predict_zheng07_batch(table, thetas[index][:4])
        assert_rel(xi[:4], expect[1], RTOL)
    many_theta = [pinned_array(synthetic.zheng07_draws(3, seed=100 + i)) for i in range(100)]
    many_out = [(pinned_empty(3), pinned_empty((3, 7))) for _ in range(100)]
    many = [halotab.predict_batch_async(t, out=o) for t, o in zip(many_theta, many_out)]
    for index in (0, 99, 50, 1, 98):
        ngal, xi = many[index].wait()
        expect = oracle.predict_zheng07_batch(table, many_theta[index])
        assert_rel(xi, expect[1], RTOL, 'ticket %d of 100' % index)
    for item in many:
        item.wait()


def test_async_rejects_pageable_buffers_and_bad_tickets():
    from tabcorr_amd import _lib, pinned_array, pinned_empty, synthetic
    table = synthetic.synthetic_table(6, 1, (5, ), 'auto', seed=2)
    halotab = make_tabcorr(table)
    device = halotab.to_device()
    lib = device.lib
    theta = synthetic.zheng07_draws(10, seed=3)
    ngal, xi = np.empty(10), np.empty((10, 5))
    ticket = ctypes.c_int64(-1)
    status = lib.tc_predict_zheng07_batch_async(
        device.handle, _lib.as_double_p(theta), 5, 10, 10, 0, _lib.as_double_p(ngal),
        _lib.as_double_p(xi), ctypes.byref(ticket))
    assert status == _lib.TC_ERR_INVALID
    assert b'page-locked' in lib.tc_last_error()
    with pytest.raises(ValueError, match='page-locked'):
        halotab.predict_batch_async(theta, out=(ngal, xi))
    with pytest.raises(ValueError, match='elements'):
        halotab.predict_batch_async(theta, out=(pinned_empty(9), pinned_empty((10, 5))))
    assert lib.tc_table_wait(device.handle, 12345) == _lib.TC_ERR_INVALID
    # an empty batch gives a ticket that completes
    empty = halotab.predict_batch_async(pinned_empty((0, 5)),
                                        out=(pinned_empty(0), pinned_empty((0, 5))))
    assert empty.wait()[1].shape == (0, 5)
    # wrong number of columns
    with pytest.raises(ValueError, match='columns'):
        halotab.predict_batch_async(pinned_array(theta[:, :4]))


def test_async_interpolator_matches_golden():
    from tabcorr_amd import Interpolator, pinned_array, pinned_empty
    data = load_golden('interp_2d_auto')
    tables = interpolator_tables_from_golden(data)
    keys = [str(k) for k in data['keys']]
    interp = Interpolator([make_tabcorr(t) for t in tables],
                          {k: data['points'][:, d] for d, k in enumerate(keys)})
    theta, x = pinned_array(data['theta']), pinned_array(data['x'])
    n = len(theta)
    n_r = data['xi'].shape[1]
    outs = [(pinned_empty(n), pinned_empty((n, n_r))) for _ in range(3)]
    pending = [interp.predict_batch_async(theta, x, out=out) for out in outs]
    import time
    deadline = time.time() + 10.0
    while not pending[-1].done():          # tc_interp_query
        assert time.time() < deadline
    for item in reversed(pending):
        ngal, xi = item.wait()
        assert_rel(ngal, data['ngal'], RTOL, 'ngal')
        assert_rel(xi, data['xi'], RTOL, 'xi', floor=1e-12)
    sync = interp.predict_batch(theta, x)
    assert_rel(outs[0][1], sync[1], 1e-12)
    vector = data['xi'][0] * 0.9
    precision = np.eye(n_r) / np.mean(vector)**2
    expect = np.einsum('bi,ij,bj->b', data['xi'] - vector, precision, data['xi'] - vector)
    n_chi, chi2 = interp.chi2_batch_async(theta, x, vector, precision).wait()
    assert_rel(chi2, expect, 1e-8)
    assert_rel(n_chi, data['ngal'], RTOL)
    with pytest.raises(ValueError):
        interp.predict_batch_async(theta, x + 100.0)


def test_threads_share_one_table():
    """predict() from several threads on the same instance (a threaded sampler
    pool): the handle and its scratch arrays sit behind a lock."""
    from tabcorr_amd import Zheng07Model, synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(10, 1, (6, ), 'auto', seed=8)
    halotab = make_tabcorr(table)
    draws = synthetic.zheng07_draws(64, seed=9)
    expect = oracle.predict_zheng07_batch(table, draws)
    errors = []

    def work(offset):
        model = Zheng07Model()
        for index in range(offset, 64, 4):
            for key, value in zip(synthetic.ZHENG07_KEYS, draws[index]):
                model.param_dict[key] = value
            ngal, xi = halotab.predict(model)
            if not (np.allclose(xi, expect[1][index], rtol=1e-10, atol=0) and
                    np.isclose(ngal, expect[0][index], rtol=1e-10)):
                errors.append(index)

    threads = [threading.Thread(target=work, args=(k, )) for k in range(4)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert errors == []


def test_threads_share_a_table_with_its_interpolator():
    """an interpolator works through the handles of its tables (the fused-likelihood
    state and caches of the first one): its lock includes theirs, so chi2 calls on the
    interpolator and predictions on its first table from other threads do not mix."""
    from tabcorr_amd import Interpolator, synthetic
    from oracle import tabcorr_oracle as oracle
    tables, keys, points = synthetic.synthetic_interpolator((4, ), 12, 1, (7, ), 'auto', seed=3)
    tabs = [make_tabcorr(t) for t in tables]
    interp = Interpolator(tabs, {k: points[:, d] for d, k in enumerate(keys)})
    theta = synthetic.zheng07_draws(300, seed=4)
    x = np.random.default_rng(5).uniform(points.min(), points.max(), (300, 1))
    setup = oracle.interpolator_setup(tables, points)
    expect_interp = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[:40], x[:40])
    expect_table = oracle.predict_zheng07_batch(tables[0], theta[:40])
    vector = np.full(7, float(np.mean(expect_interp[1])))
    precision = np.eye(7) / np.mean(vector)**2
    want = np.einsum('bi,ij,bj->b', expect_interp[1] - vector, precision,
                     expect_interp[1] - vector)
    errors = []

    def chi2_calls():
        for _ in range(30):
            ngal, chi2 = interp.chi2_batch(theta, x, vector, precision)
            if not np.allclose(chi2[:40], want, rtol=1e-8):
                errors.append('interpolator chi2')

    def table_calls():
        for _ in range(30):
            ngal, xi = tabs[0].predict_batch(theta)
            if not np.allclose(xi[:40], expect_table[1], rtol=1e-10):
                errors.append('table xi')

    threads = [threading.Thread(target=f) for f in (chi2_calls, table_calls, chi2_calls)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join()
    assert errors == []


def test_overlapping_registrations_are_rejected():
    """a second tc_host_register of memory the registry already knows would replace
    the first entry; it is refused instead."""
    from tabcorr_amd import pin, is_pinned
    own = np.zeros(4096)
    with pin(own):
        for view in (own, own[100:200], own[-8:]):
            with pytest.raises(ValueError):
                with pin(view):
                    pass
        assert is_pinned(own)
    assert not is_pinned(own)
    with pin(own[:2048]), pin(own[2048:]):          # adjacent ranges are fine
        assert is_pinned(own[:2048]) and is_pinned(own[2048:])


def test_many_walkers_in_one_launch():
    """tc_predict_zheng07_many: n independent draws through ONE launch of the un-batched
    kernel, completion polled in host memory -- against the golden vectors, the oracle and
    the three-kernel path, for every n up to the limit and beyond it (forwarded)."""
    from tabcorr_amd import _lib, synthetic
    from oracle import tabcorr_oracle as oracle
    data = load_golden('bolplanck_wp')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    device = halotab.to_device()
    lib = device.lib
    theta = np.ascontiguousarray(data['theta'])
    for n in (1, 2, 3, 7, 16, len(theta)):
        ngal, xi = np.full(n, np.nan), np.full((n, device.n_r), np.nan)
        _lib.check(lib.tc_predict_zheng07_many(
            device.handle, _lib.as_double_p(theta), 5, n, 10, 0, _lib.as_double_p(ngal),
            _lib.as_double_p(xi)))
        assert_rel(ngal, data['ngal'][:n], RTOL, 'ngal, %d walkers' % n)
        assert_rel(xi, data['xi'][:n], RTOL, 'xi, %d walkers' % n)
    # modulate_with_cenocc and n_gauss_prim through the same path; repeated calls (epochs)
    for _ in range(300):
        ngal, xi = halotab.predict_batch(theta[:5], modulate_with_cenocc=True)
    assert_rel(xi, data['xi_modulate'][:5], RTOL)
    ngal, xi = halotab.predict_batch(theta[:9], n_gauss_prim=100)
    assert_rel(xi, data['xi_ng100'][:9], RTOL)
    # the same draws through the three-kernel path and through the oracle
    draws = synthetic.zheng07_draws(64, seed=12)
    _lib.check(lib.tc_table_set_option(device.handle, b'single_draw', 0))
    batch = halotab.predict_batch(draws)
    _lib.check(lib.tc_table_set_option(device.handle, b'single_draw', 1))
    many = halotab.predict_batch(draws)
    assert_rel(many[1], batch[1], 1e-12)
    expect = oracle.predict_zheng07_batch(table, draws)
    assert_rel(many[0], expect[0], RTOL)
    assert_rel(many[1], expect[1], RTOL)
    # stream synchronisation instead of polling gives the same
    _lib.check(lib.tc_table_set_option(device.handle, b'poll_done', 0))
    assert_rel(halotab.predict_batch(draws)[1], many[1], 0.0, floor=0.0)
    _lib.check(lib.tc_table_set_option(device.handle, b'poll_done', 1))
    # more walkers than one launch takes, and a request the path cannot serve: forwarded
    big = synthetic.zheng07_draws(100, seed=13)
    ngal, xi = np.empty(100), np.empty((100, device.n_r))
    _lib.check(lib.tc_predict_zheng07_many(
        device.handle, _lib.as_double_p(big), 5, 100, 10, 0, _lib.as_double_p(ngal),
        _lib.as_double_p(xi)))
    assert_rel(xi, oracle.predict_zheng07_batch(table, big)[1], RTOL)
    ngal2, xi3 = np.empty((4, 2)), np.empty((4, 3, device.n_r))
    _lib.check(lib.tc_predict_zheng07_many(
        device.handle, _lib.as_double_p(theta), 5, 4, 10, _lib.FLAG_SEPARATE_GAL_TYPE,
        _lib.as_double_p(ngal2), _lib.as_double_p(xi3)))
    assert_rel(xi3[:, 1], data['xi_sep_centrals-satellites'][:4], RTOL)
    # NaN parameters reject the draw only
    bad = theta[:6].copy()
    bad[2, 1] = np.nan
    ngal, xi = halotab.predict_batch(bad)
    assert np.isnan(ngal[2]) and np.all(np.isnan(xi[2]))
    assert_rel(xi[[0, 1, 3, 4, 5]], data['xi'][[0, 1, 3, 4, 5]], RTOL)


def test_random_interleaving_of_all_host_paths():
    """tools/r03_stress.py for a few seconds: synchronous, un-batched, many-walker,
    asynchronous (random waits) and device-pointer calls of random sizes on ONE table, every
    result against a reference -- workspaces, tickets and completion epochs under reuse."""
    import os
    import subprocess
    import sys
    from util import REPO
    result = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'r03_stress.py'), '4'],
                            capture_output=True, text=True, timeout=300)
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-2000:]
    assert 'stress ok' in result.stdout


def test_resident_kernel_serves_unbatched_calls():
    """Option "resident": ONE launch of resident_draw_kernel answers the un-batched calls from a
    mailbox in page-locked memory -- the same results as one launch per call to the last bits, against the oracle, across
    idle time-outs (the kernel leaves, the next call launches it again), other kinds of calls
    in between (they stop it), other flags, and a handle destroyed while it runs."""
    import time
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(300, seed=8)
    halotab = make_tabcorr(table)
    halotab.set_resident(False)         # (one launch per call, not the automatic mode)
    plain = [halotab.predict_batch(theta[i:i + 1]) for i in range(300)]
    halotab.set_resident(True)
    first = []
    for i in range(300):
        ngal, xi = halotab.predict_batch(theta[i:i + 1])
        # (another instantiation of the same body: the compiler contracts a few products and
        # sums differently -- the last bits may differ, nothing else)
        assert_rel(ngal, plain[i][0], 1e-14)
        assert_rel(xi, plain[i][1], 1e-14)
        first.append((ngal, xi))
    plain = first           # from here on bit for bit: the same kernel, whatever happens around
    want = oracle.predict_zheng07_batch(table, theta[:5])
    for i in range(5):
        assert_rel(plain[i][0], want[0][i:i + 1], RTOL)
        assert_rel(plain[i][1], want[1][i:i + 1], RTOL)
    # idle time-out: the kernel leaves after 50 us without a call
    halotab.set_resident(True, idle_us=50)
    for i in range(40):
        ngal, xi = halotab.predict_batch(theta[i:i + 1])
        assert np.array_equal(xi, plain[i][1]), i
        time.sleep(0.0002 * (i % 4))
    halotab.set_resident(True, idle_us=2000)
    # a batch, several walkers, the likelihood, other flags in between
    batch = halotab.predict_batch(theta[:100])
    assert_rel(batch[1][:5], want[1], RTOL)
    ngal, xi = halotab.predict_batch(theta[7:8])
    assert np.array_equal(xi, plain[7][1])
    several = halotab.predict_batch(theta[:3])
    assert_rel(several[1], want[1][:3], RTOL)
    ngal, xi = halotab.predict_batch(theta[9:10], modulate_with_cenocc=True)
    expect = oracle.predict_zheng07_batch(table, theta[9:10], modulate_with_cenocc=True)
    assert_rel(xi, expect[1], RTOL)
    ngal, xi = halotab.predict_batch(theta[9:10])
    assert np.array_equal(xi, plain[9][1])
    # separated by galaxy type is not a resident call: served by the batched path
    ngal_s, xi_s = halotab.predict_batch(theta[9:10], separate_gal_type=True)
    total = sum(xi_s[key] for key in xi_s)
    assert_rel(total, plain[9][1], 1e-12)
    # NaN parameters
    bad = theta[11:12].copy()
    bad[0, 1] = np.nan
    ngal, xi = halotab.predict_batch(bad)
    assert np.isnan(ngal[0]) and np.all(np.isnan(xi))
    ngal, xi = halotab.predict_batch(theta[11:12])
    assert np.array_equal(xi, plain[11][1])
    # switched off: stops it; a table deleted while its kernel runs stops it too
    halotab.set_resident(False)
    ngal, xi = halotab.predict_batch(theta[12:13])
    assert_rel(xi, plain[12][1], 1e-14)
    other = make_tabcorr(table)
    other.set_resident(True)
    other.predict_batch(theta[:1])
    del other
    # the model interface (the reference's usage)
    from tabcorr_amd import Zheng07Model
    model = Zheng07Model(redshift=table['attrs']['redshift'])
    for key, value in zip(('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha'), theta[3]):
        model.param_dict[key] = value
    halotab.set_resident(True)
    ngal, xi = halotab.predict(model)
    assert np.array_equal(xi, plain[3][1][0])


def test_resident_kernel_by_itself_for_loops_of_unbatched_calls():
    """The default (option "resident" = 2): a plain loop of predict(model) is moved to the
    resident kernel by the library -- same bits as with the option set, faster than one launch
    per call --, a caller that synchronises the device between its calls never waits for the
    kernel longer than its short idle time and is served by launches again after a while, and
    random pauses, other kinds of calls and device-wide synchronisations in between change no
    result (tools/r04_resident_stress.py with nothing switched on)."""
    import time
    from tabcorr_amd import synthetic, Zheng07Model, _lib
    lib = _lib.load()
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(400, seed=8)
    halotab = make_tabcorr(table)
    halotab.set_resident(False)
    launched = [halotab.predict_batch(theta[i:i + 1]) for i in range(400)]
    halotab.set_resident(True)
    resident = [halotab.predict_batch(theta[i:i + 1]) for i in range(400)]
    # round 6: the two kernels are instances of one body compiled with
    # -ffp-contract=on (csrc/inst_single.hip), so which of them serves a call -- a matter of
    # timing in the automatic mode -- does not show in the bits
    for i in range(400):
        assert np.array_equal(launched[i][0], resident[i][0]), i
        assert np.array_equal(launched[i][1], resident[i][1]), i
    halotab.set_resident('auto')
    model = Zheng07Model(redshift=table['attrs']['redshift'])
    keys = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')

    def loop(n):
        t0 = time.perf_counter()
        by_launch = 0
        for call in range(n):
            i = call % 400
            for key, value in zip(keys, theta[i]):
                model.param_dict[key] = value
            ngal, xi = halotab.predict(model)
            # the first calls are launches, the later ones the resident kernel's -- but for the
            # odd call after a hiccup of the host longer than the kernel's idle time
            assert (np.array_equal(xi, launched[i][1][0]) or
                    np.array_equal(xi, resident[i][1][0])), call
            if call >= 16 and not np.array_equal(xi, resident[i][1][0]):
                by_launch += 1
        assert by_launch <= n // 100, by_launch
        return (time.perf_counter() - t0) / n
    per_call_auto = loop(3000)
    halotab.set_resident(False)
    t0 = time.perf_counter()
    for call in range(3000):
        halotab.predict_batch(theta[call % 400:call % 400 + 1])
    per_call_launched = (time.perf_counter() - t0) / 3000
    assert per_call_auto < 0.9 * per_call_launched, (per_call_auto, per_call_launched)
    # a caller that synchronises the device after every call: never more than the automatic
    # idle time (250 us) + the synchronisation's own cost, and launches again after the window
    halotab.set_resident('auto')
    waits = []
    for call in range(400):
        ngal, xi = halotab.predict_batch(theta[call % 400:call % 400 + 1])
        t0 = time.perf_counter()
        _lib.check(lib.tc_device_synchronize())
        waits.append(time.perf_counter() - t0)
        assert (np.array_equal(xi, launched[call % 400][1]) or
                np.array_equal(xi, resident[call % 400][1])), call
    assert max(waits) < 600e-6, max(waits)
    assert np.median(waits[-200:]) < 100e-6, np.median(waits[-200:])      # (backed off)
    # random interleaving, nothing switched on
    halotab.set_resident('auto')
    rng = np.random.default_rng(5)
    batch = halotab.predict_batch(theta[:200])
    for call in range(6000):
        what = rng.random()
        i = int(rng.integers(0, 400))
        if what < 0.93:
            ngal, xi = halotab.predict_batch(theta[i:i + 1])
            assert (np.array_equal(xi, launched[i][1]) or
                    np.array_equal(xi, resident[i][1])), (call, i)
        elif what < 0.95:
            again = halotab.predict_batch(theta[:200])
            assert np.array_equal(again[1], batch[1])
        elif what < 0.97:
            _lib.check(lib.tc_device_synchronize())
        elif what < 0.98:
            halotab.predict_batch(theta[i:i + 1], modulate_with_cenocc=True)
        else:
            time.sleep(float(rng.uniform(0, 0.0006)))
    del halotab          # (a table deleted while the kernel it started by itself runs)


def _resident_stats(halotab):
    from tabcorr_amd import _lib
    lib = _lib.load()
    values = [ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()]
    running = ctypes.c_int()
    _lib.check(lib.tc_table_resident_stats(halotab.to_device().handle,
                                           *[ctypes.byref(v) for v in values],
                                           ctypes.byref(running)))
    return tuple(v.value for v in values) + (running.value, )


def test_two_tables_per_step_with_default_options():
    """the reference's documented per-step usage is TWO tables per
    likelihood evaluation -- halotab_wp.predict(model), then halotab_ds.predict(model)
    (docs/guides/overview.rst:86-92, tests/test_database.py:17-18).  The reference's own
    example tables (bolplanck_wp: mode auto, bolplanck_ds: mode cross), predict(model)
    alternately 3000 times with nothing switched on: every result bit for bit the launched
    path's; the library moves BOTH handles to their resident kernels, which stay on the chip
    side by side (no relaunch storms, no fall-backs); a caller that synchronises the device
    never waits longer than the kernels' idle time; and TabCorr.predict_joint -- both calls
    posted before the first answer is waited for -- returns the same bits, faster."""
    import time
    from tabcorr_amd import TabCorr, Zheng07Model, _lib
    from util import REPO
    lib = _lib.load()
    golden = os.path.join(REPO, 'tests', 'golden')
    wp = TabCorr.read(os.path.join(golden, 'bolplanck_wp.hdf5'))
    ds = TabCorr.read(os.path.join(golden, 'bolplanck_ds.hdf5'))
    model = Zheng07Model(redshift=wp.attrs['redshift'])
    rng = np.random.default_rng(0)
    thetas = np.column_stack([rng.uniform(11.8, 12.6, 300), rng.uniform(0.2, 0.6, 300),
                              rng.uniform(11.0, 12.0, 300), rng.uniform(13.0, 13.8, 300),
                              rng.uniform(0.9, 1.2, 300)])
    keys = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')

    def set_theta(i):
        for key, value in zip(keys, thetas[i % 300]):
            model.param_dict[key] = value
    for tab in (wp, ds):
        tab.set_resident(False)
    launched = []
    for i in range(300):
        set_theta(i)
        launched.append((wp.predict(model), ds.predict(model)))
    # against the golden vectors of the reference itself, through the same un-batched path
    for tab, name in ((wp, 'bolplanck_wp'), (ds, 'bolplanck_ds')):
        data = load_golden(name)
        for i in range(3):
            for key, value in zip(keys, data['theta'][i]):
                model.param_dict[key] = value
            ngal, xi = tab.predict(model)
            assert_rel(ngal, data['ngal'][i], RTOL)
            assert_rel(xi.ravel(), data['xi'][i].ravel(), RTOL)
    for tab in (wp, ds):
        tab.set_resident('auto')            # (= the state of a new handle)

    def loop(n, joint):
        times = []
        for i in range(n):
            set_theta(i)
            t0 = time.perf_counter()
            if joint:
                a, b = TabCorr.predict_joint([wp, ds], model)
            else:
                a, b = wp.predict(model), ds.predict(model)
            times.append(time.perf_counter() - t0)
            assert a[0] == launched[i % 300][0][0] and b[0] == launched[i % 300][1][0], i
            assert np.array_equal(a[1], launched[i % 300][0][1]), i
            assert np.array_equal(b[1], launched[i % 300][1][1]), i
        return np.array(times) * 1e6
    alternately = loop(3000, False)
    stats = [_resident_stats(tab) for tab in (wp, ds)]
    for launches, relaunches, failures, running in stats:
        # both handles were moved to their resident kernels, which are still there; the odd
        # relaunch (a hiccup of the host longer than the idle time) aside, nothing fell back
        assert running == 1 and launches >= 1 and failures == 0, stats
        assert relaunches <= 30 and launches <= 40, stats
    assert np.median(alternately[500:]) < 30.0, np.median(alternately[500:])
    # a caller that synchronises the device after every pair
    waits = []
    for i in range(600):
        set_theta(i)
        a, b = wp.predict(model), ds.predict(model)
        t0 = time.perf_counter()
        _lib.check(lib.tc_device_synchronize())
        waits.append(time.perf_counter() - t0)
        assert np.array_equal(a[1], launched[i % 300][0][1]), i
        assert np.array_equal(b[1], launched[i % 300][1][1]), i
    assert max(waits) < 600e-6, max(waits)
    for tab in (wp, ds):
        tab.set_resident('auto')
    joint = loop(3000, True)
    assert [_resident_stats(tab)[2] for tab in (wp, ds)] == [0, 0]
    # (posting both before waiting: clearly below two calls in a row -- tools/r06_two_tables.py
    # has the numbers; the bound here only guards against a regression to serial waits)
    assert np.median(joint[500:]) < 0.85 * np.median(alternately[500:]), (
        np.median(joint[500:]), np.median(alternately[500:]))
    # launched path of the joint call: same bits as well
    for tab in (wp, ds):
        tab.set_resident(False)
    loop(300, True)


def test_automatic_resident_mode_keeps_its_failures_to_itself():
    """with nothing switched on, an error of the resident kernel (it keeps leaving
    before it answers, no memory for its mailbox, ...) must not reach a caller who never asked
    for that kernel -- the call is served by a launch, the mode backs off, and after three
    failures it stays off.  Option "resident_inject_failures" makes resident_predict fail."""
    from tabcorr_amd import synthetic, _lib
    lib = _lib.load()
    table = synthetic.synthetic_table(30, 1, (19, ), 'auto', seed=3)
    theta = synthetic.zheng07_draws(64, seed=4)
    halotab = make_tabcorr(table)
    halotab.set_resident(False)
    launched = [halotab.predict_batch(theta[i:i + 1]) for i in range(64)]
    halotab.set_resident('auto')
    handle = halotab.to_device().handle
    for failures in (1, 1, 1):
        _lib.check(lib.tc_table_set_option(handle, b'resident_inject_failures', failures))
        # (the mode engages at the eighth call of a tight loop; backed off, it needs 4096 more:
        # set again, it starts afresh)
        halotab.set_resident('auto')
        _lib.check(lib.tc_table_set_option(handle, b'resident_inject_failures', failures))
        for call in range(200):
            ngal, xi = halotab.predict_batch(theta[call % 64:call % 64 + 1])
            assert np.array_equal(ngal, launched[call % 64][0]), call
            assert np.array_equal(xi, launched[call % 64][1]), call
    # explicitly asked for, the error is the caller's to see
    halotab.set_resident(True)
    _lib.check(lib.tc_table_set_option(handle, b'resident_inject_failures', 1))
    with pytest.raises(RuntimeError):
        halotab.predict_batch(theta[:1])
    ngal, xi = halotab.predict_batch(theta[:1])
    assert np.array_equal(xi, launched[0][1])


@pytest.mark.parametrize('shape, n_prim', [((5, 5), 50), ((4, 4, 4), 50), ((4, 7), 50), ((4, 4, 4), 6)])
def test_unbatched_interpolator_in_one_round_of_workgroups(shape, n_prim):
    """Un-batched Interpolator.predict(model): the launch sized so that all tables' workgroups
    are on the chip at once (250 workgroups of two passes for 5 x 5 tables of 100 bins, 256 of
    four passes for the 64 tables of a 4 x 4 x 4 grid) gives what one pass per workgroup gives,
    and what the oracle gives."""
    from tabcorr_amd import Interpolator, Zheng07Model, synthetic, _lib
    from oracle import tabcorr_oracle as oracle
    lib = _lib.load()
    tables, keys, points = synthetic.synthetic_interpolator(shape, n_prim, 1, (19, ), 'auto',
                                                            seed=len(shape) + n_prim)
    interp = Interpolator([make_tabcorr(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    rng = np.random.default_rng(5)
    theta = synthetic.zheng07_draws(3, seed=9)
    setup = oracle.interpolator_setup(tables, points)
    handle = interp.to_device().tables[0].handle
    for i in range(3):
        model = Zheng07Model(redshift=tables[0]['attrs']['redshift'])
        for key, value in zip(('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha'), theta[i]):
            model.param_dict[key] = value
        x = np.array([rng.uniform(points[:, d].min(), points[:, d].max())
                      for d in range(len(keys))])
        for d, key in enumerate(keys):
            model.param_dict[key] = x[d]
        results = []
        for single_round in (1, 0):
            _lib.check(lib.tc_table_set_option(handle, b'single_round', single_round))
            results.append(interp.predict(model))
        _lib.check(lib.tc_table_set_option(handle, b'single_round', 1))
        assert_rel(results[0][0], results[1][0], 1e-13)
        assert_rel(results[0][1], results[1][1], 1e-12)
        want = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[i:i + 1], x[None, :])
        assert_rel(results[0][0], want[0][0], RTOL)
        assert_rel(results[0][1], want[1][0], RTOL)


@pytest.mark.parametrize('aperture', [1, 0])
def test_resident_kernel_serves_ensembles(aperture):
    """Option "resident", 2 .. 256 walkers per synchronous call: resident_ensemble_kernel answers
    without a launch (mailbox in device memory behind the PCIe aperture, or in page-locked
    memory) -- the batched path's results to rounding, the oracle's within the tolerance, a
    walker's result bit for bit the same in an ensemble of any size and from call to call;
    across idle time-outs, other kinds of calls in between, other flags, NaN parameters, more
    walkers than it takes, the reference's table, and a handle destroyed while it runs."""
    import time
    from tabcorr_amd import synthetic, _lib
    from oracle import tabcorr_oracle as oracle
    lib = _lib.load()
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(300, seed=21)
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    _lib.check(lib.tc_table_set_option(handle, b'resident_aperture', aperture))
    # (by default ensembles of fewer than 24 walkers take one launch: here every size the kernel)
    _lib.check(lib.tc_table_set_option(handle, b'resident_min_walkers', 2))
    plain = halotab.predict_batch(theta[:256])
    want = oracle.predict_zheng07_batch(table, theta[:6])
    halotab.set_resident(True)
    full = halotab.predict_batch(theta[:256])
    assert_rel(full[0], plain[0], 1e-12)
    assert_rel(full[1], plain[1], 1e-12)
    assert_rel(full[1][:6], want[1], RTOL)
    assert_rel(full[0][:6], want[0], RTOL)
    for n in (2, 3, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256):
        ngal, xi = halotab.predict_batch(theta[:n])
        assert np.array_equal(xi, full[1][:n]) and np.array_equal(ngal, full[0][:n]), n
    # a walker's place in the ensemble does not matter either
    order = np.random.default_rng(3).permutation(200)
    ngal, xi = halotab.predict_batch(theta[order])
    assert np.array_equal(xi, full[1][order])
    # idle time-out: the kernel leaves after 50 us without a call, the next call launches it
    halotab.set_resident(True, idle_us=50)
    for i in range(30):
        n = 2 + 8 * i
        ngal, xi = halotab.predict_batch(theta[:n])
        assert np.array_equal(xi, full[1][:n]), n
        time.sleep(0.0002 * (i % 4))
    halotab.set_resident(True, idle_us=2000)
    # one walker (the other resident kernel), more than 256 (the batched path), the
    # asynchronous path, other flags in between
    single = halotab.predict_batch(theta[5:6])
    assert_rel(single[1], full[1][5:6], 1e-13)
    ngal, xi = halotab.predict_batch(theta[:70])
    assert np.array_equal(xi, full[1][:70])
    many = halotab.predict_batch(theta[:300])
    assert_rel(many[1][:256], plain[1], 1e-12)
    ngal, xi = halotab.predict_batch(theta[:130])
    assert np.array_equal(xi, full[1][:130])
    ngal, xi = halotab.predict_batch(theta[:40], modulate_with_cenocc=True)
    expect = oracle.predict_zheng07_batch(table, theta[:40], modulate_with_cenocc=True)
    assert_rel(xi, expect[1], RTOL)
    ngal_s, xi_s = halotab.predict_batch(theta[:40], separate_gal_type=True)
    assert_rel(sum(xi_s[key] for key in xi_s), full[1][:40], 1e-12)
    ngal, xi = halotab.predict_batch(theta[:40])
    assert np.array_equal(xi, full[1][:40])
    # NaN parameters reject their walker only
    bad = theta[:20].copy()
    bad[7, 1] = np.nan
    ngal, xi = halotab.predict_batch(bad)
    assert np.isnan(ngal[7]) and np.all(np.isnan(xi[7]))
    keep = [i for i in range(20) if i != 7]
    assert np.array_equal(xi[keep], full[1][keep])
    # decorated occupations
    strengths = np.random.default_rng(11).uniform(-1.2, 1.2, (90, 2))
    table_ab = synthetic.synthetic_table(30, 2, (12, ), 'auto', seed=2)
    decorated = make_tabcorr(table_ab)
    _lib.check(lib.tc_table_set_option(decorated.to_device().handle, b'resident_aperture',
                                       aperture))
    _lib.check(lib.tc_table_set_option(decorated.to_device().handle, b'resident_min_walkers', 2))
    decorated.set_resident(True)
    ngal, xi = decorated.predict_batch(np.hstack([theta[:90], strengths]), assembias=True,
                                       modulate_with_cenocc=True)
    expect = oracle.predict_zheng07_batch(table_ab, theta[:90], modulate_with_cenocc=True,
                                          assembias=strengths)
    assert_rel(xi, expect[1], RTOL)
    assert_rel(ngal, expect[0], RTOL)
    # the reference's own table (G = 60), and a handle deleted while its kernel runs
    golden = make_tabcorr(table_from_golden(load_golden('bolplanck_wp')))
    reference = golden.predict_batch(theta[:100])
    golden.set_resident(True)
    ngal, xi = golden.predict_batch(theta[:100])
    assert_rel(xi, reference[1], 1e-12)
    del golden, decorated
    # the default: fewer than 24 walkers go through one launch of the un-batched kernel
    _lib.check(lib.tc_table_set_option(handle, b'resident_min_walkers', 24))
    ngal, xi = halotab.predict_batch(theta[:10])
    assert_rel(xi, full[1][:10], 1e-12)
    ngal, xi = halotab.predict_batch(theta[:24])
    assert np.array_equal(xi, full[1][:24])
    halotab.set_resident(False)
    ngal, xi = halotab.predict_batch(theta[:50])
    assert_rel(xi, plain[1][:50], 1e-15)


def test_resident_ensemble_kernel_gives_up_cleanly():
    """Workgroups of the resident ensemble kernel that wait in vain (here: a wait limit of 1 us)
    leave and say so; the launched path serves the call, after three such calls all of them,
    and setting the option again brings the kernel back."""
    from tabcorr_amd import synthetic, _lib
    lib = _lib.load()
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(200, seed=5)
    halotab = make_tabcorr(table)
    handle = halotab.to_device().handle
    plain = halotab.predict_batch(theta)
    halotab.set_resident(True)
    served = halotab.predict_batch(theta)
    assert_rel(served[1], plain[1], 1e-12)
    _lib.check(lib.tc_table_set_option(handle, b'resident_wait_us', 1))
    for n in (200, 64, 130, 30, 200):
        ngal, xi = halotab.predict_batch(theta[:n])
        # (served by the launched path, or -- when every wait happened to be over in time --
        # by the kernel)
        assert_rel(xi, plain[1][:n], 1e-12)
        assert_rel(ngal, plain[0][:n], 1e-12)
    _lib.check(lib.tc_table_set_option(handle, b'resident_wait_us', 20000))
    halotab.set_resident(True)
    ngal, xi = halotab.predict_batch(theta)
    assert np.array_equal(xi, served[1])
    halotab.set_resident(False)


@pytest.mark.parametrize('case', ['configs[1] table', 'separate', 'wide table: three kernels',
                                  'float32 rp-pi table', 'mode cross'])
def test_synchronous_calls_in_chunks(case):
    """a synchronous host-array call is cut into chunks of draws whose
    staging, kernels, transfers and copies overlap (table.cpp: predict_chunked).  Against the
    oracle on draws of every chunk and around the chunk boundaries; the same bits for any
    number of chunks where the one-launch form serves the table (its workgroups of 32 draws do
    not care where a draw sits), rounding elsewhere; ragged sizes; the serial path agrees."""
    from tabcorr_amd import synthetic, _lib
    from oracle import tabcorr_oracle as oracle
    lib = _lib.load()
    kwargs, make_kwargs, exact, rtol = {}, {}, True, RTOL
    if case == 'configs[1] table':
        table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    elif case == 'separate':
        table = synthetic.synthetic_table(30, 1, (19, ), 'auto', seed=2)
        kwargs = {'separate_gal_type': True}
    elif case == 'wide table: three kernels':
        table = synthetic.synthetic_table(120, 1, (7, ), 'auto', seed=3)
        exact = False
    elif case == 'float32 rp-pi table':
        table = synthetic.synthetic_table(20, 1, (6, 10), 'auto', seed=4)
        make_kwargs = {'compute_dtype': 'float32'}
        exact, rtol = False, 1e-5
    else:
        table = synthetic.synthetic_table(40, 2, (13, ), 'cross', seed=5)
        exact = False
    halotab = make_tabcorr(table, **make_kwargs)
    handle = halotab.to_device().handle

    def option(name, value):
        _lib.check(lib.tc_table_set_option(handle, name, value))
    n = 9000 + 37                       # (not a multiple of anything)
    theta = synthetic.zheng07_draws(n, seed=17)

    def parts(result, count):
        """(ngal, xi) as two 2-D arrays, one row per draw."""
        ngal, xi = result
        if isinstance(ngal, dict):
            return (np.stack([ngal[k] for k in ngal], -1).reshape(count, -1),
                    np.stack([xi[k] for k in xi], 1).reshape(count, -1))
        return np.asarray(ngal).reshape(count, 1), np.asarray(xi).reshape(count, -1)
    loose = 10 * rtol if rtol > 1e-8 else 1e-12
    results = {}
    for chunks in (0, 1, 3, 8, -1):
        option(b'sync_chunks', chunks)
        results[chunks] = parts(halotab.predict_batch(theta, **kwargs), n)
    # the oracle on draws of every chunk and at the boundaries of 3 and 8 chunks
    index = np.unique(np.r_[0:2, 1150:1154, 3070:3074, 4500:4502, 6142:6146, n - 2:n])
    want = parts(oracle.predict_zheng07_batch(table, theta[index], **kwargs), len(index))
    for chunks, got in results.items():
        for part in (0, 1):
            assert_rel(got[part][index], want[part], rtol, 'chunks %d vs oracle' % chunks,
                       floor=1e-13)
            assert_rel(got[part], results[1][part], loose, 'chunks %d vs one piece' % chunks,
                       floor=1e-13)
    if exact:
        for chunks in (0, 3, 8):
            for part in (0, 1):
                assert np.array_equal(results[chunks][part], results[1][part]), chunks
    # a small call with more chunks requested than it has tiles of 64 draws
    option(b'sync_chunks', 64)
    few = parts(halotab.predict_batch(theta[:100], **kwargs), 100)
    option(b'sync_chunks', 0)
    ref = parts(halotab.predict_batch(theta[:100], **kwargs), 100)
    for part in (0, 1):
        assert_rel(few[part], ref[part], loose, floor=1e-13)


@pytest.mark.parametrize('mode', ['auto', 'cross'])
def test_synchronous_interpolator_calls_in_chunks(mode):
    """Interpolator.predict_batch on host arrays in chunks (interp.cpp: interp_chunked): the
    oracle on draws of every chunk, any number of chunks against one piece to rounding, ragged
    sizes, separated by galaxy type."""
    from tabcorr_amd import Interpolator, synthetic, _lib
    from oracle import tabcorr_oracle as oracle
    lib = _lib.load()
    tables, keys, points = synthetic.synthetic_interpolator((4, 4), 12, 1, (7, ), mode, seed=3)
    interp = Interpolator([make_tabcorr(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    n = 5000 + 13
    theta = synthetic.zheng07_draws(n, seed=4)
    rng = np.random.default_rng(5)
    x = np.stack([rng.uniform(xp[0], xp[-1], size=n) for xp in interp.xp], axis=-1)
    device = interp.to_device()
    results = {}
    for chunks in (0, 1, 3, 7, -1):
        for table in device.tables:
            _lib.check(lib.tc_table_set_option(table.handle, b'sync_chunks', chunks))
        results[chunks] = interp.predict_batch(theta, x)
    index = np.unique(np.r_[0:2, 1700:1703, 2500:2502, n - 2:n])
    setup = oracle.interpolator_setup(tables, points)
    expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[index], x[index])
    for chunks, (ngal, xi) in results.items():
        assert_rel(ngal[index], expect[0], RTOL, 'chunks %d' % chunks)
        assert_rel(xi[index], expect[1], RTOL, 'chunks %d' % chunks, floor=1e-12)
        assert_rel(ngal, results[1][0], 1e-12, 'chunks %d vs one piece' % chunks)
        assert_rel(xi, results[1][1], 1e-11, 'chunks %d vs one piece' % chunks, floor=1e-12)
    for table in device.tables:
        _lib.check(lib.tc_table_set_option(table.handle, b'sync_chunks', 0))
    ngal_s, xi_s = interp.predict_batch(theta[:3000], x[:3000], separate_gal_type=True)
    total = sum(xi_s[key] for key in xi_s)
    assert_rel(total, results[1][1][:3000], 1e-10, floor=1e-12)


def test_predict_joint_over_tables_of_every_kind():
    """TabCorr.predict_joint with three tables -- mode auto, mode cross and a float32 table the
    un-batched kernels cannot serve (it goes through the batched path inside the same call) --
    returns what predict(model) returns for each, bit for bit; the same table twice is refused;
    a model the device cannot evaluate falls back to one predict() per table."""
    from tabcorr_amd import TabCorr, Zheng07Model, synthetic
    tables = [synthetic.synthetic_table(20, 1, (9, ), 'auto', seed=1),
              synthetic.synthetic_table(20, 1, (7, ), 'cross', seed=2),
              synthetic.synthetic_table(12, 1, (6, 5), 'auto', seed=3)]
    halotabs = [make_tabcorr(tables[0]), make_tabcorr(tables[1]),
                make_tabcorr(tables[2], compute_dtype='float32')]
    model = Zheng07Model(redshift=tables[0]['attrs']['redshift'])
    theta = synthetic.zheng07_draws(40, seed=9)
    keys = ('logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha')
    for mode in (False, 'auto', True):
        for tab in halotabs[:2]:
            tab.set_resident(mode)
        for i in range(40):
            for key, value in zip(keys, theta[i]):
                model.param_dict[key] = value
            single = [tab.predict(model) for tab in halotabs]
            joint = TabCorr.predict_joint(halotabs, model)
            assert len(joint) == 3
            for (n1, x1), (n2, x2), tab in zip(single, joint, halotabs):
                assert n1 == n2 and np.array_equal(x1, x2), (mode, i)
                assert x2.shape == tuple(tab.tpcf_shape)
    with pytest.raises(ValueError):
        TabCorr.predict_joint([halotabs[0], halotabs[0]], model)
    assert len(TabCorr.predict_joint(halotabs[:1], model)) == 1
