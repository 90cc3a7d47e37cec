"""Helpers and secondary legs of bench.py.

bench.py itself times the headline (BASELINE configs[1]) and prints ONE short JSON line; what
lives here are the shared helpers (events-based kernel timing, the committed PMC summaries, a
thin device-array helper) and the measurements that go to the sidecar ``bench_detail.json``:
the host-to-host rate through the asynchronous entry points, the un-batched calls of the
reference's usage pattern, the pair counter of the tabulation step and the other BASELINE
configurations / the reference's own table shapes (``bench.py --only-config TAG``).
"""

import ctypes
import os
import re
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# FP64 peak of MI355X: 78.6 TFLOP/s (AMD datasheet, vector = matrix; the local
# microbenchmarks in tools/micro measure 78.0 for v_mfma_f64_16x16x4); FP32 matrix peak
# 157.3 TFLOP/s (/opt/skills/guides/MI355X_MICROARCH.md).
FP64_PEAK_TFLOPS = 78.6
FP32_PEAK_TFLOPS = 157.3
N_PRIM, N_SEC, N_R = 50, 1, 19
N_GAUSS = 10
FLAG_SEPARATE, FLAG_ASSEMBIAS = 1, 4
ROOFLINE_LAUNCHES = 1000
ROOFLINE_WARM_SECONDS = 0.25


def pair_flops(n_bins, n_r):
    """Algorithmic flop of one draw's contraction: 2 R P + 3 P."""
    n_pairs = n_bins * (n_bins + 1) // 2
    return 2.0 * n_r * n_pairs + 3.0 * n_pairs


def pmc_file(tag):
    """Newest committed PMC summary of a configuration: profiles/rNN_pmc_counters[_tag].txt
    (tools/profile_round.sh; tag '' = the headline configuration)."""
    import glob
    suffix = '_pmc_counters%s.txt' % ('_' + tag if tag else '')
    files = sorted(f for f in glob.glob(os.path.join(REPO, 'profiles', 'r[0-9][0-9]' + suffix)))
    return files[-1] if files else None


def matrix_pipe_busy(kernel, step_seconds, simds=1024, clock_hz=2.4e9):
    """Share of the timed region in which a SIMD's matrix pipe was busy: the counter
    SQ_VALU_MFMA_BUSY_CYCLES per launch of `kernel` (committed PMC passes) over the SIMD cycles
    of one step at the nominal clock (one launch per step)."""
    cycles = pmc_counter(kernel, 'SQ_VALU_MFMA_BUSY_CYCLES')
    if cycles is None:
        return None
    return {'value': cycles / (simds * step_seconds * clock_hz),
            'what': 'SQ_VALU_MFMA_BUSY_CYCLES per launch (%s) / (%d SIMDs x step time x %.1f GHz)'
                    % (os.path.relpath(pmc_file(''), REPO), simds, clock_hz * 1e-9)}


# Look-ups that found a committed PMC file for the configuration but NOT the kernel that was
# timed (the profiles were taken with another build of the library): bench.py reports them under
# `failed_legs` instead of quietly printing "traffic": null.
pmc_failures = []


def pmc_missing(kernel, what, path):
    message = '%s of %s: not in %s (profiles of another build?)' % (
        what, kernel, os.path.relpath(path, REPO))
    if message not in pmc_failures:
        pmc_failures.append(message)
        print('detail pmc look-up failed: ' + message, file=sys.stderr)


def pmc_counter(kernel, counter, tag=''):
    """Mean of one raw counter per launch of `kernel` from the committed rocprofv3 --pmc passes
    of configuration `tag`; None when there is no committed file for the configuration, and
    None + an entry in `pmc_failures` when the file does not hold the kernel."""
    path = pmc_file(tag)
    if path is None:
        return None
    seen_counter = False
    for line in open(path).read().splitlines():
        match = re.match(r'(.*?)\s+(\w+)\s+n=\s*\d+\s+mean=([0-9.e+]+)', line)
        if match and match.group(2) == counter:
            seen_counter = True
            if same_kernel(kernel, match.group(1)):
                return float(match.group(3))
    if seen_counter:
        pmc_missing(kernel, counter, path)
    return None


def same_kernel(kernel, printed):
    """Whether a row of tools/pmc_summary.py (kernel names cut to 96 characters; 48 in the files
    of earlier rounds) is `kernel`'s."""
    printed = printed.replace('void ', '').strip()
    name = kernel.split('(')[0]
    if '(' in printed:                      # (whole up to the argument list)
        return printed.split('(')[0] == name
    return name.startswith(printed)


def pmc_traffic(kernel, tag=''):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 --pmc passes of
    configuration `tag` (FETCH_SIZE and WRITE_SIZE in KB; FETCH_SIZE doubled per the gfx950
    correction of the MI355X guide) and the file they come from; (None, None) when no
    committed file holds that kernel."""
    path = pmc_file(tag)
    if path is None:
        return None, None
    values = {}
    for line in open(path).read().splitlines():
        match = re.match(r'(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*\d+\s+mean=([0-9.e+]+)', line)
        if match and same_kernel(kernel, match.group(1)):
            values[match.group(2)] = float(match.group(3))
    if len(values) != 2:
        pmc_missing(kernel, 'FETCH_SIZE / WRITE_SIZE', path)
        return None, None
    return ((2.0 * values['FETCH_SIZE'] + values['WRITE_SIZE']) * 1024.0,
            os.path.relpath(path, REPO))


class Device:
    """Thin helper over the C ABI for device-resident arrays."""

    def __init__(self, lib, _lib):
        self.lib = lib
        self._lib = _lib
        self.allocations = []

    def malloc(self, count):
        ptr = ctypes.c_void_p()
        self._lib.check(self.lib.tc_device_malloc(ctypes.byref(ptr), max(1, count) * 8))
        self.allocations.append(ptr)
        return ptr

    def upload(self, array):
        array = np.ascontiguousarray(array, dtype=np.float64)
        ptr = self.malloc(array.size)
        self._lib.check(self.lib.tc_memcpy_h2d(
            ptr, array.ctypes.data_as(ctypes.c_void_p), array.nbytes))
        return ptr

    def download(self, ptr, count):
        host = np.empty(count)
        self._lib.check(self.lib.tc_memcpy_d2h(
            host.ctypes.data_as(ctypes.c_void_p), ptr, host.nbytes))
        return host

    def free_all(self):
        for ptr in self.allocations:
            if ptr.value:
                self.lib.tc_device_free(ptr)
        self.allocations = []


def kernel_time(lib, _lib, timer_handle, launch, synchronize, warm_seconds=ROOFLINE_WARM_SECONDS,
                n_launches=ROOFLINE_LAUNCHES, max_seconds=1.0):
    """Mean duration (ms) of the contraction kernel inside `launch()`, serialised: load the
    chip for `warm_seconds` first (the power management needs tens of milliseconds to
    settle), then time `n_launches` launches (fewer when they would take more than
    `max_seconds`) with per-launch start / stop events."""
    launch()
    synchronize()
    t0 = time.perf_counter()
    launch()
    synchronize()
    per_call = max(time.perf_counter() - t0, 1e-6)
    for _ in range(int(warm_seconds / per_call) + 1):
        launch()
    n = max(10, min(n_launches, int(max_seconds / per_call)))
    synchronize()
    _lib.check(lib.tc_table_timer_begin(timer_handle, 1))
    t0 = time.perf_counter()
    for _ in range(n):
        launch()
    synchronize()
    wall_ms = (time.perf_counter() - t0) / n * 1e3
    ms = ctypes.c_float()
    _lib.check(lib.tc_table_timer_end(timer_handle, ctypes.byref(ms)))
    count = ctypes.c_int()
    kernel_ms = ctypes.c_float()
    _lib.check(lib.tc_table_kernel_time(timer_handle, ctypes.byref(count), ctypes.byref(kernel_ms)))
    return kernel_ms.value, count.value, wall_ms


def sustained(launch, synchronize, seconds=0.4, warm_seconds=0.15):
    """Seconds per `launch()` in a sustained stream of calls."""
    launch()
    synchronize()
    t0 = time.perf_counter()
    launch()
    synchronize()
    per_call = max(time.perf_counter() - t0, 1e-6)
    for _ in range(int(warm_seconds / per_call) + 1):
        launch()
    synchronize()
    n = max(5, int(seconds / per_call))
    t0 = time.perf_counter()
    for _ in range(n):
        launch()
    synchronize()
    return (time.perf_counter() - t0) / n


def time_calls(call, seconds=0.5, warm=3):
    for _ in range(warm):
        call()
    t0 = time.perf_counter()
    call()
    per_call = max(time.perf_counter() - t0, 1e-6)
    n = max(3, int(seconds / per_call))
    t0 = time.perf_counter()
    for _ in range(n):
        call()
    return (time.perf_counter() - t0) / n


# ---- SURVEY 8d metric through the asynchronous entry points ------------------------------------

def host_pipelined(lib, _lib, handle, table, n_draws, data_vector, precision, seconds=0.5,
                   depth=6, ring=8):
    """theta in page-locked host memory -> (ngal, xi) / (ngal, chi2) in page-locked host memory
    through tc_predict_zheng07_batch_async / tc_chi2_zheng07_batch_async + tc_table_wait:
    `depth` calls in flight over a ring of `ring` distinct buffer sets, every call with its
    own draws; each ticket is waited for before its buffers are reused.  The last ring's
    results are checked against the CPU oracle."""
    from tabcorr_amd import synthetic, pinned_array, pinned_empty
    from oracle import tabcorr_oracle as oracle
    thetas = [pinned_array(synthetic.zheng07_draws(n_draws, seed=500 + i)) for i in range(ring)]
    ngals = [pinned_empty(n_draws) for _ in range(ring)]
    xis = [pinned_empty((n_draws, N_R)) for _ in range(ring)]
    chis = [pinned_empty(n_draws) for _ in range(ring)]
    p_theta = [_lib.as_double_p(a) for a in thetas]
    p_ngal = [_lib.as_double_p(a) for a in ngals]
    p_xi = [_lib.as_double_p(a) for a in xis]
    p_chi = [_lib.as_double_p(a) for a in chis]
    data_p, precision_p = _lib.as_double_p(data_vector), _lib.as_double_p(precision)
    ticket = ctypes.c_int64()
    ref = ctypes.byref(ticket)

    def run(chi2, total):
        tickets = [None] * ring
        start = time.perf_counter()
        for k in range(total):
            s = k % ring
            if k >= depth:
                _lib.check(lib.tc_table_wait(handle, tickets[(k - depth) % ring]))
            if chi2:
                _lib.check(lib.tc_chi2_zheng07_batch_async(
                    handle, p_theta[s], 5, n_draws, N_GAUSS, 0, data_p, precision_p, p_ngal[s],
                    p_chi[s], ref))
            else:
                _lib.check(lib.tc_predict_zheng07_batch_async(
                    handle, p_theta[s], 5, n_draws, N_GAUSS, 0, p_ngal[s], p_xi[s], ref))
            tickets[s] = ticket.value
        for k in range(max(0, total - depth), total):
            _lib.check(lib.tc_table_wait(handle, tickets[k % ring]))
        return (time.perf_counter() - start) / total

    out = {}
    for chi2, name, payload in ((False, 'host_to_host_pipelined', '%d B out' % (8 * (1 + N_R))),
                                (True, 'host_to_host_chi2', '16 B out')):
        run(chi2, 300)
        per = run(chi2, 100)
        per = run(chi2, max(100, int(seconds / per)))
        check = ring - 1
        expect = oracle.predict_zheng07_batch(table, thetas[check][:2])
        if chi2:
            delta = expect[1] - data_vector
            want = np.einsum('bi,ij,bj->b', delta, precision, delta)
            parity = float(np.max(np.abs(chis[check][:2] / want - 1)))
        else:
            parity = float(np.max(np.abs(xis[check][:2] / expect[1] - 1)))
        out[name] = {
            'value': n_draws / per, 'unit': 'calls/s', 'us_per_call': per * 1e6,
            'calls_in_flight': depth, 'parity_max_rel_vs_oracle': parity,
            'what': 'tc_%s_zheng07_batch_async + tc_table_wait: %d draws per call (40 B in, %s '
                    'per draw), page-locked caller buffers, PCIe included (SURVEY.md 8d)'
                    % ('chi2' if chi2 else 'predict', n_draws, payload)}
    return out


# ---- latency mode --------------------------------------------------------------------------

def unbatched(make, table, synthetic, Interpolator):
    """One predict(model) per call, the reference's usage pattern (README.md:72-75)."""
    from tabcorr_amd import Zheng07Model
    halotab = make(table)
    model = Zheng07Model()
    halotab.predict(model)
    count = [0]

    def call():
        count[0] += 1
        model.param_dict['logMmin'] = 12.0 + 1e-5 * (count[0] % 1000)
        halotab.predict(model)
    # the library's default: a loop of un-batched calls is moved to the resident kernel by
    # itself (option "resident" = 2); then one launch per call (set_resident(False)) ...
    default = time_calls(call, seconds=0.3, warm=50)
    halotab.set_resident(False)
    single = time_calls(call, seconds=0.3, warm=50)
    # ... and the resident launch asked for (TabCorr.set_resident(True): the call writes its
    # parameters into a mailbox in page-locked memory, no launch per call), checked against
    # the one-launch-per-call result
    expect = halotab.predict(model)
    resident, resident_parity = None, None
    try:
        halotab.set_resident(True)
        got = halotab.predict(model)
        resident_parity = float(max(abs(got[0] / expect[0] - 1),
                                    np.max(np.abs(got[1] / expect[1] - 1))))
        resident = time_calls(call, seconds=0.3, warm=50)
    except Exception as error:   # noqa: BLE001 -- a secondary measurement must not end the bench
        resident_parity = 'failed: %s' % error
    finally:
        try:
            halotab.set_resident(False)
        except Exception:   # noqa: BLE001
            pass
    tables, keys, points = synthetic.synthetic_interpolator((5, 5), N_PRIM, N_SEC, (N_R, ),
                                                            'auto', seed=7)
    interp = Interpolator([make(t) for t in tables],
                          {k: points[:, d] for d, k in enumerate(keys)})
    for d, key in enumerate(keys):
        model.param_dict[key] = float(np.mean(points[:, d]))
    interp.predict(model)

    def call_interp():
        count[0] += 1
        model.param_dict['logMmin'] = 12.0 + 1e-5 * (count[0] % 1000)
        interp.predict(model)
    grid = time_calls(call_interp, seconds=0.3, warm=20)
    # an ensemble sampler's step: n independent walkers per call, one launch
    # (tc_predict_zheng07_many behind predict_batch)
    walkers = {}
    for n in (1, 16, 64):
        theta = synthetic.zheng07_draws(n, seed=70 + n)
        seconds = time_calls(lambda: halotab.predict_batch(theta), seconds=0.2, warm=50)
        walkers['%d' % n] = {'us_per_call': seconds * 1e6, 'us_per_walker': seconds * 1e6 / n}
    # the same steps served by the resident ENSEMBLE kernel (2 .. 256 walkers per call, no
    # launch; checked against the launched result)
    resident_walkers = {}
    try:
        for n in (64, 256):
            theta = synthetic.zheng07_draws(n, seed=70 + n)
            halotab.set_resident(False)
            expect = halotab.predict_batch(theta)
            launched = time_calls(lambda: halotab.predict_batch(theta), seconds=0.2, warm=50)
            halotab.set_resident(True)
            got = halotab.predict_batch(theta)
            served = time_calls(lambda: halotab.predict_batch(theta), seconds=0.2, warm=50)
            resident_walkers['%d' % n] = {
                'us_per_call': served * 1e6, 'us_per_call_launched': launched * 1e6,
                'max_rel_vs_launched': float(max(np.max(np.abs(got[0] / expect[0] - 1)),
                                                 np.max(np.abs(got[1] / expect[1] - 1))))}
    except Exception as error:   # noqa: BLE001 -- a secondary measurement must not end the bench
        resident_walkers['failed'] = str(error)
    finally:
        try:
            halotab.set_resident(False)
        except Exception:   # noqa: BLE001
            pass
    two = two_tables(Zheng07Model)
    return {'predict_model': default * 1e6,
            'two_tables_per_step': two,
            'predict_model_calls_per_sec': 1.0 / default,
            'predict_model_one_launch_per_call': single * 1e6,
            'predict_model_resident': None if resident is None else resident * 1e6,
            'predict_batch_walkers_resident': resident_walkers,
            'predict_model_resident_max_rel_vs_one_launch_per_call': resident_parity,
            'interpolator_5x5_predict_model': grid * 1e6,
            'predict_batch_walkers': walkers,
            'unit': 'us per call (Python API, host model -> host results)'}


def two_tables(Zheng07Model):
    """The reference's documented likelihood step: TWO tables per model evaluation
    (docs/guides/overview.rst:86-92: halotab_wp.predict(model), then halotab_ds.predict(model))
    -- its own example tables (tests/golden/bolplanck_{wp,ds}.hdf5: mode auto and mode cross),
    default options: alternately, and posted together (TabCorr.predict_joint); us per PAIR."""
    from tabcorr_amd import TabCorr
    golden = os.path.join(REPO, 'tests', 'golden')
    try:
        wp = TabCorr.read(os.path.join(golden, 'bolplanck_wp.hdf5'))
        ds = TabCorr.read(os.path.join(golden, 'bolplanck_ds.hdf5'))
        model = Zheng07Model(redshift=wp.attrs['redshift'])
        count = [0]

        def step():
            count[0] += 1
            model.param_dict['logMmin'] = 12.0 + 1e-5 * (count[0] % 1000)

        def alternately():
            step()
            wp.predict(model)
            ds.predict(model)

        def joint():
            step()
            TabCorr.predict_joint([wp, ds], model)
        record = {}
        for tab in (wp, ds):
            tab.set_resident(False)
        record['alternately_one_launch_per_call'] = time_calls(alternately, seconds=0.2,
                                                               warm=50) * 1e6
        record['predict_joint_one_launch_per_call'] = time_calls(joint, seconds=0.2,
                                                                 warm=50) * 1e6
        model.param_dict['logMmin'] = 12.3456
        expect = (wp.predict(model), ds.predict(model))
        for tab in (wp, ds):
            tab.set_resident('auto')           # (= a new handle's state: nothing switched on)
        record['alternately'] = time_calls(alternately, seconds=0.3, warm=200) * 1e6
        record['predict_joint'] = time_calls(joint, seconds=0.3, warm=200) * 1e6
        record['unit'] = 'us per pair of predictions (Python API, host model -> host results)'
        # the model of `expect` again, posted together: the bits of one launch per call
        model.param_dict['logMmin'] = 12.3456
        both = TabCorr.predict_joint([wp, ds], model)
        record['predict_joint_same_bits_as_predict'] = bool(
            all(a[0] == b[0] and np.array_equal(a[1], b[1]) for a, b in zip(expect, both)))
        return record
    except Exception as error:   # noqa: BLE001 -- a secondary measurement must not end the bench
        return {'failed': '%s: %s' % (type(error).__name__, error)}


# ---- pair counting for the tabulation step ---------------------------------------

def tabulation(cpu_seconds):
    """DD(r_p, pi) on the GPU (tabcorr/corrfunc.py:62-84, tabcorr/tabcorr.py:846-922): a
    clustered sample in a 250 Mpc/h box, 19 r_p bins up to 30, pi_max = 40; the single pair
    count, all 100 x 100 halo-bin pairs in one pass, and the brute-force NumPy oracle on a
    subsample as the CPU baseline (bit-exact check included)."""
    from tabcorr_amd import corrfunc
    from oracle import paircount_oracle
    rng = np.random.default_rng(3)
    box, n = 250.0, 400000
    rp_bins = np.logspace(-1, np.log10(30.0), 20)
    centres = rng.uniform(0, box, (n // 60, 3))
    pos = np.mod(centres[rng.integers(0, len(centres), n)] + rng.normal(0, 3.0, (n, 3)), box)
    label = rng.integers(0, 100, n)
    corrfunc.pair_count_rppi(pos[:1000], rp_bins, 40.0, None, box)
    t0 = time.perf_counter()
    counts = corrfunc.pair_count_rppi(pos, rp_bins, 40.0, None, box)
    auto_seconds = time.perf_counter() - t0
    pairs = int(counts.sum())
    order = np.argsort(label, kind='stable')
    bins = np.split(pos[order], np.cumsum(np.bincount(label, minlength=100))[:-1])
    t0 = time.perf_counter()
    matrix = corrfunc.pair_count_matrix(bins, rp_bins, 40.0, box)
    matrix_seconds = time.perf_counter() - t0
    out = {'workload': '%d clustered points, box 250, 19 rp bins to 30, pi_max 40, host arrays '
                       'in, counts out (cell sort on the host included)' % n,
           'pairs_counted': pairs,
           'auto_count_ms': auto_seconds * 1e3, 'auto_pairs_per_sec': pairs / auto_seconds,
           'all_100x100_bin_pairs_ms': matrix_seconds * 1e3,
           'all_bin_pairs_pairs_per_sec': pairs / matrix_seconds,
           'all_bin_pairs_consistent': bool(int(matrix.sum()) == pairs)}
    if cpu_seconds > 0:
        sub = pos[:6000]
        t0 = time.perf_counter()
        expect = paircount_oracle.pair_count_rppi(sub, None, box, rp_bins, 40.0)
        spent = time.perf_counter() - t0
        got = corrfunc.pair_count_rppi(sub, rp_bins, 40.0, None, box)
        out['cpu_baseline'] = {
            'value': len(sub)**2 / spent, 'unit': 'pair tests/s', 'cores': 1, 'kind': 'port',
            'sample': 'brute-force NumPy oracle on %d points (%.1f s)' % (len(sub), spent),
            'gpu_bit_exact': bool(np.array_equal(got, expect))}
        out['gpu_pair_tests_per_sec_equivalent'] = float(n)**2 / auto_seconds
    return out


# ---- BASELINE configs[2], [3], [4] ----------------------------------------------------------

CONFIG_TAGS = ('cfg3', 'cfg4', 'cfg5f32', 'cfg5f64', 'ds4', 'ds1', 'wp', 'db')


def other_configs(lib, _lib, make, synthetic, Interpolator, cpu_seconds, only=None, lanes=0,
                  options=()):
    """BASELINE configs[2] ('cfg3': separate + assembly bias), configs[3] ('cfg4': one GPU's
    share of the 5 x 5 interpolator), configs[4] in float32 and float64 ('cfg5f32',
    'cfg5f64'): device rate, host rate, the dominant kernel serialised with per-launch
    events, its roofline fraction and committed traffic, the whole step's fraction, an oracle
    spot check of the first and last draws of the timed batch, the CPU port.  `only`: one
    tag (what tools/profile_round.sh wraps in rocprofv3)."""
    from oracle import tabcorr_oracle as oracle
    out = {}
    dev = Device(lib, _lib)
    theta = synthetic.zheng07_draws(10000, seed=1)
    cpu_budget = min(3.0, cpu_seconds / 4) if cpu_seconds > 0 else 0.0

    def wanted(tag):
        return only is None or only == tag

    def cpu_rate(call, n_max=100000):
        if cpu_budget <= 0:
            return None
        call(0)
        start = time.perf_counter()
        count = 0
        while time.perf_counter() - start < cpu_budget and count < n_max:
            call(count)
            count += 1
        spent = time.perf_counter() - start
        return {'value': count / spent, 'unit': 'calls/s', 'cores': 1, 'kind': 'port',
                'sample': '%d sequential predict() calls in %.1f s' % (count, spent)}

    def measure(name, tag, what, timer_handle, launch, synchronize, host_call, n_draws, flop,
                peak, kernel, dtype, cpu, parity, fused_kernel=None, bound='mfma',
                set_options=None, host_call_kept=None):
        """`kernel`: the dominant kernel of the three-kernel form; `fused_kernel`: the one-launch
        kernel the library may choose for the pipelined calls of this configuration (its name
        as rocprofv3 prints it).  The record's `kernel` is the one that ran in the timed
        region; the serialised three-kernel figures sit under `three_kernel_path`.
        `set_options(name, value)`: sets an option on every table handle of the workload.
        `host_call` returns fresh NumPy arrays, `host_call_kept` writes into arrays the caller
        keeps (out=): a fresh 61 MB array is 4 ms of page faults per call."""
        if set_options is None:
            def set_options(key, value):
                _lib.check(lib.tc_table_set_option(timer_handle, key, value))
        if lanes > 0:
            _lib.check(lib.tc_table_set_option(timer_handle, b'lanes', lanes))
        for option in options:          # (developer A/B: --option name=value)
            key, value = option.split('=')
            set_options(key.encode(), int(value))
        device_seconds = sustained(launch, synchronize)
        shape = [ctypes.c_int() for _ in range(4)]
        lib.tc_table_last_launch(timer_handle, *[ctypes.byref(v) for v in shape])
        # (the pipelined calls of device_calls_per_sec: one launch per call where the library
        # chose a one-launch form -- no slabs of partial sums --, else the three kernels)
        one_launch = shape[2].value == 0 and shape[1].value > 0
        pipelined = ('one launch per call: %s, %d workgroups of %d waves'
                     % (fused_kernel or 'one-launch kernel', shape[0].value, shape[1].value)
                     if one_launch else 'occupation, contraction, finalisation kernels')
        # the dominant kernel of THAT stream of calls: per-launch start / stop events, launches
        # of different lanes overlapping as in the timed region
        pipelined_ms, n_pipelined, pipelined_wall_ms = kernel_time(
            lib, _lib, timer_handle, launch, synchronize, n_launches=300, max_seconds=0.6)
        # ... and the three-kernel form, kernels serialised (one lane, one-launch forms off)
        _lib.check(lib.tc_table_set_option(timer_handle, b'pipeline', 0))
        user_fused = dict(o.split('=') for o in options).get('fused', '1')
        set_options(b'fused', 0)
        kernel_ms, n_launch, _ = kernel_time(lib, _lib, timer_handle, launch, synchronize,
                                             n_launches=300, max_seconds=0.6)
        set_options(b'fused', int(user_fused))
        _lib.check(lib.tc_table_set_option(timer_handle, b'pipeline', 1))
        host_seconds = time_calls(host_call, seconds=0.4, warm=3)
        launch()
        synchronize()
        ran = fused_kernel if one_launch and fused_kernel else kernel
        traffic, source = pmc_traffic(ran, tag)
        concurrency = pipelined_ms / pipelined_wall_ms
        record = {
            'workload': what, 'tag': tag, 'dtype': dtype, 'draws_per_call': n_draws,
            'device_calls_per_sec': n_draws / device_seconds,
            'us_per_step': device_seconds * 1e6,
            'device_calls_run_as': pipelined,
            'host_to_host_calls_per_sec': n_draws / host_seconds,
            'kernel': ran,
            'kernel_us': pipelined_ms * 1e3, 'launches_timed': n_pipelined,
            'concurrent_launches': concurrency,
            'kernel_method': 'per-launch start/stop events (hipExtLaunchKernelGGL) in the '
                             'pipelined stream of calls of device_calls_per_sec; launches of '
                             'different lanes overlap: concurrent_launches = kernel_us / wall '
                             'time per launch',
            'flop_per_launch': flop, 'peak_tflops': peak,
            'frac_by_duration': flop / (pipelined_ms * 1e-3) / 1e12 / peak,
            'frac': flop / (pipelined_ms / max(concurrency, 1.0) * 1e-3) / 1e12 / peak,
            'frac_method': 'flop_per_launch / (kernel_us / concurrent_launches) / peak: a '
                           "launch's share of the chip; frac_by_duration = flop_per_launch / "
                           'kernel_us / peak',
            'step_frac': flop / device_seconds / 1e12 / peak,
            'three_kernel_path': {
                'kernel': kernel, 'kernel_us': kernel_ms * 1e3, 'launches_timed': n_launch,
                'frac': flop / (kernel_ms * 1e-3) / 1e12 / peak,
                'note': 'dominant kernel of the three-kernel form, kernels serialised (one '
                        'lane, one-launch forms off): flop_per_launch / kernel_us / peak'},
            'traffic': traffic, 'traffic_source': source,
            'parity_max_rel_vs_oracle': parity(), 'cpu_baseline': cpu}
        insts = pmc_counter(ran, 'SQ_INSTS_VALU', tag)
        if insts is not None:
            # vector-ALU issue: one wave instruction occupies its SIMD for 4 cycles
            record['valu'] = {
                'wave_instructions_per_launch': insts,
                'frac': insts * 4.0 / (1024 * device_seconds * 2.4e9),
                'what': 'SQ_INSTS_VALU per launch of %s (%s) x 4 cycles / (1024 SIMDs x step '
                        'time x 2.4 GHz): share of the vector issue slots of the timed region '
                        'this kernel fills' % (ran, os.path.relpath(pmc_file(tag), REPO))}
            mfma = pmc_counter(ran, 'SQ_VALU_MFMA_BUSY_CYCLES', tag)
            if mfma is not None:
                record['valu']['matrix_pipe_busy'] = mfma / (1024 * device_seconds * 2.4e9)
        if host_call_kept is not None:
            record['host_to_host_kept_arrays_calls_per_sec'] = n_draws / time_calls(
                host_call_kept, seconds=0.4, warm=3)
        record['bound'] = bound
        out[name] = record

    def rel(actual, expect, floor=1e-14):
        scale = floor * np.max(np.abs(expect))
        return float(np.max(np.abs(actual - expect) / np.maximum(np.abs(expect), scale)))

    # configs[2]: separate_gal_type + assembly bias on a 2-D halo-bin grid
    if wanted('cfg3'):
        table3 = synthetic.synthetic_table(50, 2, (N_R, ), 'auto', seed=3)
        theta7 = np.hstack([theta, np.random.default_rng(0).uniform(-1, 1, (10000, 2))])
        tab3 = make(table3)
        h3 = tab3.to_device().handle
        d_theta7 = dev.upload(theta7)
        d_ngal, d_xi = dev.malloc(2 * 10000), dev.malloc(3 * N_R * 10000)
        flags3 = FLAG_SEPARATE | FLAG_ASSEMBIAS
        cache3 = {}
        ends = np.r_[0:2, 9998:10000]

        def parity3():
            ngal = dev.download(d_ngal, 2 * 10000).reshape(10000, 2)
            xi = dev.download(d_xi, 3 * N_R * 10000).reshape(10000, 3, N_R)
            expect = oracle.predict_zheng07_batch(table3, theta7[ends, :5],
                                                  separate_gal_type=True,
                                                  assembias=theta7[ends, 5:])
            worst = max(rel(ngal[ends, i], expect[0][key])
                        for i, key in enumerate(('centrals', 'satellites')))
            return max([worst] + [rel(xi[ends, i], expect[1][key]) for i, key in enumerate(
                ('centrals-centrals', 'centrals-satellites', 'satellites-satellites'))])
        measure('configs[2]', 'cfg3', 'separate_gal_type=True + Heaviside assembly bias, 50 x 2 x '
                '{cen,sat} bins (G=200, P=20100), 19 rp bins, 10^4 draws', h3,
                lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                    h3, d_theta7, 7, 10000, N_GAUSS, flags3, d_ngal, d_xi)),
                lambda: _lib.check(lib.tc_table_synchronize(h3)),
                lambda: tab3.predict_batch(theta7, separate_gal_type=True, assembias=True),
                10000, 10000 * pair_flops(200, N_R), FP64_PEAK_TFLOPS,
                'tc::contract_quad_kernel<5, false>', 'f64',
                cpu_rate(lambda i: oracle.predict_zheng07(
                    table3, theta7[i % 10000, :5], separate_gal_type=True,
                    assembias=theta7[i % 10000, 5:], cache=cache3)), parity3,
                fused_kernel='tc::predict_fused_kernel<10, 5, true, false, false, 8, 32, true, 0>')
        del tab3

    # configs[3]: one GPU's share (12 500 draws) of the 5 x 5 interpolator
    if wanted('cfg4'):
        tables, keys, points = synthetic.synthetic_interpolator((5, 5), N_PRIM, N_SEC, (N_R, ),
                                                                'auto', seed=7)
        interp = Interpolator([make(t) for t in tables],
                              {k: points[:, d] for d, k in enumerate(keys)})
        n4 = 12500
        theta4 = synthetic.zheng07_draws(n4, seed=5)
        rng = np.random.default_rng(6)
        x4 = np.ascontiguousarray(np.stack(
            [rng.uniform(xp[0], xp[-1], size=n4) for xp in interp.xp], axis=-1))
        idev = interp.to_device()
        d_theta4, d_x4 = dev.upload(theta4), dev.upload(x4)
        d_ngal4, d_xi4 = dev.malloc(n4), dev.malloc(n4 * N_R)
        setup = oracle.interpolator_setup(tables, points)
        ends4 = np.r_[0:2, n4 - 2:n4]

        def parity4():
            expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta4[ends4],
                                                               x4[ends4])
            return max(rel(dev.download(d_ngal4, n4)[ends4], expect[0]),
                       rel(dev.download(d_xi4, n4 * N_R).reshape(n4, N_R)[ends4], expect[1],
                           floor=1e-12))
        measure('configs[3]', 'cfg4', 'Interpolator.predict over a 5 x 5 grid of configs[1] '
                "tables, one GPU's share of 10^5 draws (12 500)", idev.tables[0].handle,
                lambda: _lib.check(lib.tc_interp_predict_zheng07_batch_device(
                    idev.handle, d_theta4, 5, d_x4, n4, N_GAUSS, 0, d_ngal4, d_xi4)),
                lambda: _lib.check(lib.tc_interp_synchronize(idev.handle)),
                lambda: interp.predict_batch(theta4, x4), n4,
                n4 * 25 * pair_flops(100, N_R), FP64_PEAK_TFLOPS,
                'tc::contract_quad_kernel<5, true>', 'f64',
                cpu_rate(lambda i: oracle.interpolator_predict(
                    tables, setup, oracle.Zheng07(theta4[i % n4]), x4[i % n4])), parity4)
        del interp, idev

    # ---- the reference's own table shapes (throughput evidence on them) ----------
    golden = os.path.join(REPO, 'tests', 'golden')
    valu_peak = 1024 * 2.4e9 / 4 * 1e-12     # wave instructions per second (x 1e-12)

    # the reference's AbacusSummit fixture (tests/AbacusSummit/.../ds_efficient.hdf5): mode
    # cross, G = 1104 (280 mass bins x 2 percentile bins x {cen, sat}), 13 r values, a 4-table
    # interpolator over log_eta -- the step is all occupations (vector ALU), one launch per call
    if wanted('ds4') or wanted('ds1'):
        from tabcorr_amd import TabCorr
        interp = Interpolator.read(os.path.join(golden, 'ds_efficient.hdf5'))
        rng = np.random.default_rng(0)
        theta_ds = theta.copy()
        theta_ds[:, 0] = rng.uniform(12.5, 13.3, 10000)     # (a sample this table resolves)
        theta_ds[:, 3] = rng.uniform(13.6, 14.4, 10000)
        x_ds = np.ascontiguousarray(np.stack(
            [rng.uniform(xp[0], xp[-1], size=10000) for xp in interp.xp], axis=-1))
        ds_tables = [{'gal_type': t.gal_type.as_array(), 'tpcf_matrix': t.tpcf_matrix,
                      'tpcf_shape': t.tpcf_shape, 'attrs': t.attrs}
                     for t in interp.tabcorr_list]
        n_bins_ds, n_r_ds = len(ds_tables[0]['gal_type']), 13
        d_theta_ds, d_x_ds = dev.upload(theta_ds), dev.upload(x_ds)
        d_ngal_ds, d_xi_ds = dev.malloc(10000), dev.malloc(10000 * n_r_ds)
        ends_ds = np.r_[0:2, 9998:10000]
        # (the survey's accounting: contraction 2 R G + G per table, occupations 4 G n_gauss)
        flop_table = 2.0 * n_r_ds * n_bins_ds + n_bins_ds
        flop_occ = 4.0 * n_bins_ds * N_GAUSS
        if wanted('ds4'):
            idev = interp.to_device()
            # (row k of the grid table describes tabcorr_list[tabcorr_index[k]])
            points = np.zeros((len(ds_tables), len(interp.keys)))
            index = np.asarray(interp.param_dict_table['tabcorr_index'], dtype=int)
            for d, key in enumerate(interp.keys):
                points[index, d] = np.asarray(interp.param_dict_table[key], dtype=float)
            setup = oracle.interpolator_setup(ds_tables, points)

            def parity_ds4():
                expect = oracle.interpolator_predict_zheng07_batch(
                    ds_tables, setup, theta_ds[ends_ds], x_ds[ends_ds])
                return max(rel(dev.download(d_ngal_ds, 10000)[ends_ds], expect[0]),
                           rel(dev.download(d_xi_ds, 10000 * n_r_ds).reshape(
                               10000, n_r_ds)[ends_ds], expect[1], floor=1e-12))
            measure('reference fixture: AbacusSummit interpolator', 'ds4',
                    "the reference's tests/AbacusSummit/base_c000_ph000/0p50/ds_efficient.hdf5: "
                    'Interpolator over 4 tables, mode cross, G=1104 (280 mass x 2 percentile '
                    'bins x {cen,sat}), 13 r values, 10^4 draws', idev.tables[0].handle,
                    lambda: _lib.check(lib.tc_interp_predict_zheng07_batch_device(
                        idev.handle, d_theta_ds, 5, d_x_ds, 10000, N_GAUSS, 0, d_ngal_ds,
                        d_xi_ds)),
                    lambda: _lib.check(lib.tc_interp_synchronize(idev.handle)),
                    lambda: interp.predict_batch(theta_ds, x_ds), 10000,
                    10000 * (4 * flop_table + flop_occ), FP64_PEAK_TFLOPS,
                    'tc::occ_zheng07_kernel<10, false, false, true>', 'f64',
                    cpu_rate(lambda i: oracle.interpolator_predict(
                        ds_tables, setup, oracle.Zheng07(theta_ds[i % 10000]),
                        x_ds[i % 10000])), parity_ds4,
                    fused_kernel='tc::predict_cross_fused_kernel<8, false, false, true>',
                    bound='valu')
            del idev
        if wanted('ds1'):
            tab_ds = interp.tabcorr_list[0]
            h_ds = tab_ds.to_device().handle
            cache_ds = {}

            def parity_ds1():
                expect = oracle.predict_zheng07_batch(ds_tables[0], theta_ds[ends_ds])
                return max(rel(dev.download(d_ngal_ds, 10000)[ends_ds], expect[0]),
                           rel(dev.download(d_xi_ds, 10000 * n_r_ds).reshape(
                               10000, n_r_ds)[ends_ds], expect[1]))
            measure('reference fixture: AbacusSummit table', 'ds1',
                    'the first table of that file by itself: mode cross, G=1104, 13 r values, '
                    '10^4 draws', h_ds,
                    lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                        h_ds, d_theta_ds, 5, 10000, N_GAUSS, 0, d_ngal_ds, d_xi_ds)),
                    lambda: _lib.check(lib.tc_table_synchronize(h_ds)),
                    lambda: tab_ds.predict_batch(theta_ds), 10000,
                    10000 * (flop_table + flop_occ), FP64_PEAK_TFLOPS,
                    'tc::occ_zheng07_kernel<10, false, false, true>', 'f64',
                    cpu_rate(lambda i: oracle.predict_zheng07(
                        ds_tables[0], theta_ds[i % 10000], cache=cache_ds)), parity_ds1,
                    fused_kernel='tc::predict_cross_fused_kernel<4, false, false, true>', bound='valu')
        del interp

    # the reference's example table (docs/examples/bolplanck_wp.hdf5: G = 60, 19 r values)
    if wanted('wp'):
        from tabcorr_amd import TabCorr
        tab_wp = TabCorr.read(os.path.join(golden, 'bolplanck_wp.hdf5'))
        table_wp = {'gal_type': tab_wp.gal_type.as_array(), 'tpcf_matrix': tab_wp.tpcf_matrix,
                    'tpcf_shape': tab_wp.tpcf_shape, 'attrs': tab_wp.attrs}
        h_wp = tab_wp.to_device().handle
        d_theta_wp = dev.upload(theta)
        d_ngal_wp, d_xi_wp = dev.malloc(10000), dev.malloc(10000 * N_R)
        ends_wp = np.r_[0:2, 9998:10000]
        cache_wp = {}

        def parity_wp():
            expect = oracle.predict_zheng07_batch(table_wp, theta[ends_wp])
            return max(rel(dev.download(d_ngal_wp, 10000)[ends_wp], expect[0]),
                       rel(dev.download(d_xi_wp, 10000 * N_R).reshape(10000, N_R)[ends_wp],
                           expect[1]))
        measure('reference example: bolplanck wp table', 'wp',
                "the reference's docs/examples/bolplanck_wp.hdf5 (BASELINE configs[0]'s table): "
                '30 mass bins x {cen,sat} (G=60, P=1830), 19 rp bins, 10^4 draws', h_wp,
                lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                    h_wp, d_theta_wp, 5, 10000, N_GAUSS, 0, d_ngal_wp, d_xi_wp)),
                lambda: _lib.check(lib.tc_table_synchronize(h_wp)),
                lambda: tab_wp.predict_batch(theta), 10000, 10000 * pair_flops(60, N_R),
                FP64_PEAK_TFLOPS, 'tc::contract_quad_kernel<5, false>', 'f64',
                cpu_rate(lambda i: oracle.predict_zheng07(table_wp, theta[i % 10000],
                                                          cache=cache_wp)), parity_wp,
                fused_kernel='tc::predict_fused_kernel<10, 5, false, false, false, 8, 64, false, 1>')
        del tab_wp

    # the layout of the reference's database (scripts/tabulate_snapshot.py:179-193: 30 mass bins x
    # 2 percentile bins; tabcorr/database.py:56-59: grids of up to 4 x 4 x 4 tables)
    if wanted('db'):
        tables_db, keys_db, points_db = synthetic.synthetic_interpolator(
            (4, 4, 4), 30, 2, (N_R, ), 'auto', seed=11)
        interp_db = Interpolator([make(t) for t in tables_db],
                                 {k: points_db[:, d] for d, k in enumerate(keys_db)})
        n_db = 10000
        rng = np.random.default_rng(12)
        x_db = np.ascontiguousarray(np.stack(
            [rng.uniform(xp[0], xp[-1], size=n_db) for xp in interp_db.xp], axis=-1))
        idev_db = interp_db.to_device()
        d_theta_db, d_x_db = dev.upload(theta), dev.upload(x_db)
        d_ngal_db, d_xi_db = dev.malloc(n_db), dev.malloc(n_db * N_R)
        setup_db = oracle.interpolator_setup(tables_db, points_db)
        ends_db = np.r_[0:1, n_db - 1:n_db]

        def parity_db():
            expect = oracle.interpolator_predict_zheng07_batch(tables_db, setup_db,
                                                               theta[ends_db], x_db[ends_db])
            return max(rel(dev.download(d_ngal_db, n_db)[ends_db], expect[0]),
                       rel(dev.download(d_xi_db, n_db * N_R).reshape(n_db, N_R)[ends_db],
                           expect[1], floor=1e-12))
        measure('reference database layout', 'db',
                'Interpolator over a 4 x 4 x 4 grid (tabcorr/database.py:56-59) of synthetic auto '
                'tables with 30 mass x 2 percentile bins x {cen,sat} (G=120, P=7260; '
                'scripts/tabulate_snapshot.py:179-193), 19 rp bins, 10^4 draws',
                idev_db.tables[0].handle,
                lambda: _lib.check(lib.tc_interp_predict_zheng07_batch_device(
                    idev_db.handle, d_theta_db, 5, d_x_db, n_db, N_GAUSS, 0, d_ngal_db,
                    d_xi_db)),
                lambda: _lib.check(lib.tc_interp_synchronize(idev_db.handle)),
                lambda: interp_db.predict_batch(theta, x_db), n_db,
                n_db * 64 * pair_flops(120, N_R), FP64_PEAK_TFLOPS,
                'tc::contract_quad_kernel<5, true>', 'f64',
                cpu_rate(lambda i: oracle.interpolator_predict(
                    tables_db, setup_db, oracle.Zheng07(theta[i % n_db]), x_db[i % n_db])),
                parity_db)
        del interp_db, idev_db

    # configs[4]: AbacusSummit-scale table, rp_pi (19 x 40), float32 MFMA variant and float64
    if wanted('cfg5f32') or wanted('cfg5f64'):
        table5 = synthetic.synthetic_table(100, 1, (19, 40), 'auto', seed=9)
        n_r5 = 760
        d_ngal5, d_xi5 = dev.malloc(10000), dev.malloc(10000 * n_r5)
        d_theta5 = dev.upload(theta)
        cache5 = {}
        cpu5 = cpu_rate(lambda i: oracle.predict_zheng07(table5, theta[i % 10000], cache=cache5))
        ends5 = np.r_[0, 9999]
        expect5 = oracle.predict_zheng07_batch(table5, theta[ends5])

        def parity5():
            xi = dev.download(d_xi5, 10000 * n_r5).reshape((10000, ) + expect5[1].shape[1:])
            return max(rel(dev.download(d_ngal5, 10000)[ends5], expect5[0]),
                       rel(xi[ends5], expect5[1]))
        for dtype, tag, peak, kernel in (
                ('float32', 'cfg5f32', FP32_PEAK_TFLOPS, 'tc::contract_quad_f32_kernel<4, false>'),
                ('float64', 'cfg5f64', FP64_PEAK_TFLOPS, 'tc::contract_quad_kernel<5, false>')):
            if not wanted(tag):
                continue
            tab5 = make(table5, compute_dtype=dtype)
            h5 = tab5.to_device().handle
            kept5 = np.zeros(10000), np.zeros((10000, n_r5))
            measure('configs[4] ' + dtype, tag, 'rp_pi table 19 x 40 (R=760), 100 x {cen,sat} '
                    'bins (G=200, P=20100), 10^4 draws, %s table and contraction' % dtype, h5,
                    lambda: _lib.check(lib.tc_predict_zheng07_batch_device(
                        h5, d_theta5, 5, 10000, N_GAUSS, 0, d_ngal5, d_xi5)),
                    lambda: _lib.check(lib.tc_table_synchronize(h5)),
                    lambda: tab5.predict_batch(theta), 10000, 10000 * pair_flops(200, n_r5),
                    peak, kernel, 'f32' if dtype == 'float32' else 'f64', cpu5, parity5,
                    host_call_kept=lambda: tab5.predict_batch(theta, out=kept5))
            del tab5
    dev.free_all()
    return out


def cpu_baseline(table, seconds):
    """The NumPy port of the reference's predict() (oracle), one call per draw as
    in the reference's usage (README.md:72-75), on one host core."""
    from oracle import tabcorr_oracle as oracle
    from tabcorr_amd import synthetic
    theta = synthetic.zheng07_draws(200000, seed=99)
    cache = {}
    oracle.predict_zheng07(table, theta[0], cache=cache)       # builds the caches
    start = time.perf_counter()
    count = 0
    while True:
        for t in theta[count:count + 200]:
            oracle.predict_zheng07(table, t, cache=cache)
        count += 200
        spent = time.perf_counter() - start
        if spent >= seconds or count >= len(theta):
            break
    return {'value': count / spent, 'unit': 'calls/s', 'cores': 1, 'kind': 'port',
            'sample': '%d sequential predict() calls of the same workload (same '
                      'table, draws from the same prior) in %.1f s' % (count, spent)}


def _cpu_walker(job):
    """One independent MCMC-style walker: sequential predict() calls for `seconds`."""
    seed, seconds = job
    from oracle import tabcorr_oracle as oracle
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(N_PRIM, N_SEC, (N_R, ), 'auto', seed=0)
    theta = synthetic.zheng07_draws(100000, seed=seed)
    cache = {}
    oracle.predict_zheng07(table, theta[0], cache=cache)
    start = time.perf_counter()
    count = 0
    while time.perf_counter() - start < seconds and count < len(theta):
        for t in theta[count:count + 100]:
            oracle.predict_zheng07(table, t, cache=cache)
        count += 100
    return count, time.perf_counter() - start


def cpu_baseline_all_cores(seconds):
    """predict() is single-threaded in the reference, so "all host
    cores" means one independent walker process per core."""
    import multiprocessing
    cores = min(len(os.sched_getaffinity(0)), 256)
    with multiprocessing.get_context('fork').Pool(cores) as pool:
        done = pool.map(_cpu_walker, [(1000 + i, seconds) for i in range(cores)])
    return {'value': sum(count / spent for count, spent in done), 'unit': 'calls/s',
            'cores': cores, 'kind': 'port',
            'sample': '%d walker processes x %.1f s of sequential predict() calls'
                      % (cores, seconds)}

