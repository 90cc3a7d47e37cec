"""CPU-only tests: the C ABI library loads and exports every declared symbol,
host-side mathematics of the library (quadrature nodes, packed index map,
spline matrices, work planning), the host classes' argument handling, and the
multi-process sharding over gloo (world size 2)."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from util import REPO, load_golden, table_from_golden

sys.path.insert(0, REPO)


def declared_symbols():
    """Functions declared in include/*.h: the boundary (tabcorr_amd.h) and the test hooks
    (tabcorr_amd_testing.h)."""
    import glob
    names = set()
    for path in glob.glob(os.path.join(REPO, 'include', '*.h')):
        header = re.sub(r'/\*.*?\*/', '', open(path).read(), flags=re.S)
        names.update(re.findall(r'\b(tc_[a-z0-9_]+)\s*\(', header))
    return sorted(names)


@pytest.fixture(scope='module')
def lib():
    from tabcorr_amd import build, _lib
    build.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from tabcorr_amd import _lib
    names = declared_symbols()
    assert len(names) >= 60
    for name in names:
        assert hasattr(lib, name), name
    # the boundary header declares no debug hooks
    boundary = open(os.path.join(REPO, 'include', 'tabcorr_amd.h')).read()
    assert 'tc_debug' not in boundary and 'tc_plan_debug' not in boundary
    # and the ctypes table covers them all
    assert set(names) == set(_lib.SIGNATURES) | {'tc_last_error'}


def test_library_is_bound_to_system_rocm_even_next_to_torch(lib):
    from tabcorr_amd import _lib
    before = _lib.runtime_version()
    import torch  # noqa: F401  (bundles its own libamdhip64)
    assert _lib.runtime_version() == before
    assert before >= 70200000          # ROCm 7.2 runtime, not torch's 7.0


def test_no_gpu_fails_loudly(lib):
    from tabcorr_amd import TabCorr, _lib
    if _lib.device_count() > 0:
        pytest.skip('a GPU is present')
    data = load_golden('synthetic_small_auto')
    table = table_from_golden(data)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                  table['tpcf_shape'], table['attrs'])
    with pytest.raises(_lib.TabCorrHipError):
        halotab.predict_batch(data['theta'])
    # the round-3 entry points have no fallback either
    from tabcorr_amd import corrfunc, pinned_empty, is_pinned
    with pytest.raises(_lib.TabCorrHipError):
        halotab.predict_batch_async(data['theta'])
    with pytest.raises(_lib.TabCorrHipError):
        pinned_empty(8)
    assert not is_pinned(np.zeros(4))            # (the registry needs no device)
    points = np.random.default_rng(0).uniform(0, 50, (20, 3))
    with pytest.raises(_lib.TabCorrHipError):
        corrfunc.mean_delta_sigma(points, points, 1.0, np.array([1.0, 2.0]), period=50.0)
    with pytest.raises(_lib.TabCorrHipError):
        corrfunc.wp(points, np.array([1.0, 2.0]), 10.0, period=50.0)


def test_result_of_async_validation_without_a_device():
    """Argument checks of the asynchronous Python layer that do not need a device."""
    from tabcorr_amd import pinned
    with pytest.raises(ValueError, match='out must hold 2 arrays'):
        pinned.stage_outputs([(3, ), (3, 4)], [np.zeros(3)])
    with pytest.raises(ValueError, match='page-locked'):
        pinned.stage_outputs([(3, )], [np.zeros(3)])
    with pytest.raises(ValueError, match='C-contiguous'):
        pinned.stage_outputs([(3, )], [np.zeros((3, 2))[:, 0]])
    with pytest.raises(ValueError, match='C-contiguous numpy array'):
        pinned.pin([1.0, 2.0])


def test_gauss_legendre(lib):
    from tabcorr_amd import _lib
    for n in [1, 2, 3, 7, 10, 33, 100]:
        x = np.zeros(n)
        w = np.zeros(n)
        assert lib.tc_gauss_legendre(n, _lib.as_double_p(x),
                                     _lib.as_double_p(w)) == 0
        x_ref, w_ref = np.polynomial.legendre.leggauss(n)
        np.testing.assert_allclose(x, (x_ref + 1) / 2, rtol=0, atol=3e-16)
        np.testing.assert_allclose(w, w_ref, rtol=0, atol=2e-14)


def test_pair_indices_match_reference_layout(lib):
    from tabcorr_amd import _lib
    golden = load_golden('helpers')
    for n in range(1, 8):
        p = n * (n + 1) // 2
        i1 = np.zeros(p, np.int32)
        i2 = np.zeros(p, np.int32)
        pre = np.zeros(p, np.int32)
        assert lib.tc_pair_indices(
            n, i1.ctypes.data_as(_lib.c_int32_p),
            i2.ctypes.data_as(_lib.c_int32_p),
            pre.ctypes.data_as(_lib.c_int32_p)) == 0
        assert np.array_equal(i1 * n + i2, golden['sym_index_%d' % n])
        assert np.array_equal(pre, np.where(i1 == i2, 1, 2))


def test_symmetric_matrix_to_array():
    from tabcorr_amd import symmetric_matrix_to_array
    golden = load_golden('helpers')
    for n in range(1, 8):
        index = np.arange(n * n).reshape(n, n)
        assert np.array_equal(
            symmetric_matrix_to_array(index, check_symmetry=False),
            golden['sym_index_%d' % n])
    with pytest.raises(ValueError):
        symmetric_matrix_to_array(np.arange(9).reshape(3, 3))


def test_spline_matrix(lib):
    from tabcorr_amd import _lib
    golden = load_golden('helpers')
    for n in [4, 5, 7, 12]:
        xp = np.ascontiguousarray(golden['spline_xp_%d' % n])
        a = np.zeros((n - 1, 4, n))
        assert lib.tc_spline_interpolation_matrix(
            n, _lib.as_double_p(xp), _lib.as_double_p(a)) == 0
        # monomial coefficients are ill-conditioned; compare spline VALUES
        yp = golden['spline_yp_%d' % n]
        for x, y in zip(golden['spline_x_%d' % n], golden['spline_y_%d' % n]):
            seg = min(max(np.digitize(x, xp) - 1, 0), n - 2)
            value = np.einsum('ij,j...,i', a[seg], yp, x**np.arange(4))
            np.testing.assert_allclose(value, y, rtol=1e-9, atol=1e-11)
    xp = np.arange(3.0)
    a = np.zeros((2, 4, 3))
    assert lib.tc_spline_interpolation_matrix(
        3, _lib.as_double_p(xp), _lib.as_double_p(a)) == _lib.TC_ERR_INVALID
    assert b'less than 4' in lib.tc_last_error()


@pytest.mark.parametrize('mode', [0, 1])
def test_plan_covers_every_column_once(lib, mode):
    from tabcorr_amd import _lib
    rng = np.random.default_rng(mode)
    cases = [(1, 'half'), (2, 'half'), (7, 'half'), (12, 'half'), (9, 'rand'),
             (5, 'allcen'), (6, 'allsat'), (100, 'half'), (61, 'rand'), (200, 'half'),
             (333, 'rand')]
    for n_bins, pattern in cases:
        if pattern == 'half':
            central = np.arange(n_bins) < n_bins // 2
        elif pattern == 'rand':
            central = rng.integers(0, 2, n_bins).astype(bool)
        else:
            central = np.full(n_bins, pattern == 'allcen')
        central = central.astype(np.uint8)
        n_pairs = n_bins * (n_bins + 1) // 2 if mode == 0 else n_bins
        for n_chunks in [1, 3, 8, 32, 1000]:
            n = ctypes.c_int64()
            pair = np.zeros(n_pairs, np.int32)
            chunk = np.zeros(n_pairs, np.int32)
            comp = np.zeros(n_pairs, np.int32)
            status = lib.tc_plan_debug(
                mode, n_bins, central.ctypes.data_as(_lib.c_uint8_p), n_chunks,
                ctypes.byref(n), pair.ctypes.data_as(_lib.c_int32_p),
                chunk.ctypes.data_as(_lib.c_int32_p),
                comp.ctypes.data_as(_lib.c_int32_p))
            assert status == 0, lib.tc_last_error()
            assert n.value == n_pairs
            assert sorted(pair.tolist()) == list(range(n_pairs))
            # component of every column: number of satellite members
            if mode == 0:
                rows, cols = np.tril_indices(n_bins)
                expect = (1 - central[rows]) + (1 - central[cols])
            else:
                expect = 1 - central
            assert np.array_equal(comp[np.argsort(pair)], expect)
            # chunks are contiguous runs in processing order
            assert np.all(np.diff(chunk) >= 0)


def test_host_class_argument_errors():
    from tabcorr_amd import TabCorr, Zheng07Model, GalTypeTable
    data = load_golden('synthetic_small_auto')
    table = table_from_golden(data)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                  table['tpcf_shape'], table['attrs'])
    assert isinstance(halotab.gal_type, GalTypeTable)
    assert halotab.gal_type['gal_type'][0] == 'centrals'
    assert len(halotab.gal_type) == 14
    # consistency checks run before anything touches the device
    with pytest.raises(ValueError, match='redshift'):
        halotab.predict(Zheng07Model(redshift=1.0))
    with pytest.raises(ValueError, match='primary halo'):
        halotab.predict(Zheng07Model(prim_haloprop_key='halo_m200b'))
    with pytest.raises(ValueError, match='secondary halo'):
        halotab.predict(Zheng07Model(sec_haloprop_key='halo_spin'))
    model = Zheng07Model()
    model.gal_types = ['centrals', 'satellites', 'orphans']
    with pytest.raises(ValueError, match='galaxy types'):
        halotab.predict(model)
    with pytest.raises(NotImplementedError):
        TabCorr.tabulate(None, None)


def test_npz_round_trip(tmp_path):
    from tabcorr_amd import TabCorr
    data = load_golden('synthetic_small_cross')
    table = table_from_golden(data)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                  table['tpcf_shape'], table['attrs'],
                                  tpcf_args=(np.arange(3.0), ),
                                  tpcf_kwargs={'pi_max': 40})
    fname = str(tmp_path / 'table.npz')
    halotab.write(fname)
    with pytest.raises(OSError):
        halotab.write(fname)
    halotab.write(fname, overwrite=True)
    back = TabCorr.read(fname)
    assert back.attrs == halotab.attrs
    assert back.tpcf_shape == halotab.tpcf_shape
    assert np.array_equal(back.tpcf_matrix, halotab.tpcf_matrix)
    assert back.gal_type.colnames == halotab.gal_type.colnames
    for name in back.gal_type.colnames:
        assert np.array_equal(back.gal_type[name], halotab.gal_type[name])
    assert np.array_equal(back.tpcf_args[0], np.arange(3.0))
    assert back.tpcf_kwargs['pi_max'] == 40


def test_round_robin_assembly():
    from tabcorr_amd import parallel
    rng = np.random.default_rng(0)
    for n_draws in [1, 5, 16, 17]:
        for world in [1, 2, 3, 8]:
            theta = rng.normal(size=(n_draws, 5))
            parts = [parallel.local_shard(theta, r, world)
                     for r in range(world)]
            assert all(len(p) == parallel.shard_size(n_draws, world)
                       for p in parts)
            assert np.array_equal(parallel.assemble(parts, n_draws), theta)


WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
from util import load_golden, table_from_golden
from oracle import tabcorr_oracle as oracle
from tabcorr_amd import parallel

class OracleTab:
    """Stands in for the device on a CPU-only box: the test is about the
    sharding, the gather and the reassembly, not about the kernel."""
    def __init__(self, table):
        self.table = table
        self.tpcf_shape = table['tpcf_shape']
    def predict_batch(self, theta, **kwargs):
        return oracle.predict_zheng07_batch(self.table, theta, **kwargs)

comm = parallel.Communicator.from_env(use_rccl=False)
assert comm.world_size == 2 and comm.gather_backend == 'gloo'
data = load_golden('synthetic_rp_pi')
table = table_from_golden(data)
theta = data['theta'][:15]          # odd: the last rank gets a padded shard
for separate in [False, True]:
    out = parallel.predict_batch_sharded(OracleTab(table), theta, comm,
                                         separate_gal_type=separate)
    if comm.is_root:
        expect = oracle.predict_zheng07_batch(table, theta,
                                              separate_gal_type=separate)
        if separate:
            for key in expect[1]:
                assert out[1][key].shape == (15, 5, 8)
                assert np.array_equal(out[1][key], expect[1][key]), key
            for key in expect[0]:
                assert np.array_equal(out[0][key], expect[0][key])
        else:
            assert np.array_equal(out[0], expect[0])
            assert np.array_equal(out[1], expect[1])
    else:
        assert out is None
# Interpolator-style objects: extra parameters are sharded alongside the draws
from util import interpolator_tables_from_golden
idata = load_golden('interp_2d_auto')
itables = interpolator_tables_from_golden(idata)
setup = oracle.interpolator_setup(itables, idata['points'])

class OracleInterp:
    tabcorr_list = [OracleTab(itables[0])]
    def predict_batch(self, theta, x, **kwargs):
        return oracle.interpolator_predict_zheng07_batch(itables, setup, theta, x, **kwargs)

out = parallel.predict_batch_sharded(OracleInterp(), idata['theta'][:9], comm, x=idata['x'][:9])
if comm.is_root:
    assert np.array_equal(out[0], idata['ngal'][:9]) or np.allclose(out[0], idata['ngal'][:9], rtol=1e-11, atol=0)
    assert np.allclose(out[1], idata['xi'][:9], rtol=1e-10, atol=0)
out = parallel.predict_batch_sharded(OracleInterp(), idata['theta'][:9], comm, x=idata['x'][:9],
                                     separate_gal_type=True)
if comm.is_root:
    for key in ('centrals-centrals', 'centrals-satellites', 'satellites-satellites'):
        assert out[1][key].shape == idata['xi'][:9].shape
        assert np.allclose(out[1][key], idata['xi_sep_' + key][:9], rtol=1e-10,
                           atol=1e-14 * np.max(np.abs(idata['xi'])))
    for key in ('centrals', 'satellites'):
        assert np.allclose(out[0][key], idata['ngal_sep_' + key][:9], rtol=1e-11, atol=0)
else:
    assert out is None
# the fused likelihood: 16 bytes per draw gathered
class OracleChi2(OracleTab):
    def chi2_batch(self, theta, data, precision, **kwargs):
        kwargs.pop('family')
        assert kwargs.pop('assembias', False) is False
        ngal, xi = oracle.predict_zheng07_batch(self.table, theta, **kwargs)
        delta = xi.reshape(len(theta), -1) - data
        return ngal, np.einsum('bi,ij,bj->b', delta, precision, delta)
observed = np.full(40, 3.0)
weight = np.eye(40) * 0.5
out = parallel.chi2_batch_sharded(OracleChi2(table), theta, observed, weight, comm)
if comm.is_root:
    expect = OracleChi2(table).chi2_batch(theta, observed, weight, family='zheng07')
    assert out[0].shape == (15, ) and np.array_equal(out[0], expect[0])
    assert np.array_equal(out[1], expect[1])
else:
    assert out is None
assert comm.max(comm.rank + 1.0) == 2.0
assert comm.sum(1.0) == 2.0
comm.barrier()
print('rank', comm.rank, 'ok')
'''


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def test_sharded_predict_world_size_2_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'repo': REPO})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), str(script)],
        env=env, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stdout + result.stderr
    assert result.stdout.count('ok') == 2


RING_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r)
sys.path.insert(0, os.path.join(%(repo)r, 'tests'))
from util import load_golden, table_from_golden
from oracle import tabcorr_oracle as oracle
from tabcorr_amd import parallel

comm = parallel.Communicator.from_env(use_rccl=False)
world, rank = comm.world_size, comm.rank
assert world == int(os.environ['EXPECT_WORLD'])

# ---- the result ring of bench.py, drained over a fake data plane (gloo on host arrays) ----
n_out = 6
for every, n_blocks, steps in ((4, 4, 37), (3, 4, 12), (5, 2, 23), (1, 4, 9), (7, 4, 5)):
    ring_buffer = np.full(n_blocks * every * n_out, np.nan)
    recv = np.full(n_blocks * world * every * n_out, np.nan) if comm.is_root else None
    seen = {}
    released = []

    def gather(block, send_offset, recv_offset, count):
        parts = comm.gather_host(ring_buffer[send_offset:send_offset + count].copy())
        if comm.is_root:
            # rank-major, as ncclGather delivers it
            recv[recv_offset:recv_offset + world * count] = np.concatenate(parts)

    ring = parallel.ResultRing(n_out, every, world, gather, released.append, n_blocks)
    assert ring.ring_elements == ring_buffer.size
    assert not comm.is_root or ring.recv_elements == recv.size
    consumed = 0
    for index in range(steps):
        ring.before_step(index)
        offset = ring.slot_offset(index)
        # the "prediction" of step `index` on this rank: a signature per element
        ring_buffer[offset:offset + n_out] = 1000.0 * rank + index + 0.01 * np.arange(n_out)
        ring.after_step(index)
        # the root consumes every gather as soon as it is logged
        while consumed < len(ring.log):
            if comm.is_root:
                for r in range(world):
                    for step, where in ring.steps_in(ring.log[consumed], r):
                        assert (r, step) not in seen, (r, step)
                        seen[(r, step)] = recv[where:where + n_out].copy()
            consumed += 1
    ring.flush(steps)
    while consumed < len(ring.log):
        if comm.is_root:
            for r in range(world):
                for step, where in ring.steps_in(ring.log[consumed], r):
                    assert (r, step) not in seen, (r, step)
                    seen[(r, step)] = recv[where:where + n_out].copy()
        consumed += 1
    if comm.is_root:
        # every rank's every step exactly once, with that rank's and step's values
        assert sorted(seen) == [(r, k) for r in range(world) for k in range(steps)], (every, steps)
        for (r, k), values in seen.items():
            assert np.array_equal(values, 1000.0 * r + k + 0.01 * np.arange(n_out)), (r, k)
    # a block is released exactly when it is about to be rewritten after a gather
    expect_released = [(k %% (n_blocks * every)) // every for k in range(steps)
                       if k %% every == 0 and k >= n_blocks * every]
    assert released == expect_released, (released, expect_released)
    # the gathers are collectives: every rank logged the same sequence
    assert comm.max(len(ring.log)) == len(ring.log)

# ---- pack / gather / unpack arithmetic of the RCCL route, over the same fake --------------
data = load_golden('synthetic_rp_pi')
table = table_from_golden(data)
n_r = int(np.prod(table['tpcf_shape']))
for n_draws in (1, 7, 15, 16):
    theta = data['theta'][:n_draws]
    shard = parallel.local_shard(theta, rank, world)
    n_local = len(shard)
    for separate in (False, True):
        n_ngal, n_comp = (2, 3) if separate else (1, 1)
        ngal, xi = oracle.predict_zheng07_batch(table, shard, separate_gal_type=separate)
        if separate:
            ngal = np.stack([ngal['centrals'], ngal['satellites']], axis=1)
            xi = np.stack([xi[k].reshape(n_local, n_r) for k in
                           ('centrals-centrals', 'centrals-satellites',
                            'satellites-satellites')], axis=1)
        # what the device leaves in d_out: [ngal (n_local, n_ngal) | xi (n_local, n_comp, n_r)]
        packed = np.concatenate([np.ravel(ngal), np.ravel(xi)])
        assert packed.size == parallel.packed_count(n_local, n_ngal, n_comp, n_r)
        parts = comm.gather_host(packed)
        if comm.is_root:
            got_ngal, got_xi = parallel.unpack_gathered(
                np.concatenate(parts), world, n_local, n_ngal, n_comp, n_r, n_draws)
            e_ngal, e_xi = oracle.predict_zheng07_batch(table, theta, separate_gal_type=separate)
            if separate:
                assert np.array_equal(got_ngal[:, 1], e_ngal['satellites'])
                assert np.array_equal(got_xi[:, 1].reshape(e_xi['centrals-satellites'].shape),
                                      e_xi['centrals-satellites'])
            else:
                assert np.array_equal(got_ngal[:, 0], e_ngal)
                assert np.array_equal(got_xi[:, 0].reshape(e_xi.shape), e_xi)
comm.barrier()
print('rank', rank, 'ok')
'''


@pytest.mark.parametrize('world_size', [2, 3])
def test_result_ring_and_rccl_route_arithmetic_over_gloo(tmp_path, world_size):
    """bench.py's result ring (slots, blocks, releases, the partial flush, receive offsets) and
    the pack / unpack arithmetic of the RCCL route of predict_batch_sharded, driven with
    `steps` not divisible by `every` over a fake data plane: rank 0 must hold every rank's
    every step exactly once."""
    script = tmp_path / 'ring_worker.py'
    script.write_text(RING_WORKER % {'repo': REPO})
    port = _free_port()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', EXPECT_WORLD=str(world_size))
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', str(world_size), '--master-addr', '127.0.0.1',
         '--master-port', str(port), str(script)],
        env=env, capture_output=True, text=True, timeout=600)
    assert result.returncode == 0, result.stdout + result.stderr
    assert result.stdout.count('ok') == world_size


def test_result_ring_single_rank_bookkeeping():
    from tabcorr_amd.parallel import ResultRing
    calls = []
    ring = ResultRing(10, 3, 2, lambda *a: calls.append(a), n_blocks=4)
    for index in range(8):
        ring.before_step(index)
        ring.after_step(index)
    ring.flush(8)
    # blocks 0 and 1 complete, block 2 holds steps 6 and 7
    assert calls == [(0, 0, 0, 30), (1, 30, 60, 30), (2, 60, 120, 20)]
    assert [entry[:3] for entry in ring.log] == [(0, 0, 3), (1, 3, 3), (2, 6, 2)]
    assert ring.steps_in(ring.log[2], 1) == [(6, 120 + 20), (7, 120 + 30)]
    with pytest.raises(ValueError):
        ResultRing(10, 0, 2, None)


# -- HDF5 through the C library (no h5py / astropy) -------------------------------------

def _need_hdf5():
    from tabcorr_amd import hdf5
    if not hdf5.available():
        pytest.skip('libhdf5 not found')


@pytest.mark.parametrize('name', ['bolplanck_wp', 'bolplanck_ds'])
def test_read_reference_hdf5_files(name):
    """The reference's own data files (kept as fixtures) parse to exactly the arrays
    the reference's reader produced (npz fixtures)."""
    _need_hdf5()
    from tabcorr_amd import TabCorr
    from util import GOLDEN
    halotab = TabCorr.read(os.path.join(GOLDEN, name + '.hdf5'))
    data = load_golden(name)
    table = table_from_golden(data)
    assert np.array_equal(halotab.tpcf_matrix, table['tpcf_matrix'])
    assert halotab.tpcf_matrix.dtype == np.float64     # tabcorr.py:399
    assert halotab.tpcf_shape == table['tpcf_shape']
    for key, value in table['attrs'].items():
        assert halotab.attrs[key] == value, key
    assert halotab.gal_type.colnames == list(table['gal_type'].dtype.names)
    for column in halotab.gal_type.colnames:
        assert np.array_equal(halotab.gal_type.as_array()[column],
                              table['gal_type'][column]), column
    assert halotab.gal_type['gal_type'][-1] == 'satellites'
    if name == 'bolplanck_wp':
        assert halotab.tpcf_kwargs['pi_max'] == 40
        assert halotab.tpcf_args[0].shape == (20, )


def test_read_reference_interpolator_file_and_round_trip(tmp_path):
    _need_hdf5()
    from tabcorr_amd import Interpolator, TabCorr
    from util import GOLDEN
    interp = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5'))
    data = load_golden('ds_efficient')
    assert interp.keys == ['log_eta']
    assert np.array_equal(interp.points, data['points'])
    for i, halotab in enumerate(interp.tabcorr_list):
        table = table_from_golden(data, 'table%d_' % i)
        assert np.array_equal(halotab.tpcf_matrix, table['tpcf_matrix'])
        assert halotab.attrs['simname'] == 'base_c000_ph000'
    for d in range(1):
        np.testing.assert_allclose(interp.xp[d], data['xp%d' % d], rtol=0)

    fname = str(tmp_path / 'interp.hdf5')
    interp.write(fname)
    with pytest.raises(OSError):
        interp.write(fname)                       # 'w-' semantics
    interp.write(fname, overwrite=True)
    back = Interpolator.read(fname)
    assert np.array_equal(back.points, interp.points)
    for a, b in zip(back.tabcorr_list, interp.tabcorr_list):
        assert np.array_equal(a.tpcf_matrix, b.tpcf_matrix)
        assert a.attrs == b.attrs
        assert a.tpcf_shape == b.tpcf_shape
        assert [x.shape for x in a.tpcf_args] == [x.shape for x in b.tpcf_args]
        for column in a.gal_type.colnames:
            assert np.array_equal(a.gal_type.as_array()[column],
                                  b.gal_type.as_array()[column])

    single = str(tmp_path / 'single.hdf5')
    halotab = interp.tabcorr_list[2]
    halotab.write(single, matrix_dtype=np.float64)
    again = TabCorr.read(single)
    assert np.array_equal(again.tpcf_matrix, halotab.tpcf_matrix)
    npz = str(tmp_path / 'interp.npz')
    interp.write(npz)
    assert np.array_equal(Interpolator.read(npz).points, interp.points)


def test_fastmath_accuracy_on_host():
    """The table-driven erf / log2 / exp2 / exp10 of the occupation kernel (same inline
    code on the host) against scipy / numpy."""
    import ctypes
    from scipy import special
    from tabcorr_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)

    def run(kind, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        _lib.check(lib.tc_debug_fastmath(
            kind, x.size, x.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            y.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return y

    x = np.concatenate([rng.uniform(-7, 7, 200000), np.arange(0, 769) / 128.0,
                        (np.arange(0, 768) + 0.5) / 128.0, [0.0, -0.0, 1e-300, 50.0, -50.0]])
    assert np.max(np.abs(run(0, x) - special.erf(x))) < 4e-16
    y = np.concatenate([10**rng.uniform(-290, 290, 100000), rng.uniform(0.5, 2.0, 100000),
                        1.0 + np.arange(257) / 256.0, [1.0, 2.0, 0.5, 1e-300]])
    expect = np.log2(y)
    assert np.max(np.abs(run(1, y) - expect) / np.maximum(1.0, np.abs(expect))) < 4e-16
    z = np.concatenate([rng.uniform(-900, 900, 200000), rng.uniform(-1, 1, 50000),
                        np.arange(-512, 512) / 256.0])
    assert np.max(np.abs(run(2, z) / np.exp2(z) - 1.0)) < 4e-16
    assert run(2, np.array([-5000.0]))[0] < 1e-300
    w = rng.uniform(9.0, 16.0, 100000)
    assert np.max(np.abs(run(3, w) / 10.0**w - 1.0)) < 6e-16


# ---- model recognition (tabcorr.py:556-563 call sites; halotools itself is not installed) --

def _fake_halotools(name, **attributes):
    """An object whose class carries a halotools class name and module, as the
    components of a HodModelFactory do."""
    cls = type(name, (), {'__module__': 'halotools.empirical_models.occupation_models'})
    component = cls()
    component.prim_haloprop_key = 'halo_mvir'
    for key, value in attributes.items():
        setattr(component, key, value)
    return component


class _FakeFactory:
    def __init__(self, cens, sats, **param_dict):
        self.gal_types = ['centrals', 'satellites']
        self.redshift = 0.0
        self._input_model_dictionary = {'centrals_occupation': cens,
                                        'satellites_occupation': sats}
        self.param_dict = {'logMmin': 12.1, 'sigma_logM': 0.3, 'logM0': 11.5,
                           'logM1': 13.2, 'alpha': 1.1}
        self.param_dict.update(param_dict)


def _assembias_pair(**overrides):
    attributes = {'_assembias_strength_abscissa': [2.0], '_split_ordinates': [0.5],
                  '_split_abscissa': [2.0], 'sec_haloprop_key': 'halo_nfw_conc'}
    attributes.update(overrides)
    return (_fake_halotools('AssembiasZheng07Cens', **attributes),
            _fake_halotools('AssembiasZheng07Sats', **attributes))


def test_device_spec_accepts_only_what_the_kernel_implements():
    from tabcorr_amd import Zheng07Model
    from tabcorr_amd.models import device_spec, ASSEMBIAS_KEYS
    strengths = {ASSEMBIAS_KEYS[0]: 0.4, ASSEMBIAS_KEYS[1]: -0.7}

    # plain zheng07 composite, extra (interpolator) keys in param_dict are fine
    spec = device_spec(_FakeFactory(_fake_halotools('Zheng07Cens'),
                                    _fake_halotools('Zheng07Sats'), log_eta=0.1))
    assert spec is not None and not spec.assembias and not spec.modulate_with_cenocc
    np.testing.assert_array_equal(spec.theta, [12.1, 0.3, 11.5, 13.2, 1.1])
    # modulate_with_cenocc with Zheng07 centrals
    sats = _fake_halotools('Zheng07Sats', modulate_with_cenocc=True,
                           central_occupation_model=_fake_halotools('Zheng07Cens'))
    spec = device_spec(_FakeFactory(_fake_halotools('Zheng07Cens'), sats))
    assert spec is not None and spec.modulate_with_cenocc
    # decorated pair with one constant strength and the median split
    spec = device_spec(_FakeFactory(*_assembias_pair(), **strengths))
    assert spec is not None and spec.assembias
    np.testing.assert_array_equal(spec.theta[5:], [0.4, -0.7])
    # this package's own model
    assert device_spec(Zheng07Model()) is not None

    # ---- everything below must take the host route ----
    # mass-dependent strength: two abscissae and a _param2 key
    cens, sats = _assembias_pair(_assembias_strength_abscissa=[1e12, 1e14])
    assert device_spec(_FakeFactory(cens, sats, **strengths)) is None
    two = dict(strengths)
    two['mean_occupation_centrals_assembias_param2'] = 0.1
    assert device_spec(_FakeFactory(*_assembias_pair(), **two)) is None
    # a split away from the median, or split information missing altogether
    assert device_spec(_FakeFactory(*_assembias_pair(_split_ordinates=[0.3]),
                                    **strengths)) is None
    cens, sats = _assembias_pair()
    del cens._split_ordinates
    assert device_spec(_FakeFactory(cens, sats, **strengths)) is None
    # strengths missing from param_dict
    assert device_spec(_FakeFactory(*_assembias_pair())) is None
    # assembly-bias keys on an undecorated model
    assert device_spec(_FakeFactory(_fake_halotools('Zheng07Cens'),
                                    _fake_halotools('Zheng07Sats'), **strengths)) is None
    # a user class with halotools' name (may override anything), a subclass defined
    # elsewhere, another occupation family, mixed pairs
    user = type('Zheng07Cens', (), {'__module__': 'my_models'})()
    assert device_spec(_FakeFactory(user, _fake_halotools('Zheng07Sats'))) is None
    assert device_spec(_FakeFactory(_fake_halotools('Leauthaud11Cens'),
                                    _fake_halotools('Leauthaud11Sats'))) is None
    assert device_spec(_FakeFactory(_fake_halotools('Zheng07Cens'),
                                    _assembias_pair()[1], **strengths)) is None
    # instance-level override of the occupation function
    cens = _fake_halotools('Zheng07Cens', mean_occupation=lambda **kw: 1.0)
    assert device_spec(_FakeFactory(cens, _fake_halotools('Zheng07Sats'))) is None
    # satellites modulated by something that is not Zheng07 centrals
    sats = _fake_halotools('Zheng07Sats', modulate_with_cenocc=True,
                           central_occupation_model=_fake_halotools('Leauthaud11Cens'))
    assert device_spec(_FakeFactory(_fake_halotools('Zheng07Cens'), sats)) is None
    sats = _fake_halotools('Zheng07Sats', modulate_with_cenocc=True)
    assert device_spec(_FakeFactory(_fake_halotools('Zheng07Cens'), sats)) is None
    # subclasses of this package's model that change the occupation functions or the split
    class Steeper(Zheng07Model):
        def mean_occupation_satellites(self, **kwargs):
            return 2.0 * super().mean_occupation_satellites(**kwargs)
    assert device_spec(Steeper()) is None
    shifted = Zheng07Model(sec_haloprop_key='halo_nfw_conc')
    shifted.split = 0.3
    assert device_spec(shifted) is None
    # no components at all
    assert device_spec(object()) is None


def test_heaviside_strength_is_clipped_like_halotools():
    """halotools' HeavisideAssembias.assembias_strength clips to [-1, 1]; the oracle, the
    host model and (GPU tests) the kernel do the same."""
    from tabcorr_amd import Zheng07Model
    from tabcorr_amd.models import ASSEMBIAS_KEYS
    from oracle import tabcorr_oracle as oracle
    mass = np.logspace(11, 15, 50)
    percentile = np.tile([0.25, 0.75], 25)
    theta = [12.3, 0.4, 11.8, 13.1, 1.05]
    for strength in (1.7, -2.5, np.nan):
        clipped = np.clip(strength, -1, 1)
        model = oracle.Zheng07(theta, assembias=(strength, strength))
        expect = oracle.Zheng07(theta, assembias=(clipped, clipped))
        for name in ('mean_occupation_centrals', 'mean_occupation_satellites'):
            np.testing.assert_array_equal(getattr(model, name)(mass, percentile),
                                          getattr(expect, name)(mass, percentile))
        host = Zheng07Model(sec_haloprop_key='halo_nfw_conc', logMmin=theta[0],
                            sigma_logM=theta[1], logM0=theta[2], logM1=theta[3],
                            alpha=theta[4], **{ASSEMBIAS_KEYS[0]: strength,
                                               ASSEMBIAS_KEYS[1]: strength})
        np.testing.assert_allclose(
            host.mean_occupation_centrals(prim_haloprop=mass,
                                          sec_haloprop_percentile=percentile),
            expect.mean_occupation_centrals(mass, percentile), rtol=1e-14, equal_nan=True)
        np.testing.assert_allclose(
            host.mean_occupation_satellites(prim_haloprop=mass,
                                            sec_haloprop_percentile=percentile),
            expect.mean_occupation_satellites(mass, percentile), rtol=1e-13,
            equal_nan=True)
        occupation = expect.mean_occupation_centrals(mass, percentile)
        if not np.isnan(strength):
            assert np.all((occupation >= 0) & (occupation <= 1))


def test_interpolator_param_table_and_nan_range():
    from tabcorr_amd import Interpolator, TabCorr, synthetic
    tables, keys, points = synthetic.synthetic_interpolator((4, 4), 5, 1, (3, ), 'auto',
                                                            seed=3)
    tabs = [TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'], t['attrs'])
            for t in tables]
    interp = Interpolator(tabs, {k: points[:, d] for d, k in enumerate(keys)})
    table = interp.param_dict_table
    assert table.colnames == list(keys) + ['tabcorr_index']      # interpolator.py:59-61
    assert len(table) == 16
    assert np.all(np.diff(table[keys[0]]) >= 0)
    # NaN is outside of the interpolation range (np.digitize sorts it past the last node)
    x = np.array([[points[0, 0], np.nan]])
    with pytest.raises(ValueError, match='outside of the interpolation'):
        interp._check_range(x, False)
    interp._check_range(x, True)


def test_host_logic_under_address_and_undefined_sanitizers(tmp_path):
    """the library's host logic (planner, schedules, table layouts, spline
    and quadrature setup, the pair counter's cell sort) built with
    -fsanitize=address,undefined and run over a sweep of shapes (tools/sanitize)."""
    import shutil
    if shutil.which('g++') is None or shutil.which('make') is None:
        pytest.skip('g++ / make not available')
    work = tmp_path / 'sanitize'
    shutil.copytree(os.path.join(REPO, 'tools', 'sanitize'), work,
                    ignore=shutil.ignore_patterns('host_driver'))
    makefile = (work / 'Makefile').read_text().replace(
        'SRC = ../../tabcorr_amd/csrc', 'SRC = ' + os.path.join(REPO, 'tabcorr_amd', 'csrc'))
    (work / 'Makefile').write_text(makefile)
    source = (work / 'host_driver.cpp').read_text().replace(
        '../../tabcorr_amd/csrc', os.path.join(REPO, 'tabcorr_amd', 'csrc'))
    (work / 'host_driver.cpp').write_text(source)
    build = subprocess.run(['make', '-C', str(work)], capture_output=True, text=True,
                           timeout=600)
    if build.returncode != 0 and 'sanitize' in build.stderr and 'cannot find' in build.stderr:
        pytest.skip('sanitizer runtime libraries not installed')
    assert build.returncode == 0, build.stdout + build.stderr
    run = subprocess.run([str(work / 'host_driver')], capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert 'all checks passed' in run.stdout


# ---- quadratic-form contraction: layout, equal-share schedule, merge (no GPU needed) --------

@pytest.mark.parametrize(
    'n_bins,n_central,by_type,n_tiles,n_rtiles,n_tables,separate,max_waves,order', [
        (100, 50, 1, 313, 1, 1, 0, 2048, 0),      # BASELINE configs[1]
        (200, 100, 1, 313, 1, 1, 1, 2048, 0),     # configs[2]
        (100, 50, 1, 391, 1, 25, 0, 2048, 1),     # configs[3]: table-major order
        (100, 50, 1, 7, 1, 25, 1, 2048, 1),
        (200, 100, 1, 40, 38, 1, 0, 2048, 2),     # configs[4] in float64: r-tile-major order
        (200, 100, 1, 313, 1, 1, 1, 2048, 3),     # configs[2]: unit-major order, separated
        (120, 60, 0, 313, 1, 64, 0, 3072, 4),     # database grid 4 x 4 x 4: table-synchronous
        (60, 30, 1, 37, 1, 25, 1, 1024, 4),       # ... tables not a multiple of 8, separated
        (20, 10, 0, 3, 2, 5, 0, 64, 4),           # ... fewer tables than XCDs, two r tiles
        (20, 10, 0, 3, 1, 5, 0, 30, 4),           # ... waves not a multiple of 8: table-major
        (200, 100, 0, 313, 38, 1, 0, 2048, 5),    # configs[4] in float64: unit-synchronous
        (200, 100, 1, 40, 3, 1, 1, 2048, 5),      # ... separated by galaxy type
        (24, 12, 0, 5, 2, 1, 0, 64, 5),           # ... few units: one part per (tile, r tile)
        (200, 100, 1, 313, 1, 1, 0, 2048, 3),     # ... and the total
        (37, 20, 1, 5, 2, 1, 1, 64, 3),           # unit-major with parts smaller than a share
        (13, 5, 0, 3, 2, 1, 0, 64, 0),
        (13, 5, 0, 3, 3, 1, 1, 64, 2),
        (13, 5, 1, 3, 2, 4, 1, 64, 1),
        (1, 1, 1, 1, 1, 1, 0, 2048, 0),
        (60, 30, 1, 1, 1, 1, 0, 2048, 0),         # un-batched call
    ])
def test_quad_schedule_covers_every_unit_once(lib, n_bins, n_central, by_type, n_tiles, n_rtiles,
                                              n_tables, separate, max_waves, order):
    """Every (draw tile, r tile, component, table, unit) in exactly one run, every slab
    written once and inside its group's range, equal shares (tc_debug_quad_schedule)."""
    n_waves, n_runs, n_slabs = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    from tabcorr_amd import _lib
    _lib.check(lib.tc_debug_quad_schedule(
        n_bins, n_central, by_type, n_tiles, n_rtiles, n_tables, separate, max_waves, 8, order,
        ctypes.byref(n_waves), ctypes.byref(n_runs), ctypes.byref(n_slabs), ctypes.byref(lo),
        ctypes.byref(hi)))
    assert 1 <= n_waves.value <= max_waves
    # equal shares (per r tile in r-tile-major order: a unit of slack per pass)
    # (table-synchronous: a unit per table piece of an XCD's range)
    # (unit-synchronous: whole items of up to all units of a (draw tile, r tile); a few per cent)
    n_units = (n_bins // 4 + 1) * (n_bins // 4 + 2) // 2 + 8
    assert hi.value - lo.value <= (n_rtiles if order == 2 else
                                   n_tables // 8 + 2 if order == 4 else
                                   n_units if order == 5 else 1)
    assert n_slabs.value >= 1 and n_runs.value >= n_waves.value


@pytest.mark.parametrize('n_prim,n_sec,n_r,separate,n_draws', [
    (7, 1, 5, 0, 70), (7, 1, 5, 1, 70), (6, 2, 23, 1, 33), (25, 1, 19, 0, 200),
    (3, 1, 1, 1, 64), (10, 1, 45, 0, 40)])
def test_quad_emulation_matches_the_oracle(lib, n_prim, n_sec, n_r, separate, n_draws):
    """The kernel's table layout, unit walk, operand lanes, equal-share schedule, the
    workgroup-level merge of the slabs and the finalisation's grouping, executed on the host
    lane by lane (tc_debug_quad_emulate), against the oracle's packed sums
    (tabcorr.py:641-655)."""
    from tabcorr_amd import _lib, synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(n_prim, n_sec, (n_r, ), 'auto', seed=n_prim + n_r)
    gal_type = table['gal_type']
    rng = np.random.default_rng(n_draws)
    shuffle = rng.permutation(len(gal_type))               # any row order is accepted
    gal_type = gal_type[shuffle]
    n_bins = len(gal_type)
    # the packed matrix of the shuffled rows
    index_1, index_2, _ = oracle.pair_indices(n_bins)
    full = np.zeros((n_r, n_bins, n_bins))
    lower = np.tril_indices(n_bins)
    full[:, lower[0], lower[1]] = table['tpcf_matrix']
    full = full + np.transpose(np.tril(full, -1), (0, 2, 1))
    full = full[:, shuffle][:, :, shuffle]
    matrix = np.ascontiguousarray(full[:, index_1, index_2])
    densities = np.exp(rng.normal(size=(n_bins, n_draws)))
    ldb = (n_draws + 63) // 64 * 64
    padded = np.zeros((n_bins, ldb))
    padded[:, :n_draws] = densities
    is_central = np.ascontiguousarray(oracle.is_centrals(gal_type), dtype=np.uint8)
    n_comp = 3 if separate else 1
    for max_waves, order in ((5, 0), (2048, 0), (5, 2), (2048, 2), (5, 3), (2048, 3), (16, 5),
                             (2048, 5)):
        out = np.zeros((n_draws, n_comp, n_r))
        _lib.check(lib.tc_debug_quad_emulate(
            n_bins, n_r, _lib.as_double_p(matrix), is_central.ctypes.data_as(_lib.c_uint8_p),
            1, separate, _lib.as_double_p(padded), ldb, n_draws, max_waves, 8, order,
            _lib.as_double_p(out)))
        prefactor = np.where(index_1 == index_2, 1.0, 2.0)
        weights = prefactor[None] * densities[index_1].T * densities[index_2].T   # (B, P)
        if separate:
            cen = is_central.astype(bool)
            kind = (~cen[index_1]).astype(int) + (~cen[index_2]).astype(int)
            for c in range(3):
                expect = np.einsum('rp,bp->br', matrix[:, kind == c], weights[:, kind == c])
                np.testing.assert_allclose(out[:, c], expect, rtol=1e-12,
                                           atol=1e-13 * np.max(np.abs(expect)) + 1e-300)
        else:
            expect = np.einsum('rp,bp->br', matrix, weights)
            np.testing.assert_allclose(out[:, 0], expect, rtol=1e-12)


def test_triangle_parts_of_the_one_launch_kernel():
    """The equal contiguous parts of the unit triangle that the waves of predict_fused_kernel
    walk: every unit exactly once, in row-major order, parts within one unit of equal size."""
    import ctypes
    from tabcorr_amd import _lib
    lib = _lib.load()
    for n_rb in (1, 2, 3, 4, 7, 25, 26, 50, 251):
        for n_parts in (1, 2, 4, 8, 16):
            arrays = [(ctypes.c_int * n_parts)() for _ in range(3)]
            _lib.check(lib.tc_debug_triangle_parts(n_rb, n_parts, *arrays))
            rb0, cb0, count = (list(a) for a in arrays)
            n_units = n_rb * (n_rb + 1) // 2
            assert sum(count) == n_units
            assert max(count) - min(count) <= 1
            position = 0
            for part in range(n_parts):
                if count[part] == 0:
                    continue
                assert 0 <= cb0[part] <= rb0[part] < n_rb
                assert rb0[part] * (rb0[part] + 1) // 2 + cb0[part] == position
                position += count[part]
            assert position == n_units


def test_node_groups_of_bins_that_share_their_quadrature_nodes():
    """hostmath.h: find_node_groups -- the secondary-percentile bins of a mass bin
    (tabcorr/tabcorr.py:186-205: same log_prim_haloprop_min / max, same galaxy type) form one
    group; every bin is in exactly one group, members ascending, groups ordered by their first
    member, the groups of centrals first; the real AbacusSummit fixture of the reference
    (cross, 1104 bins = 280 mass bins x 2 percentile bins x 2 types minus empty bins)."""
    import ctypes
    from tabcorr_amd import _lib, Interpolator
    lib = _lib.load()

    def groups_of(log_min, log_max, n_central):
        n = len(log_min)
        begin = np.zeros(n + 1, dtype=np.int32)
        member = np.zeros(n, dtype=np.int32)
        n_groups, n_cen = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.tc_debug_node_groups(
            n, n_central, _lib.as_double_p(np.ascontiguousarray(log_min, dtype=float)),
            _lib.as_double_p(np.ascontiguousarray(log_max, dtype=float)),
            begin.ctypes.data_as(_lib.c_int32_p), member.ctypes.data_as(_lib.c_int32_p),
            ctypes.byref(n_groups), ctypes.byref(n_cen)))
        begin = begin[:n_groups.value + 1]
        return [list(member[a:b]) for a, b in zip(begin[:-1], begin[1:])], n_cen.value

    def check(log_min, log_max, n_central):
        groups, n_cen = groups_of(log_min, log_max, n_central)
        assert sorted(g for group in groups for g in group) == list(range(len(log_min)))
        firsts = [group[0] for group in groups]
        assert firsts == sorted(firsts)
        assert all(group == sorted(group) for group in groups)
        assert all((group[0] < n_central) == (i < n_cen) for i, group in enumerate(groups))
        for group in groups:
            for g in group:
                assert (g < n_central) == (group[0] < n_central)
                assert log_min[g] == log_min[group[0]] and log_max[g] == log_max[group[0]]
        keys = {(g < n_central, log_min[g], log_max[g]) for g in range(len(log_min))}
        assert len(groups) == len(keys)
        return groups, n_cen

    # no secondary bins: every bin alone
    edges = np.linspace(10.5, 15.0, 13)
    lo, hi = np.tile(edges[:-1], 2), np.tile(edges[1:], 2)
    groups, n_cen = check(lo, hi, 12)
    assert [len(g) for g in groups] == [1] * 24 and n_cen == 12
    # three percentile bins per mass bin, secondary index slowest (the reference's row order)
    lo, hi = np.tile(edges[:-1], 6), np.tile(edges[1:], 6)
    groups, n_cen = check(lo, hi, 36)
    assert len(groups) == 24 and n_cen == 12
    assert groups[0] == [0, 12, 24] and groups[12] == [36, 48, 60]
    # only centrals / only satellites; a NaN edge stays alone
    assert check(lo[:36], hi[:36], 36)[1] == 12
    assert check(lo[:36], hi[:36], 0)[1] == 0
    bad = lo.copy()
    bad[[0, 12]] = np.nan
    groups, _ = groups_of(bad, hi, 36)
    assert [0] in groups and [12] in groups and len(groups) == 26
    # shuffled rows, ragged groups
    rng = np.random.default_rng(3)
    keep = np.sort(rng.permutation(72)[:55])
    cen = keep < 36
    order = np.concatenate([rng.permutation(keep[cen]), rng.permutation(keep[~cen])])
    check(lo[order], hi[order], int(np.sum(cen)))
    # the reference's AbacusSummit table
    table = Interpolator.read(os.path.join(REPO, 'tests', 'golden', 'ds_efficient.hdf5')).tabcorr_list[0]
    gal_type = table.gal_type
    central = np.asarray(gal_type['gal_type']) == 'centrals'
    order = np.concatenate([np.flatnonzero(central), np.flatnonzero(~central)])
    lo = np.asarray(gal_type['log_prim_haloprop_min'])[order]
    hi = np.asarray(gal_type['log_prim_haloprop_max'])[order]
    groups, n_cen = check(lo, hi, int(np.sum(central)))
    assert len(lo) == 1104 and len(groups) == 560 and max(len(g) for g in groups) == 2


def test_interpolator_lock_includes_its_tables():
    """the lock of an interpolator handle takes the locks of its tables as well, in
    one global order (two interpolators sharing tables cannot deadlock), and is re-entrant."""
    import threading
    from tabcorr_amd.interpolator import _HandleLocks

    class Table:
        def __init__(self):
            self.lock = threading.RLock()

    a, b, c = Table(), Table(), Table()
    first, second = _HandleLocks([a, b, a]), _HandleLocks([b, c, a])
    assert len(first._locks) == 3 and len(second._locks) == 4
    with first:
        with first:                                  # re-entrant on the same thread
            pass
        got = []
        thread = threading.Thread(target=lambda: got.append(a.lock.acquire(False)))
        thread.start()
        thread.join()
        assert got == [False]                        # a table of the interpolator is locked
        thread = threading.Thread(target=lambda: got.append(second.acquire(False)))
        thread.start()
        thread.join()
        assert got == [False, False]
        def try_unrelated():
            got.append(c.lock.acquire(False))
            c.lock.release()
        thread = threading.Thread(target=try_unrelated)
        thread.start()
        thread.join()
        assert got == [False, False, True]           # (an unrelated table is not)
    order = [id(lock) for lock in second._locks[1:]]
    assert order == sorted(order)
    done = []

    def hammer(locks):
        for _ in range(2000):
            with locks:
                pass
        done.append(1)
    threads = [threading.Thread(target=hammer, args=(l, )) for l in (first, second, first)]
    for thread in threads:
        thread.start()
    for thread in threads:
        thread.join(timeout=60)
    assert len(done) == 3


def test_central_moment_expansion_reproduces_the_node_loop():
    """csrc/series.h: the Taylor / Hermite re-ordering of a central bin's Gauss-Legendre sum
    (same inline code as the kernels, run on the host) against the node loop it replaces: to
    rounding wherever the expansion applies, with the term counts 8 ... 24 chosen from
    (half bin width) / sigma; and erf_gauss_fast's derivative against exp(-x^2)."""
    import ctypes
    from tabcorr_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-7, 7, 20000), [0.0, -0.0, 6.0, -6.0, 5.999, 1e-300, 30.0]])
    out = np.empty_like(x)
    _lib.check(lib.tc_debug_fastmath(5, len(x), _lib.as_double_p(x), _lib.as_double_p(out)))
    exact = (2 / np.sqrt(np.pi, dtype=np.longdouble) *
             np.exp(-x.astype(np.longdouble)**2)).astype(float)
    exact[np.abs(x) >= 6.0] = 0.0               # (zero from the clamp of erf_fast on)
    np.testing.assert_allclose(out, exact, rtol=1e-15)
    _lib.check(lib.tc_debug_fastmath(4, len(x), _lib.as_double_p(x), _lib.as_double_p(out)))
    from scipy.special import erf
    np.testing.assert_allclose(out, erf(x), rtol=0, atol=4.5e-16)

    used = {}
    for width, dist_index, n_gauss, sigma_range in (
            (0.09, -2.1, 10, (0.08, 0.9)),        # BASELINE configs[1]: 50 bins over 4.5 dex
            (0.09, -1.6, 10, (0.3, 0.9)),
            (0.0121, 7.5, 10, (0.05, 0.9)),       # the reference's AbacusSummit table
            (0.0121, -10.0, 10, (0.02, 0.2)),
            (0.15, -2.0, 10, (0.1, 0.9)),         # bolplanck wp table: 30 bins
            (0.09, -2.0, 100, (0.1, 0.9)),        # any n_gauss_prim: the moments are the bin's
            (0.09, -2.0, 3, (0.1, 0.9)),
            (0.3, -2.0, 10, (0.05, 0.5))):        # wide bins: mostly the node loop
        n = 4000
        log_min = rng.uniform(10.5, 15.0 - width)
        log_m_min = rng.uniform(log_min - 4.0, log_min + 4.0, n)
        # (an infinite or huge logMmin with a regular sigma sits on a plateau --
        # erf = -+1 -- where the recurrence overflows: the expansion must return -+m_0, not NaN)
        log_m_min[:8] = [np.inf, -np.inf, 1e15, -1e15, 1e300, -1e300, 1e12, -1e12]
        sigma = rng.uniform(*sigma_range, n) * rng.choice([1.0, 1.0, 1.0, -1.0], n)
        series, nodes = np.empty(n), np.empty(n)
        terms = np.zeros(n, dtype=np.int32)
        _lib.check(lib.tc_debug_central_series(
            n_gauss, log_min, log_min + width, dist_index, n, _lib.as_double_p(log_m_min),
            _lib.as_double_p(sigma), _lib.as_double_p(series), _lib.as_double_p(nodes),
            terms.ctypes.data_as(_lib.c_int32_p)))
        assert set(terms) <= {0, 8, 12, 16, 20, 24}
        np.testing.assert_allclose(series, nodes, rtol=0, atol=1.5e-15,
                                   err_msg='width %g, %d nodes' % (width, n_gauss))
        assert np.all(np.isfinite(series[:8])) and np.all(np.abs(nodes[:8]) > 0.99)
        # the number of terms follows (half width) / |sigma|
        h = 0.5 * width / np.abs(sigma)
        for count, limit in ((8, 0.0275), (12, 0.1100), (16, 0.2366), (20, 0.3879), (24, 0.5506)):
            assert np.all(h[terms == count] < limit * 1.001)
        assert np.all(h[terms == 0] > 0.5506 * 0.99)
        used.setdefault((width, n_gauss), set()).update(int(t) for t in terms)
    assert {0, 16, 20, 24} <= used[0.09, 10]
    assert {8, 12} <= used[0.0121, 10] and 0 in used[0.3, 10]


def test_satellite_binomial_expansion_reproduces_the_node_loop():
    """csrc/series.h (namespace sat): the binomial re-ordering of a satellite bin's node sum
    around the bin's reference mass, on the host with the kernels' inline code, against the node
    loop: to rounding of the log2 / exp2 chain wherever it applies (bins well above M0,
    0 <= alpha <= 4), with 12 ... 32 terms chosen from M0 alone."""
    from tabcorr_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    seen = {}
    for width, lo_range, dist_index, n_gauss in (
            (0.09, (10.5, 14.9), -2.0, 10),          # BASELINE configs[1]
            (0.0121, (11.8, 15.4), 7.0, 10),         # the reference's AbacusSummit table
            (0.0121, (11.8, 15.4), -10.0, 10),
            (0.15, (10.5, 14.8), -1.6, 10),          # bolplanck wp table
            (0.09, (10.5, 14.9), -2.0, 100),
            (0.09, (10.5, 14.9), -2.0, 4)):
        for trial in range(12):
            log_min = rng.uniform(*lo_range)
            n = 1500
            log_m0 = rng.uniform(10.5, 13.5, n)
            log_m1 = rng.uniform(12.5, 14.5, n)
            alpha = rng.uniform(-0.5, 4.5, n)
            series, nodes = np.empty(n), np.empty(n)
            terms = np.zeros(n, dtype=np.int32)
            _lib.check(lib.tc_debug_satellite_series(
                n_gauss, log_min, log_min + width, dist_index, n, _lib.as_double_p(log_m0),
                _lib.as_double_p(log_m1), _lib.as_double_p(alpha), _lib.as_double_p(series),
                _lib.as_double_p(nodes), terms.ctypes.data_as(_lib.c_int32_p)))
            assert set(terms) <= {0, 12, 16, 20, 24, 28, 32}
            assert np.all(terms[(alpha < 0) | (alpha > 4)] == 0)
            used = terms > 0
            assert np.all(10.0**log_m0[used] < 10.0**(log_min + 0.5 * width))
            np.testing.assert_allclose(series[used], nodes[used], rtol=2e-14)
            assert np.array_equal(series[~used], nodes[~used])
            seen.setdefault(width, set()).update(int(t) for t in terms)
    assert {0, 20, 24} <= seen[0.09] and {0, 12} <= seen[0.0121]


def test_no_kernel_spills_vector_registers():
    """every shipped instance of the prediction kernels compiles without
    vector-register spills and scratch (hipcc's kernel-resource-usage remarks of the inst_*.hip units,
    tools/kernel_resources.py) -- the two resident kernels included since their per-call
    address arithmetic stays inside the loop of calls (round 4: 1 and 19 spilled registers)."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import kernel_resources
    from tabcorr_amd import build
    if shutil.which('c++filt') is None:
        pytest.skip('c++filt not found')
    kernels = kernel_resources.parse(build.kernel_resource_remarks())
    names = kernel_resources.demangle(list(kernels))
    assert len(kernels) > 150
    allowed = {}
    spilling = {}
    for mangled, usage in kernels.items():
        name = names[mangled].replace('void ', '').split('(')[0]
        spills = usage.get('VGPRs Spill', 0)
        if spills > allowed.get(name, 0):
            spilling[names[mangled]] = (spills, usage.get('ScratchSize', 0))
        if name not in allowed:
            assert usage.get('ScratchSize', 0) == 0, (names[mangled], usage)
    assert not spilling, spilling
    # the instances whose two workgroups per CU need four waves per SIMD stay within 128
    for mangled, usage in kernels.items():
        name = names[mangled]
        if ('predict_fused_kernel<10, 5, false, false, false, 8, 64, false, ' in name or
                'predict_cross_fused_kernel<8,' in name or 'predict_cross_small_kernel' in name):
            assert usage['VGPRs'] <= 128, (name, usage['VGPRs'])


def test_consistency_checks_notice_in_place_changes():
    """the check a successful predict() lets later calls skip is keyed on the
    VALUES the checks compare (tabcorr/tabcorr.py:496-535: the components' halo-property keys,
    the redshifts), not only on object identities -- an in-place change is checked again."""
    from tabcorr_amd import TabCorr, Zheng07Model, synthetic
    table = synthetic.synthetic_table(6, 1, (4, ), 'auto', seed=1)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                                  table['attrs'])
    model = Zheng07Model(redshift=table['attrs']['redshift'])
    calls = []
    original = halotab._check_consistency
    halotab._check_consistency = lambda m: (calls.append(1), original(m))
    halotab._check_consistency_cached(model)
    halotab._check_consistency_cached(model)
    assert len(calls) == 1                              # (the second call is served by the cache)
    component = model._input_model_dictionary['satellites_occupation']
    component.prim_haloprop_key = 'halo_m200b'          # (same object, another value)
    with pytest.raises(ValueError, match='primary halo properties'):
        halotab._check_consistency_cached(model)
    component.prim_haloprop_key = table['attrs']['prim_haloprop_key']
    halotab._check_consistency_cached(model)
    halotab.attrs['redshift'] = 2.0                     # (the table's side of the comparison)
    with pytest.raises(ValueError, match='redshift'):
        halotab._check_consistency_cached(model)
    halotab.attrs['redshift'] = table['attrs']['redshift']
    model.redshift = table['attrs']['redshift'] + 0.01
    halotab._check_consistency_cached(model)
    # (first call, the two failures, the model's new redshift; the restored values equal what
    # last passed and are served by the cache)
    assert len(calls) == 4


def test_profiles_hold_the_kernels_they_quote():
    """the newest committed bench record quotes memory traffic only from
    PMC files that hold the very kernel it timed -- no `"traffic": null` for a configuration
    whose PMC file exists, no stale file of another build (bench_legs.pmc_traffic reports such a
    look-up under `failed_legs` instead of returning None quietly)."""
    import glob
    import json
    import bench_legs
    details = sorted(glob.glob(os.path.join(REPO, 'profiles', 'r[0-9][0-9]_bench_detail.json')))
    if not details:
        pytest.skip('no committed bench detail')
    detail = json.load(open(details[-1]))
    round_tag = os.path.basename(details[-1])[:3]
    assert 'pmc_counters' not in detail.get('failed_legs', {}), detail['failed_legs']
    records = [('headline', detail['roofline'])]
    records += [(record['tag'], record) for record in detail.get('other_configs', {}).values()]
    checked = 0
    for tag, record in records:
        source = record.get('traffic_source')
        assert record.get('traffic') is not None and source, (tag, 'no traffic quoted')
        path = os.path.join(REPO, source.split(' ')[0])
        assert os.path.basename(path).startswith(round_tag), (tag, source, 'another round')
        rows = [line for line in open(path).read().splitlines()
                if 'FETCH_SIZE' in line and not line.startswith('#')]
        kernel = record['kernel']
        assert any(bench_legs.same_kernel(kernel, row.split('FETCH_SIZE')[0]) for row in rows), (
            tag, kernel, 'not in', source)
        checked += 1
    assert checked >= 8, checked
    # ... and the look-up says so when a file does not hold a kernel
    del bench_legs.pmc_failures[:]
    assert bench_legs.pmc_traffic('tc::no_such_kernel<1>', '') == (None, None)
    assert bench_legs.pmc_failures and 'no_such_kernel' in bench_legs.pmc_failures[0]
    del bench_legs.pmc_failures[:]
