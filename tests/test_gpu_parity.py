"""Parity of the HIP path (through the C ABI) with the golden vectors recorded
from the reference and with the CPU oracle.  Needs an MI355X."""

import os
import sys

import numpy as np
import pytest

from util import (load_golden, table_from_golden, xi_keys, assert_rel)

pytestmark = pytest.mark.gpu

# north_star: within 1e-10 relative of the reference's CPU predict() in fp64.
RTOL = 1e-10


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def make_tabcorr(table, **kwargs):
    from tabcorr_amd import TabCorr
    return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                               table['tpcf_shape'], table['attrs'], **kwargs)


def check_against_golden(halotab, data, table, suffix='', prefix='',
                         **kwargs):
    theta = data['theta']
    if 'assembias' in kwargs:
        theta = np.hstack([theta, kwargs.pop('assembias')])
        kwargs['assembias'] = True
    ngal, xi = halotab.predict_batch(theta, **kwargs)
    assert_rel(ngal, data[prefix + 'ngal' + suffix], RTOL, 'ngal')
    assert_rel(xi, data[prefix + 'xi' + suffix], RTOL, 'xi')
    # one draw at a time: the un-batched path (a single fused launch where it applies)
    for index in range(min(3, len(theta))):
        ngal_1, xi_1 = halotab.predict_batch(theta[index:index + 1], **kwargs)
        assert_rel(ngal_1[0], data[prefix + 'ngal' + suffix][index], RTOL, 'ngal')
        assert_rel(xi_1[0], data[prefix + 'xi' + suffix][index], RTOL, 'xi')
    ngal_sep, xi_sep = halotab.predict_batch(
        theta, separate_gal_type=True, **kwargs)
    assert list(ngal_sep.keys()) == ['centrals', 'satellites']
    assert list(xi_sep.keys()) == xi_keys(table)
    for key in ngal_sep:
        assert_rel(ngal_sep[key], data[prefix + 'ngal_sep_' + key + suffix],
                   RTOL, key)
    for key in xi_sep:
        assert_rel(xi_sep[key], data[prefix + 'xi_sep_' + key + suffix], RTOL,
                   key)
    # the reference's own invariant (tests/test_general.py:8-28)
    assert_rel(sum(ngal_sep.values()), ngal, 1e-12)
    assert_rel(sum(xi_sep.values()), xi, 1e-12)


@pytest.mark.parametrize('name', ['leauthaud11_bolplanck_wp', 'leauthaud11_synthetic'])
def test_leauthaud11_family_on_device(name):
    """Second occupation family on the device: Leauthaud et al. (2011) on
    the Behroozi et al. (2010) relation, its inverse solved per quadrature node in the
    kernel, against fixtures recorded by running the reference with a duck model."""
    from tabcorr_amd import Leauthaud11Model
    from tabcorr_amd.models import LEAUTHAUD11_KEYS
    data = load_golden(name)
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    for modulate, suffix in ((True, ''), (False, '_nomodulate')):
        check_against_golden(halotab, data, table, suffix, family='leauthaud11',
                             modulate_with_cenocc=modulate)
        occupation = halotab.mean_occupation_batch(
            data['theta'], family='leauthaud11', modulate_with_cenocc=modulate)
        assert_rel(occupation, data['mean_occupation' + suffix], RTOL, floor=1e-13)
    # the model class (halotools' parameter names) through the scalar API
    theta = data['theta'][3]
    model = Leauthaud11Model(threshold=theta[11], redshift=table['attrs']['redshift'])
    for key, value in zip(('smhm_m0_0', 'smhm_m1_0', 'smhm_beta_0', 'smhm_delta_0',
                           'smhm_gamma_0'), theta[:5]):
        model.param_dict[key] = value
        model.param_dict[key[:-1] + 'a'] = 0.0
    for key, value in zip(LEAUTHAUD11_KEYS[10:], theta[5:11]):
        model.param_dict[key] = value
    ngal, xi = halotab.predict(model)
    assert_rel(ngal, data['ngal'][3], RTOL)
    assert_rel(xi, data['xi'][3], RTOL)
    assert_rel(halotab.mean_occupation(model), data['mean_occupation'][3], RTOL, floor=1e-13)
    # a NaN parameter rejects the draw, as the reference's NumPy arithmetic would
    bad = data['theta'][:4].copy()
    bad[1, 5] = np.nan        # scatter: centrals (and modulated satellites) NaN
    bad[2, 6] = np.nan        # alphasat: satellites NaN
    ngal, xi = halotab.predict_batch(bad, family='leauthaud11', modulate_with_cenocc=True)
    assert np.isnan(ngal[1]) and np.isnan(ngal[2]) and np.all(np.isnan(xi[1:3]))
    assert_rel(xi[[0, 3]], data['xi'][[0, 3]], RTOL)
    with pytest.raises(ValueError, match='14 columns'):
        halotab.predict_batch(data['theta'][:, :5], family='leauthaud11')
    with pytest.raises(ValueError):
        halotab.predict_batch(data['theta'], family='leauthaud11', assembias=True)


@pytest.mark.parametrize('name', ['bolplanck_wp', 'bolplanck_ds'])
def test_real_tables(name):
    data = load_golden(name)
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    check_against_golden(halotab, data, table)
    check_against_golden(halotab, data, table, '_ng1', n_gauss_prim=1)
    check_against_golden(halotab, data, table, '_ng100', n_gauss_prim=100)
    check_against_golden(halotab, data, table, '_modulate',
                         modulate_with_cenocc=True)
    for n_gauss, suffix in [(1, '_ng1'), (10, ''), (100, '_ng100')]:
        occ = halotab.mean_occupation_batch(data['theta'],
                                            n_gauss_prim=n_gauss)
        assert_rel(occ, data['mean_occupation' + suffix], RTOL)

    # the ndarray seam, one by one and as a batch
    for occ, ngal, xi in zip(data['occ_in'], data['occ_ngal'],
                             data['occ_xi']):
        n, x = halotab.predict(occ)
        assert isinstance(n, float) and x.shape == tuple(table['tpcf_shape'])
        assert_rel(n, ngal, RTOL)
        assert_rel(x, xi, RTOL)
    n, x = halotab.predict(data['occ_in'])
    assert_rel(n, data['occ_ngal'], RTOL)
    assert_rel(x, data['occ_xi'], RTOL)


def test_scalar_api_matches_reference_signature():
    from tabcorr_amd import Zheng07Model
    data = load_golden('bolplanck_wp')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    model = Zheng07Model(redshift=0.0)
    keys = ['logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha']
    for i in range(3):
        for key, value in zip(keys, data['theta'][i]):
            model.param_dict[key] = value
        ngal, xi = halotab.predict(model)
        assert isinstance(ngal, float)
        assert xi.shape == (19, )
        assert_rel(ngal, data['ngal'][i], RTOL)
        assert_rel(xi, data['xi'][i], RTOL)
        ngal_sep, xi_sep = halotab.predict(model, separate_gal_type=True)
        assert len(ngal_sep) == 2 and len(xi_sep) == 3
        assert_rel(xi_sep['centrals-satellites'],
                   data['xi_sep_centrals-satellites'][i], RTOL)
        occ = halotab.mean_occupation(model)
        assert_rel(occ, data['mean_occupation'][i], RTOL)
        ngal, xi = halotab.predict(model, n_gauss_prim=1)
        assert_rel(xi, data['xi_ng1'][i], RTOL)

    # mismatches raise ValueError like tabcorr/tabcorr.py:496-535
    with pytest.raises(ValueError):
        halotab.predict(Zheng07Model(redshift=0.5))
    with pytest.raises(ValueError):
        halotab.predict(Zheng07Model(prim_haloprop_key='halo_m200m'))
    with pytest.raises(ValueError):
        halotab.predict(Zheng07Model(sec_haloprop_key='halo_spin'))
    bad = Zheng07Model()
    bad.gal_types = ['centrals']
    with pytest.raises(ValueError):
        halotab.predict(bad)
    halotab.predict(Zheng07Model(redshift=0.5), check_consistency=False)
    with pytest.raises(ValueError):
        halotab.predict(np.ones(7))


@pytest.mark.parametrize('name', [
    'synthetic_cfg2', 'synthetic_small_auto', 'synthetic_small_cross',
    'synthetic_rp_pi', 'synthetic_r1'])
def test_synthetic(name):
    data = load_golden(name)
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    check_against_golden(halotab, data, table)
    if 'ngal_ng1' in data.files:
        check_against_golden(halotab, data, table, '_ng1', n_gauss_prim=1)
    if name == 'synthetic_small_auto':
        legacy = dict(table)
        names = [n for n in table['gal_type'].dtype.names
                 if n != 'prim_haloprop_dist_index']
        legacy['gal_type'] = table['gal_type'][names]
        check_against_golden(make_tabcorr(legacy), data, legacy,
                             prefix='legacy_')


def test_synthetic_assembias():
    data = load_golden('synthetic_cfg3')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    check_against_golden(halotab, data, table, assembias=data['assembias'])
    check_against_golden(halotab, data, table, prefix='plain_')


def test_generic_model_host_route():
    """A model the kernel does not know goes through host callbacks + device
    contraction and gives the same answer."""
    from tabcorr_amd import Zheng07Model

    class Opaque(Zheng07Model):
        _tabcorr_amd_device_model = None

    data = load_golden('synthetic_cfg3')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    keys = ['logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha']
    for i in range(2):
        model = Opaque(sec_haloprop_key='halo_nfw_conc')
        for key, value in zip(keys, data['theta'][i]):
            model.param_dict[key] = value
        model.param_dict['mean_occupation_centrals_assembias_param1'] = (
            data['assembias'][i, 0])
        model.param_dict['mean_occupation_satellites_assembias_param1'] = (
            data['assembias'][i, 1])
        ngal, xi = halotab.predict(model)
        assert_rel(ngal, data['ngal'][i], RTOL)
        assert_rel(xi, data['xi'][i], RTOL)


def test_unsorted_gal_type_rows():
    """Rows in arbitrary order (not centrals first) are handled by the
    library's internal permutation."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from oracle import tabcorr_oracle as oracle
    data = load_golden('synthetic_small_auto')
    table = table_from_golden(data)
    rng = np.random.default_rng(0)
    n_bins = len(table['gal_type'])
    perm = rng.permutation(n_bins)
    shuffled = dict(table)
    shuffled['gal_type'] = table['gal_type'][perm]
    # permute the packed matrix consistently
    full = np.zeros((table['tpcf_matrix'].shape[0], n_bins, n_bins))
    rows, cols = np.tril_indices(n_bins)
    full[:, rows, cols] = table['tpcf_matrix']
    full[:, cols, rows] = table['tpcf_matrix']
    full = full[:, perm][:, :, perm]
    shuffled['tpcf_matrix'] = np.ascontiguousarray(full[:, rows, cols])
    theta = data['theta']
    expect = oracle.predict_zheng07_batch(shuffled, theta)
    assert_rel(expect[0], data['ngal'], 1e-12)
    halotab = make_tabcorr(shuffled)
    ngal, xi = halotab.predict_batch(theta)
    assert_rel(ngal, data['ngal'], RTOL)
    assert_rel(xi, data['xi'], RTOL)
    ngal_sep, xi_sep = halotab.predict_batch(theta, separate_gal_type=True)
    for key in xi_sep:
        assert_rel(xi_sep[key], data['xi_sep_' + key], RTOL)


def test_ragged_and_empty_batches():
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from oracle import tabcorr_oracle as oracle
    from tabcorr_amd import synthetic
    data = load_golden('synthetic_cfg2')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    ngal, xi = halotab.predict_batch(np.zeros((0, 5)))
    assert ngal.shape == (0, ) and xi.shape == (0, 19)
    for n_draws in [1, 63, 64, 65, 200]:
        theta = synthetic.zheng07_draws(n_draws, seed=100 + n_draws)
        ngal, xi = halotab.predict_batch(theta)
        expect = oracle.predict_zheng07_batch(table, theta)
        assert_rel(ngal, expect[0], RTOL)
        assert_rel(xi, expect[1], RTOL)
    # either side of the 1 MB limit below which a call uses no copy commands (the kernels
    # address page-locked host buffers): same draws, same results (up to the summation
    # order of a different launch decomposition)
    theta = synthetic.zheng07_draws(5400, seed=7)
    ngal_copy, xi_copy = halotab.predict_batch(theta)              # 1.08 MB: copy engines
    ngal_direct, xi_direct = halotab.predict_batch(theta[:5200])   # 1.04 MB: direct
    assert_rel(ngal_copy[:5200], ngal_direct, 1e-13)
    assert_rel(xi_copy[:5200], xi_direct, 1e-13)
    expect = oracle.predict_zheng07_batch(table, theta[::270])
    assert_rel(ngal_copy[::270], expect[0], RTOL)
    assert_rel(xi_copy[::270], expect[1], RTOL)


# -- Interpolator -------------------------------------------------------------------

def make_interpolator(tables, keys, points):
    from tabcorr_amd import Interpolator
    return Interpolator([make_tabcorr(t) for t in tables],
                        {key: points[:, d] for d, key in enumerate(keys)})


def check_interpolator(interp, data, keys_xi):
    ngal, xi = interp.predict_batch(data['theta'], data['x'])
    assert_rel(ngal, data['ngal'], RTOL, 'ngal')
    assert_rel(xi, data['xi'], RTOL, 'xi')
    ngal_sep, xi_sep = interp.predict_batch(data['theta'], data['x'],
                                            separate_gal_type=True)
    assert list(ngal_sep.keys()) == ['centrals', 'satellites']
    assert list(xi_sep.keys()) == keys_xi
    for key in ngal_sep:
        assert_rel(ngal_sep[key], data['ngal_sep_' + key], RTOL, key)
    for key in xi_sep:
        assert_rel(xi_sep[key], data['xi_sep_' + key], RTOL, key)
    # out of range: ValueError unless extrapolate (interpolator.py:322-328)
    n_out = len(data['x_out'])
    theta_out = (data['theta'][3:3 + n_out] if n_out == 4 else
                 np.repeat(data['theta'][:1], n_out, axis=0))
    with pytest.raises(ValueError):
        interp.predict_batch(theta_out, data['x_out'])
    ngal, xi = interp.predict_batch(theta_out, data['x_out'], extrapolate=True)
    assert_rel(ngal, data['ngal_out'], RTOL)
    assert_rel(xi, data['xi_out'], RTOL)


@pytest.mark.parametrize('name', ['interp_2d_auto', 'interp_3d_cross',
                                  'interp_2d_mixed'])
def test_synthetic_interpolators(name):
    from util import interpolator_tables_from_golden
    data = load_golden(name)
    tables = interpolator_tables_from_golden(data)
    interp = make_interpolator(tables, [str(k) for k in data['keys']],
                               data['points'])
    for d in range(data['points'].shape[1]):
        assert_rel(interp.xp[d], data['xp%d' % d], 1e-15)
    check_interpolator(interp, data, xi_keys(tables[0]))


def test_abacus_interpolator_scalar_api():
    from tabcorr_amd import Zheng07Model
    data = load_golden('ds_efficient')
    tables = [table_from_golden(data, 'table%d_' % i) for i in range(4)]
    interp = make_interpolator(tables, [str(k) for k in data['keys']],
                               data['points'])
    check_interpolator(interp, data, ['centrals', 'satellites'])

    model = Zheng07Model(prim_haloprop_key='halo_m258m', redshift=0.5)
    keys = ['logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha']
    for i in range(4):
        for key, value in zip(keys, data['theta'][i]):
            model.param_dict[key] = value
        with pytest.raises(ValueError):       # log_eta missing
            interp.predict(model)
        model.param_dict['log_eta'] = data['x'][i, 0]
        ngal, xi = interp.predict(model)
        assert isinstance(ngal, float) and xi.shape == (13, )
        assert_rel(ngal, data['ngal'][i], RTOL)
        assert_rel(xi, data['xi'][i], RTOL)
        ngal_sep, xi_sep = interp.predict(model, separate_gal_type=True)
        assert_rel(xi_sep['satellites'], data['xi_sep_satellites'][i], RTOL)
        del model.param_dict['log_eta']
    model.param_dict['log_eta'] = data['x_out'][0, 0]
    with pytest.raises(ValueError):
        interp.predict(model)
    ngal, xi = interp.predict(model, extrapolate=True)

    # single table of the file through TabCorr
    halotab = interp.tabcorr_list[0]
    check_against_golden(halotab, data, tables[0], prefix='table0_')


def test_interpolator_generic_model_route():
    from tabcorr_amd import Zheng07Model
    from util import interpolator_tables_from_golden

    class Opaque(Zheng07Model):
        _tabcorr_amd_device_model = None

    data = load_golden('interp_2d_auto')
    tables = interpolator_tables_from_golden(data)
    keys = [str(k) for k in data['keys']]
    interp = make_interpolator(tables, keys, data['points'])
    names = ['logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha']
    for i in [0, 3, 7]:
        model = Opaque()
        for key, value in zip(names, data['theta'][i]):
            model.param_dict[key] = value
        for key, value in zip(keys, data['x'][i]):
            model.param_dict[key] = value
        ngal, xi = interp.predict(model)
        assert_rel(ngal, data['ngal'][i], RTOL)
        assert_rel(xi, data['xi'][i], RTOL)
        ngal_sep, xi_sep = interp.predict(model, separate_gal_type=True)
        for key in xi_sep:
            assert_rel(xi_sep[key], data['xi_sep_' + key][i], RTOL)


def test_interpolator_grid_validation():
    from tabcorr_amd import Interpolator
    from util import interpolator_tables_from_golden
    data = load_golden('interp_2d_auto')
    tables = interpolator_tables_from_golden(data)
    keys = [str(k) for k in data['keys']]
    halotabs = [make_tabcorr(t) for t in tables]
    points = data['points'].copy()
    with pytest.raises(ValueError):
        Interpolator(halotabs[:-1], {k: points[:, d] for d, k in
                                     enumerate(keys)})
    bad = points.copy()
    bad[0] = bad[1]
    with pytest.raises(ValueError, match='grid'):
        Interpolator(halotabs, {k: bad[:, d] for d, k in enumerate(keys)})
    with pytest.raises(ValueError, match='less than'):
        Interpolator(halotabs[:3], {'a': np.arange(3.0)})


# -- multi-GPU plumbing that can be exercised on one GPU ------------------------------

def test_rccl_communicator_single_rank():
    """dlopen of librccl, ncclCommInitRank, ncclGather on the comm stream behind the
    table's stream, slot events -- with one rank (the box has one GPU)."""
    import ctypes
    from tabcorr_amd import _lib, synthetic
    lib = _lib.load()
    buffer = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.check(lib.tc_comm_unique_id(buffer))
    comm = ctypes.c_void_p()
    _lib.check(lib.tc_comm_create(buffer.raw, 1, 0, ctypes.byref(comm)))
    data = load_golden('synthetic_cfg2')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    device = halotab.to_device()
    theta = _lib.contiguous(data['theta'])
    n_draws, n_r = len(theta), device.n_r
    count = n_draws * (1 + n_r)

    def dmalloc(n):
        ptr = ctypes.c_void_p()
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), n * 8))
        return ptr
    d_theta, d_out, d_recv = dmalloc(theta.size), dmalloc(2 * count), dmalloc(2 * count)
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p),
                                 theta.nbytes))
    for step in range(4):
        slot = step % 2
        if step >= 2:
            _lib.check(lib.tc_comm_release(comm, device.handle, slot))
        out = ctypes.c_void_p(d_out.value + slot * count * 8)
        _lib.check(lib.tc_predict_zheng07_batch_device(
            device.handle, d_theta, 5, n_draws, 10, 0, out,
            ctypes.c_void_p(out.value + n_draws * 8)))
        _lib.check(lib.tc_comm_gather(
            comm, device.handle, out,
            ctypes.c_void_p(d_recv.value + slot * count * 8), count, 0, slot))
    _lib.check(lib.tc_comm_synchronize(comm))
    _lib.check(lib.tc_comm_barrier(comm))
    host = np.empty(2 * count)
    _lib.check(lib.tc_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), d_recv,
                                 host.nbytes))
    for slot in range(2):
        part = host[slot * count:(slot + 1) * count]
        assert_rel(part[:n_draws], data['ngal'], RTOL)
        assert_rel(part[n_draws:].reshape(n_draws, n_r), data['xi'], RTOL)
    for ptr in (d_theta, d_out, d_recv):
        lib.tc_device_free(ptr)
    _lib.check(lib.tc_comm_destroy(comm))


@pytest.mark.parametrize('mode', ['default', 'both payloads', 'chi2 gather', 'interp5x5',
                                  'interp5x5 chi2'])
def test_bench_under_torchrun_single_rank(mode):
    """bench.py as the driver launches it (torch.distributed.run), one rank, with the
    communicator forced on: gloo control plane next to the HIP library, RCCL gather per
    block of steps -- the default workload, the 16-bytes-per-draw likelihood gather, and
    BASELINE configs[3] (Interpolator over a 5 x 5 grid, results gathered over RCCL)."""
    import json
    import subprocess
    from util import REPO
    env = dict(os.environ, TABCORR_AMD_FORCE_COMM='1', MASTER_ADDR='127.0.0.1')
    extra = {'default': ['--steps', '20', '--warmup', '3'],
             'both payloads': ['--steps', '20', '--warmup', '3', '--second-payload', '1',
                               '--gather-every', '5'],
             'chi2 gather': ['--steps', '20', '--warmup', '3', '--gather', 'chi2'],
             'interp5x5': ['--workload', 'interp5x5', '--draws', '12500', '--steps', '6',
                           '--warmup', '2', '--gather-every', '2'],
             'interp5x5 chi2': ['--workload', 'interp5x5', '--draws', '12500', '--steps', '6',
                                '--warmup', '2', '--gather-every', '2', '--gather',
                                'chi2']}[mode]
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port',
         str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '1', '--cpu-seconds', '0',
         '--settle-seconds', '0.05', '--other-configs', '0'] + extra,
        env=env, capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-2000:]
    # the record is the LAST line of stdout, short enough for the driver's parser, and its
    # roofline fraction follows from its own ms_per_step
    line = result.stdout.strip().splitlines()[-1]
    assert line.startswith('{') and len(line) < 4096, len(line)
    record = json.loads(line)
    roofline = record['roofline']
    from_step = roofline['flop_per_launch'] / (record['ms_per_step'] * 1e-3) / 1e12 / roofline['peak']
    assert abs(from_step / roofline['frac'] - 1) < 1e-9
    assert 'launches of the timed region' in roofline['launch_ms_source']
    assert record['detail'] == 'bench_detail.json'
    detail = json.load(open(os.path.join(REPO, 'bench_detail.json')))
    assert detail['value'] == record['value'] and 'three_kernel_path' in detail['roofline']
    assert record['n_gpus'] == 1 and record['value'] > 1e6
    assert record['config']['gather'] == 'rccl', record['config']
    assert record['parity_max_rel_vs_oracle'] < 1e-10
    assert 0.1 < record['roofline']['frac'] < 1.0
    if mode.startswith('interp5x5'):
        assert record['scaling'] == 'strong' and record['config']['n_tables'] == 25
        assert record['roofline']['kernel'] == 'tc::contract_quad_kernel<5, true>'
    if mode.endswith('chi2') or mode == 'chi2 gather':
        assert '16 B' in record['config']['gather_payload']
    else:
        assert '160 B' in record['config']['gather_payload']
    assert record['value_definition'] == 'device-resident'
    assert 0.05 < roofline['frac_by_duration'] < 1.0
    if not mode.startswith('interp5x5'):      # one launch per step, four lanes overlapping
        assert roofline['frac_by_duration'] <= roofline['frac'] * 1.001
        assert roofline['concurrent_launches'] >= 1.0
    if mode == 'both payloads':
        # (what --gpus > 1 times by default: the full results, then the likelihood payload in a
        # second region of the same run)
        second = record['second_payload']
        assert '16 B' in second['gather_payload'] and second['value'] > 1e6
        assert second['steps'] == record['steps']


@pytest.mark.parametrize('workload', ['cfg2', 'interp5x5'])
def test_bench_with_two_ranks_on_one_device(workload):
    """the >= 2-rank path of bench.py (ring of result blocks, barriers,
    max over ranks, gather offsets) next to real kernels -- two ranks under
    torch.distributed.run sharing the one device of this box (local_rank % device_count).  RCCL
    refuses two ranks on one device, so the agreed fallback is the expected outcome: every rank
    falls back to gloo ("gather": "gloo", `rccl_error` set) and the blocks travel on host
    arrays; rank 0 then holds every step of both ranks and compares each with its own
    evaluation of that rank's draws (a single-rank run of the same seeds): bit for bit
    (`gather_check`).  BASELINE configs[1] (weak scaling) and configs[3] (Interpolator over a
    5 x 5 grid, draws sharded round-robin: strong scaling)."""
    import json
    import subprocess
    from util import REPO
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    extra = {'cfg2': ['--steps', '20', '--warmup', '3'],
             'interp5x5': ['--workload', 'interp5x5', '--draws', '12500', '--steps', '6',
                           '--warmup', '2', '--gather-every', '2']}[workload]
    result = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', '2', '--master-addr', '127.0.0.1', '--master-port',
         str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--cpu-seconds', '0',
         '--settle-seconds', '0.05', '--other-configs', '0'] + extra,
        env=env, capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-3000:]
    line = result.stdout.strip().splitlines()[-1]
    assert line.startswith('{') and len(line) < 4096, len(line)
    record = json.loads(line)
    assert record['n_gpus'] == 2 and record['steps'] == int(extra[extra.index('--steps') + 1])
    config = record['config']
    assert config['gather'] in ('gloo', 'rccl'), config
    if config['gather'] == 'gloo':
        assert config.get('rccl_error'), config
        assert config['rccl_ranks'] == 0
    else:                              # (should RCCL ever accept two ranks on one device)
        assert config['rccl_ranks'] == 2
    check = record['gather_check']
    assert check['ranks'] == 2 and check['bit_equal'] and check['steps_checked'] >= 4, check
    assert record['parity_max_rel_vs_oracle'] < 1e-10
    assert record['scaling'] == ('strong' if workload == 'interp5x5' else 'weak')
    # the whole-job rate counts both ranks' draws over the max of the ranks' times
    per_step = 12500 if workload == 'interp5x5' else 2 * 10000
    assert abs(record['value'] * record['ms_per_step'] * 1e-3 / per_step - 1) < 0.02, record
    # the second payload (default for more than one rank) ran in the same job
    assert '16 B' in record['second_payload']['gather_payload']


def test_read_hdf5_and_predict():
    """End to end as a user of the reference would: read the reference's example file,
    predict with a model object."""
    from tabcorr_amd import TabCorr, Interpolator, Zheng07Model, hdf5
    from util import GOLDEN
    if not hdf5.available():
        pytest.skip('libhdf5 not found')
    data = load_golden('bolplanck_wp')
    halotab = TabCorr.read(os.path.join(GOLDEN, 'bolplanck_wp.hdf5'))
    model = Zheng07Model(redshift=0.0, **dict(zip(
        ['logMmin', 'sigma_logM', 'logM0', 'logM1', 'alpha'], data['theta'][0])))
    ngal, xi = halotab.predict(model)
    assert_rel(ngal, data['ngal'][0], RTOL)
    assert_rel(xi, data['xi'][0], RTOL)
    golden = load_golden('ds_efficient')
    interp = Interpolator.read(os.path.join(GOLDEN, 'ds_efficient.hdf5'))
    ngal, xi = interp.predict_batch(golden['theta'], golden['x'])
    assert_rel(ngal, golden['ngal'], RTOL)
    assert_rel(xi, golden['xi'], RTOL)


def test_chi2_fused_likelihood():
    data = load_golden('synthetic_cfg2')
    table = table_from_golden(data)
    halotab = make_tabcorr(table)
    rng = np.random.default_rng(3)
    observed = data['xi'][5] * (1 + 0.05 * rng.normal(size=19))
    a = rng.normal(size=(19, 19))
    precision = a @ a.T / np.outer(observed, observed)
    ngal, chi2 = halotab.chi2_batch(data['theta'], observed, precision)
    delta = data['xi'] - observed
    expect = np.einsum('bi,ij,bj->b', delta, precision, delta)
    assert_rel(ngal, data['ngal'], RTOL)
    assert_rel(chi2, expect, 1e-9)
    # again (cached data vector), one draw at a time, a changed data vector, a large batch
    ngal, chi2 = halotab.chi2_batch(data['theta'], observed, precision)
    assert_rel(chi2, expect, 1e-9)
    ngal_1, chi2_1 = halotab.chi2_batch(data['theta'][2:3], observed, precision)
    assert_rel(chi2_1[0], expect[2], 1e-9)
    shifted = observed * 1.01
    ngal, chi2 = halotab.chi2_batch(data['theta'], shifted, precision)
    delta = data['xi'] - shifted
    assert_rel(chi2, np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-9)
    big = np.tile(data['theta'], (1 + 70000 // len(data['theta']), 1))
    ngal, chi2 = halotab.chi2_batch(big, shifted, precision)
    assert_rel(chi2[:len(data['theta'])],
               np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-9)
    with pytest.raises(ValueError):
        halotab.chi2_batch(data['theta'], observed[:5], precision)


@pytest.mark.parametrize('shape', [(64, ), (70, ), (13, 40), (1500, )])
def test_chi2_wide_tables(shape):
    """Fused likelihood beyond the staged-matrix size (n_r > 64), with a non-symmetric
    weight matrix, ragged batch sizes and the reduced draws-per-workgroup geometry."""
    from tabcorr_amd import synthetic
    table = synthetic.synthetic_table(6, 1, shape, 'auto', seed=77)
    halotab = make_tabcorr(table)
    n_r = int(np.prod(shape))
    rng = np.random.default_rng(n_r)
    for n_draws in (1, 7, 8, 9, 203):
        theta = synthetic.zheng07_draws(n_draws, seed=n_draws)
        ngal_p, xi = halotab.predict_batch(theta)
        xi = xi.reshape(n_draws, n_r)
        observed = xi[0] * (1 + 0.05 * rng.normal(size=n_r))
        weight = rng.normal(size=(n_r, n_r)) / np.outer(np.abs(observed), np.abs(observed))
        ngal, chi2 = halotab.chi2_batch(theta, observed, weight)
        delta = xi - observed
        expect = np.einsum('bi,ij,bj->b', delta, weight, delta)
        scale = np.einsum('bi,ij,bj->b', np.abs(delta), np.abs(weight), np.abs(delta))
        assert np.array_equal(ngal, ngal_p)
        # a quadratic form with mixed signs cancels: error relative to the sum of magnitudes
        assert np.all(np.abs(chi2 - expect) <= 1e-12 * scale)


def test_sharded_predict_over_rccl_single_rank():
    """Product API for sharded prediction with the RCCL data plane (one rank here)."""
    import subprocess
    from util import REPO
    script = '''
import os, sys
import numpy as np
sys.path.insert(0, %(repo)r); sys.path.insert(0, os.path.join(%(repo)r, "tests"))
from util import load_golden, table_from_golden
from tabcorr_amd import TabCorr, parallel
comm = parallel.Communicator.from_env()
assert comm.gather_backend == "rccl", (comm.gather_backend, comm.rccl_error)
data = load_golden("synthetic_cfg2")
table = table_from_golden(data)
halotab = TabCorr.from_arrays(table["gal_type"], table["tpcf_matrix"], table["tpcf_shape"], table["attrs"])
ngal, xi = parallel.predict_batch_sharded(halotab, data["theta"], comm)
np.testing.assert_allclose(ngal, data["ngal"], rtol=1e-10)
np.testing.assert_allclose(xi, data["xi"], rtol=1e-10)
ngal_sep, xi_sep = parallel.predict_batch_sharded(halotab, data["theta"], comm, separate_gal_type=True)
np.testing.assert_allclose(sum(xi_sep.values()), data["xi"], rtol=1e-10)
for key in xi_sep:
    np.testing.assert_allclose(xi_sep[key], data["xi_sep_" + key], rtol=1e-10,
                               atol=1e-14 * np.max(np.abs(data["xi"])))
for key in ngal_sep:
    np.testing.assert_allclose(ngal_sep[key], data["ngal_sep_" + key], rtol=1e-10)
# Interpolator.predict sharded: device buffers + ncclGather as well (BASELINE configs[3])
from util import interpolator_tables_from_golden
from tabcorr_amd import Interpolator
idata = load_golden("interp_2d_auto")
tabs = [TabCorr.from_arrays(t["gal_type"], t["tpcf_matrix"], t["tpcf_shape"], t["attrs"])
        for t in interpolator_tables_from_golden(idata)]
keys = [str(k) for k in idata["keys"]]
interp = Interpolator(tabs, {k: idata["points"][:, d] for d, k in enumerate(keys)})
ngal, xi = parallel.predict_batch_sharded(interp, idata["theta"], comm, x=idata["x"])
np.testing.assert_allclose(ngal, idata["ngal"], rtol=1e-10)
np.testing.assert_allclose(xi, idata["xi"], rtol=1e-10, atol=1e-13 * np.max(np.abs(idata["xi"])))
ngal_sep, xi_sep = parallel.predict_batch_sharded(interp, idata["theta"], comm, x=idata["x"],
                                                  separate_gal_type=True)
np.testing.assert_allclose(sum(xi_sep.values()), idata["xi"], rtol=1e-9,
                           atol=1e-12 * np.max(np.abs(idata["xi"])))
bad = idata["x"].copy(); bad[3, 0] = np.nan
try:
    parallel.predict_batch_sharded(interp, idata["theta"], comm, x=bad)
    raise SystemExit("NaN x accepted")
except ValueError:
    pass
# the fused likelihood sharded: 16 bytes per draw through the gather
observed = data["xi"][1] * 1.02
weight = np.eye(19) / observed**2
ngal, chi2 = parallel.chi2_batch_sharded(halotab, data["theta"], observed, weight, comm)
delta = data["xi"] - observed
np.testing.assert_allclose(ngal, data["ngal"], rtol=1e-10)
np.testing.assert_allclose(chi2, np.einsum("bi,ij,bj->b", delta, weight, delta), rtol=1e-8)
n_r = idata["xi"].shape[1]
observed = idata["xi"][1] * 0.97
weight = np.eye(n_r) / observed**2
ngal, chi2 = parallel.chi2_batch_sharded(interp, idata["theta"], observed, weight, comm, x=idata["x"])
delta = idata["xi"] - observed
np.testing.assert_allclose(ngal, idata["ngal"], rtol=1e-10)
np.testing.assert_allclose(chi2, np.einsum("bi,ij,bj->b", delta, weight, delta), rtol=1e-8)
comm.close()
print("sharded ok")
''' % {'repo': REPO}
    env = dict(os.environ, TABCORR_AMD_FORCE_COMM='1', MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    result = subprocess.run([sys.executable, '-c', script], env=env,
                            capture_output=True, text=True, timeout=900)
    assert result.returncode == 0, result.stdout[-2000:] + result.stderr[-2000:]
    assert 'sharded ok' in result.stdout


def test_random_shapes_against_oracle():
    """Seeded sweep over table shapes the fixtures do not cover: every r tile width
    (R = 1 .. 70: sub-tiles of 4, one to three r tiles), odd bin counts, both modes,
    shuffled gal_type rows, batch sizes around the 64-draw tile -- float64 HIP path against
    the NumPy oracle."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    from oracle import tabcorr_oracle as oracle
    from tabcorr_amd import synthetic
    rng = np.random.default_rng(2024)
    cases = [(1, 1, (1, ), 'auto'), (1, 1, (3, ), 'cross'), (3, 1, (4, ), 'auto'),
             (5, 1, (5, ), 'auto'), (7, 1, (8, ), 'cross'), (9, 1, (11, ), 'auto'),
             (11, 1, (13, ), 'auto'), (6, 2, (16, ), 'auto'), (13, 1, (17, ), 'cross'),
             (8, 1, (21, ), 'auto'), (10, 1, (24, ), 'auto'), (4, 3, (27, ), 'auto'),
             (12, 1, (29, ), 'cross'), (9, 1, (32, ), 'auto'), (7, 1, (33, ), 'auto'),
             (5, 2, (6, 7), 'auto'), (17, 1, (64, ), 'auto'), (6, 1, (7, 10), 'cross'),
             (40, 1, (19, ), 'auto'), (33, 2, (10, ), 'auto')]
    for index, (n_prim, n_sec, shape, mode) in enumerate(cases):
        table = synthetic.synthetic_table(n_prim, n_sec, shape, mode, seed=100 + index)
        n_bins = len(table['gal_type'])
        if index % 2 == 1 and mode == 'auto' and n_bins > 2:
            # arbitrary row order: permute rows and the packed matrix consistently
            perm = rng.permutation(n_bins)
            full = np.zeros((table['tpcf_matrix'].shape[0], n_bins, n_bins))
            rows, cols = np.tril_indices(n_bins)
            full[:, rows, cols] = table['tpcf_matrix']
            full[:, cols, rows] = table['tpcf_matrix']
            full = full[:, perm][:, :, perm]
            table['gal_type'] = table['gal_type'][perm]
            table['tpcf_matrix'] = np.ascontiguousarray(full[:, rows, cols])
        n_draws = int(rng.choice([1, 2, 37, 64, 65, 130]))
        theta = synthetic.zheng07_draws(n_draws, seed=500 + index)
        halotab = make_tabcorr(table)
        expect_ngal, expect_xi = oracle.predict_zheng07_batch(table, theta)
        ngal, xi = halotab.predict_batch(theta)
        assert xi.shape == (n_draws, ) + tuple(shape)
        assert_rel(ngal, expect_ngal, RTOL)
        assert_rel(xi, expect_xi, RTOL)
        if index % 3 == 0:
            expect_n, expect_x = oracle.predict_zheng07_batch(
                table, theta, separate_gal_type=True)
            ngal_sep, xi_sep = halotab.predict_batch(theta, separate_gal_type=True)
            for key in expect_x:
                assert_rel(xi_sep[key], expect_x[key], RTOL)
            for key in expect_n:
                assert_rel(ngal_sep[key], expect_n[key], RTOL)


def test_leauthaud11_through_the_interpolator_and_chi2():
    """The occupation family flag travels through every entry point: Interpolator (batched
    and un-batched) and the fused likelihood, against the oracle."""
    from tabcorr_amd import Interpolator, Leauthaud11Model
    from oracle import tabcorr_oracle as oracle
    from util import interpolator_tables_from_golden
    data = load_golden('interp_2d_auto')
    tables = interpolator_tables_from_golden(data)
    keys = [str(k) for k in data['keys']]
    interp = make_interpolator(tables, keys, data['points'])
    theta = load_golden('leauthaud11_synthetic')['theta'][:6]
    x = data['x'][:6]
    ngal, xi = interp.predict_batch(theta, x, family='leauthaud11', modulate_with_cenocc=True)
    setup = oracle.interpolator_setup(tables, data['points'])
    for i in range(6):
        expect = oracle.interpolator_predict(tables, setup, oracle.Leauthaud11(theta[i]), x[i])
        assert_rel(ngal[i], expect[0], RTOL)
        assert_rel(xi[i], expect[1], RTOL, floor=1e-12)
    # un-batched: the model object through Interpolator.predict (one launch for all tables)
    model = Leauthaud11Model(threshold=theta[2, 11], redshift=0.0)
    for key, value in zip(('smhm_m0_0', 'smhm_m1_0', 'smhm_beta_0', 'smhm_delta_0',
                           'smhm_gamma_0', 'scatter_model_param1', 'alphasat', 'bsat', 'betasat',
                           'bcut', 'betacut'), theta[2, :11]):
        model.param_dict[key] = value
    for key, value in zip(keys, x[2]):
        model.param_dict[key] = value
    n1, x1 = interp.predict(model, check_consistency=False)
    assert_rel(n1, ngal[2], 1e-12)
    assert_rel(x1, xi[2], 1e-10, floor=1e-12)
    # fused likelihood on a single table
    table = tables[0]
    halotab = make_tabcorr(table)
    n_r = halotab.to_device().n_r
    observed = np.full(n_r, 30.0)
    precision = np.eye(n_r) * 0.01
    ngal_t, xi_t = halotab.predict_batch(theta, family='leauthaud11', modulate_with_cenocc=True)
    ngal_c, chi2 = halotab.chi2_batch(theta, observed, precision, family='leauthaud11',
                                      modulate_with_cenocc=True)
    delta = xi_t.reshape(len(theta), -1) - observed
    assert_rel(ngal_c, ngal_t, 1e-13)
    assert_rel(chi2, np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-9)


def test_flag_sweep_matches_the_oracle():
    """Every combination of the prediction flags (separate_gal_type, modulate_with_cenocc,
    Heaviside assembly bias, the Leauthaud11 family), several n_gauss_prim (compile-time
    unrolled 10 and the general loop), mode auto and cross, one- and two-dimensional
    tpcf_shape, primary x secondary bins, batch sizes around the tile sizes: seeded tables and
    draws against the NumPy oracle (`tabcorr/tabcorr.py:465-683`)."""
    from oracle import tabcorr_oracle as oracle
    from tabcorr_amd import synthetic
    rng = np.random.default_rng(77)
    leauthaud = load_golden('leauthaud11_synthetic')['theta']
    shapes = [(9, 1, (7, ), 'auto'), (6, 2, (19, ), 'auto'), (12, 1, (5, 6), 'auto'),
              (10, 2, (9, ), 'cross'), (21, 1, (23, ), 'auto')]
    n_cases = 0
    for index, (n_prim, n_sec, shape, mode) in enumerate(shapes):
        table = synthetic.synthetic_table(n_prim, n_sec, shape, mode, seed=300 + index)
        halotab = make_tabcorr(table)
        for family in ('zheng07', 'leauthaud11'):
            for modulate in (False, True):
                for assembias in ((False, True) if family == 'zheng07' else (False, )):
                    for separate in (False, True):
                        n_gauss = int(rng.choice([1, 3, 10, 17]))
                        n_draws = int(rng.choice([1, 5, 33, 64, 97]))
                        if family == 'zheng07':
                            theta = synthetic.zheng07_draws(n_draws, seed=900 + n_cases)
                            strengths = rng.uniform(-1.2, 1.2, (n_draws, 2))
                            expect = oracle.predict_zheng07_batch(
                                table, theta, separate_gal_type=separate, n_gauss_prim=n_gauss,
                                modulate_with_cenocc=modulate,
                                assembias=strengths if assembias else None)
                            batch = np.hstack([theta, strengths]) if assembias else theta
                        else:
                            batch = leauthaud[rng.integers(0, len(leauthaud), n_draws)]
                            expect = oracle.predict_leauthaud11_batch(
                                table, batch, separate_gal_type=separate, n_gauss_prim=n_gauss,
                                modulate_with_cenocc=modulate)
                        ngal, xi = halotab.predict_batch(
                            batch, separate_gal_type=separate, n_gauss_prim=n_gauss,
                            modulate_with_cenocc=modulate, assembias=assembias, family=family)
                        what = '%s %s n_gauss=%d modulate=%s assembias=%s separate=%s B=%d' % (
                            mode, shape, n_gauss, modulate, assembias, separate, n_draws)
                        if separate:
                            assert list(xi.keys()) == list(expect[1].keys()), what
                            for key in expect[0]:
                                assert_rel(ngal[key], expect[0][key], RTOL, what)
                            for key in expect[1]:
                                assert xi[key].shape == (n_draws, ) + tuple(shape), what
                                assert_rel(xi[key], expect[1][key], RTOL, what, floor=1e-13)
                        else:
                            assert xi.shape == (n_draws, ) + tuple(shape), what
                            assert_rel(ngal, expect[0], RTOL, what)
                            assert_rel(xi, expect[1], RTOL, what)
                        n_cases += 1
    assert n_cases == 5 * (8 + 4)


def test_interpolator_flag_sweep_matches_the_oracle():
    """The same sweep through `Interpolator.predict_batch` (`tabcorr/interpolator.py:124-216`):
    grids of one to three dimensions, mode auto and cross, every flag combination, batch sizes
    of 1 (the one-launch path) to a few tiles, extrapolation beyond the grid."""
    from oracle import tabcorr_oracle as oracle
    from tabcorr_amd import synthetic
    rng = np.random.default_rng(5)
    leauthaud = load_golden('leauthaud11_synthetic')['theta']
    grids = [((4, ), 7, 1, (6, ), 'auto'), ((4, 5), 5, 2, (9, ), 'auto'),
             ((4, 4, 4), 4, 1, (3, 4), 'auto'), ((5, 4), 6, 1, (11, ), 'cross')]
    n_cases = 0
    for index, (grid, n_prim, n_sec, shape, mode) in enumerate(grids):
        tables, keys, points = synthetic.synthetic_interpolator(
            grid, n_prim, n_sec, shape, mode, seed=40 + index)
        interp = make_interpolator(tables, keys, points)
        setup = oracle.interpolator_setup(tables, points)
        low, high = points.min(axis=0), points.max(axis=0)
        for family in ('zheng07', 'leauthaud11'):
            for modulate in (False, True):
                for assembias in ((False, True) if family == 'zheng07' else (False, )):
                    separate = bool(n_cases % 2)
                    extrapolate = n_cases % 3 == 0
                    n_draws = int(rng.choice([1, 3, 40, 70]))
                    span = 0.15 if extrapolate else 0.0
                    x = rng.uniform(low - span * (high - low), high + span * (high - low),
                                    (n_draws, len(grid)))
                    strengths = rng.uniform(-1, 1, (n_draws, 2))
                    if family == 'zheng07':
                        theta = synthetic.zheng07_draws(n_draws, seed=70 + n_cases)
                        models = [oracle.Zheng07(theta[i], modulate,
                                                 strengths[i] if assembias else None)
                                  for i in range(n_draws)]
                        batch = np.hstack([theta, strengths]) if assembias else theta
                    else:
                        batch = leauthaud[rng.integers(0, len(leauthaud), n_draws)]
                        models = [oracle.Leauthaud11(batch[i], modulate) for i in range(n_draws)]
                    ngal, xi = interp.predict_batch(
                        batch, x, separate_gal_type=separate, extrapolate=extrapolate,
                        modulate_with_cenocc=modulate, assembias=assembias, family=family)
                    what = '%s grid %s modulate=%s assembias=%s separate=%s B=%d' % (
                        family, grid, modulate, assembias, separate, n_draws)
                    for i in range(n_draws):
                        expect = oracle.interpolator_predict(
                            tables, setup, models[i], x[i], separate_gal_type=separate,
                            extrapolate=extrapolate)
                        if separate:
                            for key in expect[0]:
                                assert_rel(ngal[key][i], expect[0][key], RTOL, what)
                            for key in expect[1]:
                                assert_rel(xi[key][i], expect[1][key], RTOL, what, floor=1e-11)
                        else:
                            assert_rel(ngal[i], expect[0], RTOL, what)
                            assert_rel(xi[i], expect[1], RTOL, what, floor=1e-11)
                    n_cases += 1
    assert n_cases == 4 * 6


def test_interpolator_chi2_fused_likelihood():
    """`Interpolator.chi2_batch`: the fused likelihood behind the interpolated prediction
    (C ABI: tc_interp_chi2_zheng07_batch), against einsum on `predict_batch`'s results."""
    from util import interpolator_tables_from_golden
    data = load_golden('interp_2d_auto')
    tables = interpolator_tables_from_golden(data)
    keys = [str(k) for k in data['keys']]
    interp = make_interpolator(tables, keys, data['points'])
    theta, x = data['theta'], data['x']
    n_r = data['xi'].shape[1]
    rng = np.random.default_rng(9)
    observed = data['xi'][2] * (1 + 0.05 * rng.normal(size=n_r))
    a = rng.normal(size=(n_r, n_r))
    precision = a @ a.T / np.outer(observed, observed)
    expect_delta = data['xi'] - observed
    expect = np.einsum('bi,ij,bj->b', expect_delta, precision, expect_delta)
    for _ in range(2):                   # (the second call finds the data vector cached)
        ngal, chi2 = interp.chi2_batch(theta, x, observed, precision)
        assert_rel(ngal, data['ngal'], RTOL)
        assert_rel(chi2, expect, 1e-8)
    ngal_1, chi2_1 = interp.chi2_batch(theta[3:4], x[3:4], observed, precision)
    assert_rel(chi2_1[0], expect[3], 1e-8)
    shifted = observed * 0.98
    delta = data['xi'] - shifted
    ngal, chi2 = interp.chi2_batch(theta, x, shifted, precision, modulate_with_cenocc=False)
    assert_rel(chi2, np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-8)
    big = 1 + 3000 // len(theta)
    ngal, chi2 = interp.chi2_batch(np.tile(theta, (big, 1)), np.tile(x, (big, 1)), shifted,
                                   precision)
    assert_rel(chi2[:len(theta)], np.einsum('bi,ij,bj->b', delta, precision, delta), 1e-8)
    with pytest.raises(ValueError):
        interp.chi2_batch(theta, x, observed[:3], precision)
    with pytest.raises(ValueError):
        interp.chi2_batch(theta, x[:, :1], observed, precision)


def test_occupation_shortcuts_on_clustered_draws():
    """Posterior-like ensembles (all 64 draws of a tile close together) take the occupation
    kernel's wave-uniform shortcuts -- bins on the erf plateaus, satellite bins below every
    M0 -- which must give what the node loop and the oracle give, also next to tiles that
    do not qualify and with modulate_with_cenocc."""
    from tabcorr_amd import synthetic
    from oracle import tabcorr_oracle as oracle
    table = synthetic.synthetic_table(50, 1, (19, ), 'auto', seed=0)
    halotab = make_tabcorr(table)
    rng = np.random.default_rng(31)
    centre = np.array([12.3, 0.12, 13.4, 13.9, 1.05])
    tight = centre + rng.normal(0, 1, (192, 5)) * np.array([0.02, 0.005, 0.03, 0.03, 0.02])
    wide = synthetic.zheng07_draws(64, seed=32)
    theta = np.vstack([tight[:64], wide, tight[64:]])       # shortcut, no shortcut, shortcut
    for modulate in (False, True):
        occupation = halotab.mean_occupation_batch(theta, modulate_with_cenocc=modulate)
        ngal, xi = halotab.predict_batch(theta, modulate_with_cenocc=modulate)
        expect_occ = np.array([oracle.mean_occupation(table, oracle.Zheng07(t, modulate))
                               for t in theta[[0, 63, 64, 127, 128, 255]]])
        assert_rel(occupation[[0, 63, 64, 127, 128, 255]], expect_occ, RTOL, floor=1e-13)
        # the plateaus are exact: whole bins at 0 and at 1 for the tight draws
        assert np.any(occupation[0, :50] == 1.0) and np.any(occupation[0, 50:] == 0.0)
        expect = oracle.predict_zheng07_batch(table, theta[[0, 63, 64, 127, 128, 255]],
                                              modulate_with_cenocc=modulate)
        assert_rel(ngal[[0, 63, 64, 127, 128, 255]], expect[0], RTOL)
        assert_rel(xi[[0, 63, 64, 127, 128, 255]], expect[1], RTOL)
    # one draw at a time goes through the un-batched kernel (no shortcuts): same numbers
    single = halotab.predict_batch(theta[:1])
    assert_rel(single[1][0], halotab.predict_batch(theta)[1][0], 1e-12)
