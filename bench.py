#!/usr/bin/env python3
"""Benchmark of the batched ``TabCorr.predict()`` path on MI355X.

Metric (BASELINE.json): predict() calls per second -- Zheng07 HOD, 50 mass bins x
{centrals, satellites} (G = 100 halo/galaxy bins, P = 5050 packed pair columns), 19
r_p bins, float64.  One *step* is one pass of the hot path over one batch of 10^4
parameter draws against the resident synthetic table (BASELINE configs[1]): occupation
kernel -> contraction kernel -> finalisation kernel, with the draws already resident
in HBM and the results left in HBM.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: weak scaling, one process per GPU, every rank runs its own 10^4 draws per
step against its own replica of the table (the path shards over draws without any
data-path collective); the results of all steps are collected on rank 0 by one RCCL
gather over xGMI per block of --gather-every steps, on a second stream, overlapped with
the following steps.  PyTorch is only used for the gloo control plane (rendezvous,
barrier, max over ranks).

Rank 0 prints ONE JSON line.  Extra objects: ``roofline`` (contraction kernel:
algorithmic flop per launch / mean launch duration from HIP events on the kernel's
own stream, against the FP64 matrix/vector peak; measured with the batches serialised
(`TC_PIPELINE=0`, what `TC_LANES=1 rocprofv3 --kernel-trace --stats` shows), next to the
stretched duration in the overlapped timed region and the whole-step fraction) and ``cpu_baseline`` (the NumPy port
of the reference's predict(), oracle/tabcorr_oracle.py, timed on one host core).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

# FP64 peak of MI355X: 78.6 TFLOP/s (AMD datasheet, vector = matrix); the local
# microbenchmarks (tools/micro) measure 78.0 (v_mfma_f64_16x16x4) and 61
# (v_fma_f64, clock-limited) TFLOP/s.
FP64_PEAK_TFLOPS = 78.6
# HBM-side traffic of one contraction launch of the default workload (10^4 draws) from the
# committed PMC passes (profiles/r01_pmc_counters.txt: separate rocprofv3 --pmc
# FETCH_SIZE / WRITE_SIZE runs of this script; FETCH_SIZE doubled per the gfx950
# correction): 2 x 7299 KB + 12560 KB.  Compulsory bytes: table 0.83 MB + densities
# 8.0 MB + group partials 12.9 MB.  PMC counters cannot be read inside this process.
PMC_TRAFFIC_BYTES_DEFAULT_WORKLOAD = (2 * 7299.0 + 12560.0) * 1024

N_PRIM, N_SEC, N_R = 50, 1, 19
N_GAUSS = 10


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    # defaults: about one second of timed steps -- the chip's power management needs tens of
    # milliseconds of load to settle (tools/ramp.py: 10-ms regions started from idle run
    # 10-20 % slower than the sustained rate)
    parser.add_argument('--steps', type=int, default=20000)
    parser.add_argument('--warmup', type=int, default=2000)
    parser.add_argument('--draws', type=int, default=10000,
                        help='draws per GPU per step')
    parser.add_argument('--gather-every', type=int, default=32,
                        help='multi-GPU: steps per RCCL gather of the results')
    parser.add_argument('--cpu-seconds', type=float, default=12.0,
                        help='budget of the CPU baseline sample (0: skip)')
    parser.add_argument('--cpu-all-cores', type=int, default=1,
                        help='also time one independent CPU walker per host core')
    args = parser.parse_args()

    from tabcorr_amd import TabCorr, synthetic, _lib
    from tabcorr_amd.parallel import Communicator

    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run '
                     '--nproc-per-node %d' % (args.gpus, args.gpus))
        sys.exit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world_size))
    # The all-cores CPU baseline forks its workers, so it runs before this process
    # touches the GPU.
    cpu_all = None
    if world_size == 1 and args.cpu_seconds > 0 and args.cpu_all_cores:
        cpu_all = cpu_baseline_all_cores(args.cpu_seconds / 2)
    lib = _lib.load()
    _lib.require_device()
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    _lib.check(lib.tc_set_device(local_rank))
    comm = Communicator.from_env()
    rank = comm.rank

    table = synthetic.synthetic_table(N_PRIM, N_SEC, (N_R, ), 'auto', seed=0)
    halotab = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                  table['tpcf_shape'], table['attrs'])
    device = halotab.to_device()
    handle = device.handle
    n_draws = args.draws
    theta = synthetic.zheng07_draws(n_draws, seed=1 + rank)
    n_out = n_draws * (1 + N_R)        # ngal (B) followed by xi (B, R)

    def dmalloc(count):
        ptr = ctypes.c_void_p()
        _lib.check(lib.tc_device_malloc(ctypes.byref(ptr), count * 8))
        return ptr

    d_theta = dmalloc(theta.size)
    _lib.check(lib.tc_memcpy_h2d(d_theta, theta.ctypes.data_as(ctypes.c_void_p),
                                 theta.nbytes))
    # Results ring: 2 blocks of `every` steps.  A block is gathered on rank 0 (RCCL, own
    # stream) once its last step is queued; the other block keeps filling meanwhile.
    every = max(1, args.gather_every)
    n_slots = 2 * every
    d_out = dmalloc(n_slots * n_out)
    use_rccl = comm.comm is not None
    d_recv = dmalloc(2 * comm.world_size * every * n_out) if (
        use_rccl and comm.is_root) else ctypes.c_void_p()

    def out_ptr(slot, offset=0):
        return ctypes.c_void_p(d_out.value + (slot * n_out + offset) * 8)

    def gather_block(block, n_steps):
        recv = ctypes.c_void_p(
            d_recv.value + block * comm.world_size * every * n_out * 8) if (
                comm.is_root) else None
        comm.gather_device(handle, out_ptr(block * every), recv, n_steps * n_out, block)

    def step(index):
        slot = index % n_slots
        block = slot // every
        if use_rccl and index >= n_slots and slot % every == 0:
            comm.release(handle, block)         # the block's previous gather is done
        _lib.check(lib.tc_predict_zheng07_batch_device(
            handle, d_theta, 5, n_draws, N_GAUSS, 0, out_ptr(slot),
            out_ptr(slot, n_draws)))
        if use_rccl and slot % every == every - 1:
            gather_block(block, every)

    def flush(n_steps):
        """Gather the steps of a trailing, partly filled block."""
        if use_rccl and n_steps % every:
            gather_block(((n_steps - 1) % n_slots) // every, n_steps % every)

    def drain():
        _lib.check(lib.tc_table_synchronize(handle))
        comm.synchronize()
        _lib.check(lib.tc_device_synchronize())

    for index in range(args.warmup):
        step(index)
    flush(args.warmup)
    drain()

    # ---- timed region: exactly `steps` steps between barrier + device sync -----------
    comm.barrier()
    drain()
    t0 = time.perf_counter()
    for index in range(args.steps):
        step(index)
    flush(args.steps)
    drain()
    comm.barrier()
    elapsed = comm.max(time.perf_counter() - t0)

    if comm.dist is not None and not use_rccl:
        # RCCL unavailable: collect the last batch over gloo so that the job still
        # ends with the results on rank 0 (reported as "gather": "gloo").
        host = np.empty(n_out)
        _lib.check(lib.tc_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p),
                                     out_ptr((args.steps - 1) % n_slots), host.nbytes))
        comm.gather_host(host)

    # ---- dominant kernel: contraction, HIP events on its own stream --------------------
    n_bins = 2 * N_PRIM * N_SEC
    n_pairs = n_bins * (n_bins + 1) // 2
    flop_contract = n_draws * (2.0 * N_R * n_pairs + 3.0 * n_pairs)
    drain()
    def kernel_pass():
        ms = ctypes.c_float()
        _lib.check(lib.tc_table_timer_begin(handle, 1))
        for index in range(min(args.steps, 1000)):
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n_draws, N_GAUSS, 0, out_ptr(index % n_slots),
                out_ptr(index % n_slots, n_draws)))
        _lib.check(lib.tc_table_timer_end(handle, ctypes.byref(ms)))
        n_launch = ctypes.c_int()
        kernel_ms = ctypes.c_float()
        _lib.check(lib.tc_table_kernel_time(handle, ctypes.byref(n_launch),
                                            ctypes.byref(kernel_ms)))
        return n_launch, kernel_ms

    # as in the timed region: consecutive batches overlap on the table's two lanes, so
    # the contraction shares the chip with the next batch's occupation kernel
    n_launch, kernel_ms = kernel_pass()
    # the kernel alone (batches serialised), for reference
    os.environ['TC_PIPELINE'] = '0'
    _, isolated_ms = kernel_pass()
    os.environ['TC_PIPELINE'] = '1'
    launch = [ctypes.c_int() for _ in range(4)]
    lib.tc_table_last_launch(handle, *[ctypes.byref(v) for v in launch])

    result = None
    if comm.is_root:
        # spot check against the CPU oracle (4 draws)
        host = np.empty(n_out)
        _lib.check(lib.tc_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), out_ptr(0),
                                     host.nbytes))
        from oracle import tabcorr_oracle as oracle
        expect = oracle.predict_zheng07_batch(table, theta[:4])
        xi = host[n_draws:].reshape(n_draws, N_R)
        parity = float(max(np.max(np.abs(host[:4] / expect[0] - 1)),
                           np.max(np.abs(xi[:4] / expect[1] - 1))))

        # the kernel's own roofline: launches serialised (TC_PIPELINE=0 pass above); in the
        # timed region kernels of neighbouring batches share the chip, which stretches
        # every launch (reported as overlapped_*)
        achieved = flop_contract / (isolated_ms.value * 1e-3) / 1e12
        overlapped = flop_contract / (kernel_ms.value * 1e-3) / 1e12
        total_draws = comm.world_size * n_draws * args.steps
        result = {
            'metric': 'predict_calls_per_sec',
            'value': total_draws / elapsed,
            'unit': 'calls/s',
            'n_gpus': comm.world_size,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': 'BASELINE configs[1]: Zheng07 predict(), synthetic auto '
                            'table 50 mass bins x {cen,sat} (G=100, P=5050), 19 rp '
                            'bins, n_gauss_prim=10, batch of %d draws per GPU per '
                            'step, draws and results resident in HBM' % n_draws,
                'draws_per_gpu_per_step': n_draws,
                'n_bins': n_bins, 'n_pairs': n_pairs, 'n_r': N_R,
                'parallelism': 'draws sharded over %d GPU(s), table replicated' %
                               comm.world_size,
                'gather': comm.gather_backend,
                'gather_every_steps': every,
            },
            'roofline': {
                'kernel': 'tc::contract_mfma_kernel<20, false>',
                'bound': 'mfma',
                'achieved': achieved,
                'peak': FP64_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': achieved / FP64_PEAK_TFLOPS,
                'traffic': PMC_TRAFFIC_BYTES_DEFAULT_WORKLOAD if n_draws == 10000 else None,
                'traffic_source': 'profiles/r01_pmc_counters.txt (offline PMC passes)',
                'flop_per_launch': flop_contract,
                'mean_launch_ms': isolated_ms.value,
                'launches_timed': n_launch.value,
                'overlapped_launch_ms': kernel_ms.value,
                'overlapped_frac': overlapped / FP64_PEAK_TFLOPS,
                'step_frac': flop_contract / (elapsed / args.steps) / 1e12 /
                             FP64_PEAK_TFLOPS,
                'workgroups': launch[0].value,
                'waves_per_workgroup': launch[1].value,
                'lds_bytes': launch[3].value,
            },
            'parity_max_rel_vs_oracle': parity,
            'device': _lib.device_name(),
        }
        if comm.rccl_error:
            result['config']['rccl_error'] = comm.rccl_error
        if comm.world_size == 1 and args.cpu_seconds > 0:
            result['cpu_baseline'] = cpu_baseline(table, args.cpu_seconds)
            if cpu_all is not None:
                result['cpu_baseline']['all_cores'] = cpu_all
        print(json.dumps(result), flush=True)

    for ptr in (d_theta, d_out, d_recv):
        if ptr.value:
            lib.tc_device_free(ptr)
    comm.barrier()
    comm.close()


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


if __name__ == '__main__':
    main()
