#!/usr/bin/env python3
"""Benchmark of the batched ``TabCorr.predict()`` path on MI355X.

Metric (BASELINE.json): predict() calls per second -- Zheng07 HOD, 50 mass bins x
{centrals, satellites} (G = 100 halo/galaxy bins, P = 5050 packed pair columns), 19
r_p bins, float64.  One *step* is one pass of the hot path over one batch of 10^4
parameter draws against the resident synthetic table (BASELINE configs[1]): one
launch of predict_fused_kernel (occupations -> quadratic form -> results in a workgroup).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \\
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

``value`` is the DEVICE-RESIDENT rate: draws already in HBM when the timed region starts,
results left in HBM (multi-GPU: gathered on rank 0 over RCCL).  The rate SURVEY.md section
8d defines -- theta on the host to (ngal, xi) on the host, PCIe included -- is reported
next to it as ``host_to_host`` and is never ``value``.

Multi-GPU: weak scaling, one process per GPU, every rank runs its own 10^4 draws per
step against its own replica of the table (the path shards over draws without any
data-path collective); the results of all steps are collected on rank 0 by one RCCL
gather over xGMI per block of --gather-every steps, on a second stream, overlapped with
the following steps.  The SAME workload and payload for every N: (ngal, xi) of every draw,
160 bytes; with --gpus > 1 the fused likelihood (``--gather chi2``: 16 bytes per draw, what
an MCMC needs back) is timed as well, in a second region of the same run
(``second_payload``).  ``--workload interp5x5`` is BASELINE configs[3]: ``Interpolator.predict`` over a
5 x 5 grid of such tables, 10^5 draws per step sharded round-robin over the ranks (strong
scaling), gathered the same way.  PyTorch is only used for the gloo control plane.

Rank 0 prints ONE short JSON line (< 4 KB) as the LAST line of stdout:

  metric, value, unit, n_gpus, steps, warmup, ms_per_step, dtype, config, ...
  roofline       the kernel of the timed region.  ``frac`` = algorithmic flop of one step /
                 (elapsed / steps) / peak OF THE TIMED REGION (reproducible from
                 ``ms_per_step``: flop_per_launch / ms_per_step / peak; one launch per step).
                 ``mean_launch_ms``: that kernel's launches with their own start / stop events
                 (hipExtLaunchKernelGGL: the dispatch's begin and end, the interval
                 `rocprofv3 --kernel-trace --stats` reports) -- over the timed region itself
                 when it has <= 1024 steps (the driver's command), else over 1000 launches of
                 the same stream of calls right after it (``launch_ms_source`` says which);
                 launches of the four lanes overlap by design, ``concurrent_launches`` =
                 mean_launch_ms / ms_per_step and ``frac_by_duration`` = flop_per_launch /
                 mean_launch_ms / peak.  ``traffic``: HBM bytes per launch from the committed
                 PMC passes.
  cpu_baseline   the NumPy port of the reference's predict() (oracle/tabcorr_oracle.py),
                 timed on one host core and with one walker per core.
  value_host_to_host   the SURVEY 8d rate (PCIe included), never ``value``.

Everything else goes to the sidecar ``bench_detail.json`` (next to this file, and a copy under
gpurun_out/ when that directory exists) and to ``detail ...`` lines printed BEFORE the final
one: the full roofline record (kernel alone, three-kernel path, methods), host_to_host*,
batch_sizes, unbatched_us (one predict(model) per call, README.md:72-75), tabulation (SURVEY
8f.4), other_configs (BASELINE configs[2], [3], [4] and the reference's own table shapes; the
legs live in bench_legs.py).  A secondary leg that fails is recorded there and never costs the
headline line.
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

from bench_legs import (CONFIG_TAGS, FP64_PEAK_TFLOPS, N_GAUSS, N_PRIM, N_R, N_SEC,   # noqa: E402
                        Device, cpu_baseline, cpu_baseline_all_cores, host_pipelined,
                        kernel_time, matrix_pipe_busy, other_configs, pair_flops, pmc_traffic,
                        sustained, tabulation, time_calls, unbatched)


HEADLINE_KEYS = ('metric', 'value', 'value_definition', 'unit', 'n_gpus', 'steps', 'warmup',
                 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
ROOFLINE_KEYS = ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_by_duration',
                 'flop_per_launch', 'mean_launch_ms', 'launch_ms_source', 'concurrent_launches',
                 'traffic', 'traffic_source')
CONFIG_KEYS = ('workload', 'draws_per_gpu_per_step', 'n_tables', 'gather', 'gather_payload',
               'gather_every_steps', 'lanes', 'rccl_ranks', 'rccl_error')
MAX_LINE = 4096


def headline(result, detail_path):
    """The short record: what the driver parses and cross-checks (< MAX_LINE bytes)."""
    line = {key: result[key] for key in HEADLINE_KEYS}
    line['config'] = {key: result['config'][key] for key in CONFIG_KEYS
                      if key in result['config']}
    roofline = {key: result['roofline'][key] for key in ROOFLINE_KEYS}
    if roofline.get('traffic_source'):
        roofline['traffic_source'] = roofline['traffic_source'].split(' ')[0]
    line['roofline'] = roofline
    cpu = result.get('cpu_baseline')
    if cpu is not None:
        line['cpu_baseline'] = {key: cpu[key] for key in ('value', 'unit', 'cores', 'kind',
                                                          'sample')}
        if cpu.get('all_cores'):
            line['cpu_baseline']['all_cores'] = {key: cpu['all_cores'][key]
                                                 for key in ('value', 'cores')}
    else:
        line['cpu_baseline'] = None
    line['value_host_to_host'] = result.get('value_host_to_host')
    line['parity_max_rel_vs_oracle'] = result['parity_max_rel_vs_oracle']
    if 'second_payload' in result:
        line['second_payload'] = {key: result['second_payload'][key]
                                  for key in ('gather_payload', 'value', 'ms_per_step', 'steps')}
    if result.get('gather_check'):
        line['gather_check'] = {key: result['gather_check'][key]
                                for key in ('ranks', 'steps_checked', 'bit_equal')}
    if result.get('failed_legs'):
        line['failed_legs'] = sorted(result['failed_legs'])
    line['detail'] = detail_path
    text = json.dumps(line)
    if len(text) >= MAX_LINE:          # (cannot happen with the keys above; never lose the line)
        line['config'] = {'workload': line['config']['workload'][:200]}
        line['roofline'].pop('launch_ms_source', None)
        text = json.dumps(line)
    return text


def check_gathered(region, comm, lib, _lib, dev, handle, interp_mode, chi2_mode, n_draws, n_r,
                   n_gauss, data_p, precision_p, synthetic, interp, synchronize):
    """Rank 0: every step of every rank in the receive buffer == this rank's own evaluation of
    that rank's draws (cfg2: seed 1 + rank; interp5x5: the rank's round-robin shard)."""
    fetch = region.gathered()
    synchronize()
    if not comm.is_root or fetch is None:
        return None
    world = comm.world_size
    if interp_mode:
        theta_all = synthetic.zheng07_draws(n_draws * world, seed=5)
        rng = np.random.default_rng(6)
        x_all = np.stack([rng.uniform(xp[0], xp[-1], size=len(theta_all))
                          for xp in interp.xp], axis=-1)
    scratch = dev.malloc(region.n_out)
    second = ctypes.c_void_p(scratch.value + n_draws * 8)
    steps = mismatches = 0
    # (the newest gather of every block: older entries of the log were overwritten; at most the
    # two newest blocks -- a block of 32 steps of 8 ranks is 400 MB to bring to the host)
    newest = {}
    for entry in region.ring.log:
        newest[entry[0]] = entry
    newest = dict(sorted(newest.items(), key=lambda item: item[1][1])[-2:])
    for r in range(world):
        if interp_mode:
            theta_r = np.ascontiguousarray(theta_all[r::world])
            d_x = dev.upload(np.ascontiguousarray(x_all[r::world]))
        else:
            theta_r = synthetic.zheng07_draws(n_draws, seed=1 + r)
        d_theta = dev.upload(theta_r)
        if interp_mode and chi2_mode:
            _lib.check(lib.tc_interp_chi2_zheng07_batch_device(
                handle, d_theta, 5, d_x, n_draws, n_gauss, 0, data_p, precision_p, scratch,
                second))
        elif interp_mode:
            _lib.check(lib.tc_interp_predict_zheng07_batch_device(
                handle, d_theta, 5, d_x, n_draws, n_gauss, 0, scratch, second))
        elif chi2_mode:
            _lib.check(lib.tc_chi2_zheng07_batch_device(
                handle, d_theta, 5, n_draws, n_gauss, 0, data_p, precision_p, scratch, second))
        else:
            _lib.check(lib.tc_predict_zheng07_batch_device(
                handle, d_theta, 5, n_draws, n_gauss, 0, scratch, second))
        synchronize()
        own = dev.download(scratch, region.n_out)
        for entry in newest.values():
            for step, offset in region.ring.steps_in(entry, r):
                steps += 1
                if not np.array_equal(fetch(offset, region.n_out), own, equal_nan=True):
                    mismatches += 1
    return {'ranks': world, 'steps_checked': steps, 'steps_differing': mismatches,
            'bit_equal': mismatches == 0 and steps > 0,
            'what': 'every step of every rank held by rank 0 after the timed region against '
                    'rank 0\'s own evaluation of that rank\'s draws'}


def emit(result):
    """Sidecar with everything, `detail` lines, then the ONE short JSON line (last on stdout)."""
    import bench_legs
    if bench_legs.pmc_failures:
        # (a committed PMC file that does not hold the kernel that was timed: say so)
        result.setdefault('failed_legs', {})['pmc_counters'] = '; '.join(bench_legs.pmc_failures)
    detail_path = None
    for directory in (REPO, os.path.join(REPO, 'gpurun_out')):
        if not os.path.isdir(directory):
            continue
        try:
            with open(os.path.join(directory, 'bench_detail.json'), 'w') as stream:
                json.dump(result, stream, indent=1)
            detail_path = detail_path or 'bench_detail.json'
        except OSError:
            pass
    for key, value in result.items():
        if isinstance(value, dict) and key not in ('config', ):
            if key == 'other_configs':
                for name, record in value.items():
                    print('detail other_configs[%s] %s' % (name, json.dumps(record)))
            else:
                print('detail %s %s' % (key, json.dumps(value)))
    sys.stdout.flush()
    print(headline(result, detail_path), flush=True)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    # defaults: about one second of timed steps -- the chip's power management needs tens of
    # milliseconds of load to settle (tools/archive/ramp.py: 10-ms regions started from idle run
    # 10-20 % slower than the sustained rate)
    parser.add_argument('--steps', type=int, default=None)
    parser.add_argument('--warmup', type=int, default=None)
    parser.add_argument('--workload', choices=['cfg2', 'interp5x5'], default='cfg2')
    parser.add_argument('--draws', type=int, default=None,
                        help='cfg2: draws per GPU per step (10^4); interp5x5: draws per '
                             'step over ALL GPUs (10^5)')
    parser.add_argument('--gather', choices=['full', 'chi2'], default=None,
                        help='what every step leaves behind / multi-GPU: what is gathered on '
                             'rank 0 in the region `value` is timed on: (ngal, xi) of every '
                             'draw (160 B per draw; default for every N: the configuration '
                             'BASELINE.json names), or the fused likelihood (ngal, chi2) (16 B '
                             'per draw: what an MCMC needs back).  With --gpus > 1 the other '
                             'payload is timed as well, in a second region of the same run '
                             '(`second_payload`)')
    parser.add_argument('--second-payload', type=int, default=None,
                        help='time the other payload in a second region (default: 1 for '
                             '--gpus > 1, else 0)')
    parser.add_argument('--gather-every', type=int, default=0,
                        help='multi-GPU: steps per RCCL gather of the results (default: 32, or '
                             'a quarter of --steps for short runs so that the gathers overlap '
                             'the steps instead of trailing them)')
    parser.add_argument('--lanes', type=int, default=0,
                        help='pipelining lanes of the timed region (default: library '
                             'default, 4); 1 serialises the kernels, e.g. under rocprofv3')
    parser.add_argument('--option', action='append', default=[],
                        help='developer A/B: tc_table_set_option name=value (repeatable)')
    parser.add_argument('--settle-seconds', type=float, default=0.3,
                        help='untimed load before the warm-up steps (power management)')
    parser.add_argument('--cpu-seconds', type=float, default=12.0,
                        help='budget of the CPU baseline sample (0: skip)')
    parser.add_argument('--cpu-all-cores', type=int, default=1,
                        help='also time one independent CPU walker per host core')
    parser.add_argument('--other-configs', type=int, default=1,
                        help='measure BASELINE configs[2..4] as well (single GPU only)')
    parser.add_argument('--detail', type=int, default=1,
                        help='0 = headline only: skip the secondary legs (host-to-host, batch '
                             'sizes, un-batched calls, tabulation, other configurations)')
    parser.add_argument('--check-gather', type=int, default=1,
                        help='multi-rank: rank 0 compares every gathered step with its own '
                             'evaluation of the sending rank\'s draws, bit for bit (after the '
                             'timed region)')
    parser.add_argument('--only-config', choices=CONFIG_TAGS, default=None,
                        help='measure ONLY that configuration of other_configs and print its '
                             'record (for rocprofv3 runs: tools/profile_round.sh)')
    args = parser.parse_args()
    interp_mode = args.workload == 'interp5x5'
    if args.gather is None:
        args.gather = 'full'      # the same workload and payload for every N
    if args.steps is None:
        args.steps = 200 if interp_mode else 20000
    if args.warmup is None:
        args.warmup = 20 if interp_mode else 2000

    from tabcorr_amd import TabCorr, Interpolator, synthetic, _lib
    from tabcorr_amd.parallel import Communicator, ResultRing

    world_size = int(os.environ.get('WORLD_SIZE', '1'))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            sys.exit('bench.py --gpus %d must be launched with torch.distributed.run '
                     '--nproc-per-node %d' % (args.gpus, args.gpus))
        sys.exit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world_size))
    # The all-cores CPU baseline forks its workers, so it runs before this process
    # touches the GPU.
    cpu_all = None
    if world_size == 1 and args.cpu_seconds > 0 and args.cpu_all_cores and not interp_mode:
        cpu_all = cpu_baseline_all_cores(args.cpu_seconds / 2)
    lib = _lib.load()
    _lib.require_device()
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # (more ranks than devices -- two ranks rehearsing the multi-rank path on a one-GPU box --
    # share devices round-robin; RCCL then refuses the communicator and the gather runs over
    # gloo: Communicator._init_rccl)
    _lib.check(lib.tc_set_device(local_rank % max(1, _lib.device_count())))
    if args.only_config:
        def make_only(table, **kwargs):
            return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                       table['tpcf_shape'], table['attrs'], **kwargs)
        print(json.dumps(other_configs(lib, _lib, make_only, synthetic, Interpolator,
                                       args.cpu_seconds, only=args.only_config,
                                       lanes=args.lanes, options=args.option)), flush=True)
        return
    comm = Communicator.from_env()
    rank = comm.rank
    dev = Device(lib, _lib)

    def make(table, **kwargs):
        return TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'],
                                   table['tpcf_shape'], table['attrs'], **kwargs)

    # ---- workload -------------------------------------------------------------------------
    table = synthetic.synthetic_table(N_PRIM, N_SEC, (N_R, ), 'auto', seed=0)
    interp = None
    if interp_mode:
        tables, keys, points = synthetic.synthetic_interpolator((5, 5), N_PRIM, N_SEC, (N_R, ),
                                                                'auto', seed=7)
        interp = Interpolator([make(t) for t in tables],
                              {k: points[:, d] for d, k in enumerate(keys)})
        total_draws = args.draws or 100000
        # round-robin shard of this rank (BASELINE configs[3]); equal shares per rank
        n_draws = (total_draws + world_size - 1) // world_size
        theta_all = synthetic.zheng07_draws(n_draws * world_size, seed=5)
        rng = np.random.default_rng(6)
        x_all = np.stack([rng.uniform(xp[0], xp[-1], size=len(theta_all))
                          for xp in interp.xp], axis=-1)
        theta = np.ascontiguousarray(theta_all[rank::world_size])
        x = np.ascontiguousarray(x_all[rank::world_size])
        device = interp.to_device()
        handle = device.handle
        timer_handle = device.tables[0].handle
        n_tables = 25
    else:
        halotab = make(table)
        device = halotab.to_device()
        handle = timer_handle = device.handle
        n_draws = args.draws or 10000
        theta = synthetic.zheng07_draws(n_draws, seed=1 + rank)
        x = None
        n_tables = 1
    if args.lanes > 0:
        _lib.check(lib.tc_table_set_option(timer_handle, b'lanes', args.lanes))
    for option in args.option:
        name, value = option.split('=')
        _lib.check(lib.tc_table_set_option(timer_handle, name.encode(), int(value)))
    d_theta = dev.upload(theta)
    d_x = dev.upload(x) if interp_mode else None
    data_vector = np.full(N_R, 50.0)
    precision = np.eye(N_R) * 1e-2
    data_p, precision_p = _lib.as_double_p(data_vector), _lib.as_double_p(precision)
    every = args.gather_every if args.gather_every > 0 else (
        32 if args.steps >= 128 else max(2, args.steps // 4))
    use_rccl = comm.comm is not None
    interp_handle = handle if interp_mode else None

    class Region:
        """One payload: the ring of result slots (tabcorr_amd.parallel.ResultRing: 4 blocks of
        `every` steps; a block is gathered on rank 0 -- RCCL, own stream -- once its last step
        is queued, the other blocks keep filling meanwhile), the step and its gathers."""

        def __init__(self, chi2_mode):
            self.chi2_mode = chi2_mode
            self.n_out = n_draws * (2 if chi2_mode else 1 + N_R)   # ngal | chi2, or ngal | xi
            # RCCL not usable on every rank (Communicator.rccl_error): the blocks travel over
            # gloo on host arrays -- slower (a device synchronisation, a download and a TCP
            # round trip per block), same layout on the root, "gather": "gloo" in the record
            use_gloo = comm.dist is not None and not use_rccl
            self.h_recv = None
            self.ring = ResultRing(
                self.n_out, every, comm.world_size,
                gather=self.gather if use_rccl else self.gather_gloo if use_gloo
                else (lambda *a: None),
                release=(lambda block: comm.release(
                    timer_handle, block, interp_handle=interp_handle)) if use_rccl
                else None)               # (the communicator has four send-buffer slots)
            self.n_slots = self.ring.n_slots
            self.d_out = dev.malloc(self.ring.ring_elements)
            self.d_recv = (dev.malloc(self.ring.recv_elements)
                           if (use_rccl and comm.is_root) else ctypes.c_void_p())
            if use_gloo and comm.is_root:
                self.h_recv = np.zeros(self.ring.recv_elements)

        def gather_gloo(self, block, send_offset, recv_offset, count):
            synchronize()
            host = dev.download(ctypes.c_void_p(self.d_out.value + send_offset * 8), count)
            parts = comm.gather_host(host)
            if comm.is_root:
                for r, part in enumerate(parts):
                    begin = recv_offset + r * count
                    self.h_recv[begin:begin + count] = part

        def gathered(self):
            """fetch(offset, count) -> that stretch of the root's receive buffer as a host
            array, or None (not the root / nothing gathered)."""
            if not comm.is_root:
                return None
            if self.h_recv is not None:
                return lambda offset, count: self.h_recv[offset:offset + count]
            if self.d_recv.value:
                comm.synchronize()
                return lambda offset, count: dev.download(
                    ctypes.c_void_p(self.d_recv.value + offset * 8), count)
            return None

        def gather(self, block, send_offset, recv_offset, count):
            recv = (ctypes.c_void_p(self.d_recv.value + recv_offset * 8)
                    if comm.is_root else None)
            comm.gather_device(timer_handle,
                               ctypes.c_void_p(self.d_out.value + send_offset * 8), recv,
                               count, block, interp_handle=interp_handle)

        def out_ptr(self, slot, offset=0):
            return ctypes.c_void_p(self.d_out.value + (slot * self.n_out + offset) * 8)

        def predict(self, slot):
            # (the slot's two pointers made once: 1 us of interpreter time per call otherwise)
            pointers = self.__dict__.setdefault('_slot_pointers', {})
            if slot not in pointers:
                pointers[slot] = (self.out_ptr(slot), self.out_ptr(slot, n_draws))
            out, second = pointers[slot]
            if interp_mode and self.chi2_mode:
                _lib.check(lib.tc_interp_chi2_zheng07_batch_device(
                    handle, d_theta, 5, d_x, n_draws, N_GAUSS, 0, data_p, precision_p,
                    out, second))
            elif interp_mode:
                _lib.check(lib.tc_interp_predict_zheng07_batch_device(
                    handle, d_theta, 5, d_x, n_draws, N_GAUSS, 0, out, second))
            elif self.chi2_mode:
                _lib.check(lib.tc_chi2_zheng07_batch_device(
                    handle, d_theta, 5, n_draws, N_GAUSS, 0, data_p, precision_p, out, second))
            else:
                _lib.check(lib.tc_predict_zheng07_batch_device(
                    handle, d_theta, 5, n_draws, N_GAUSS, 0, out, second))

        def step(self, index):
            self.ring.before_step(index)
            self.predict(self.ring.slot(index))
            self.ring.after_step(index)

        def flush(self, n_steps):
            """Gather the steps of a trailing, partly filled block."""
            self.ring.flush(n_steps)

    region = Region(args.gather == 'chi2')
    chi2_mode, n_out, n_slots = region.chi2_mode, region.n_out, region.n_slots
    predict, step, flush, out_ptr = region.predict, region.step, region.flush, region.out_ptr
    lanes_used = args.lanes if args.lanes > 0 else 4
    if use_rccl and args.lanes == 0 and not interp_mode and every < 16 and not chi2_mode:
        # Frequent gathers of the full results: the communicator's stream is a fifth stream on
        # the runtime's four hardware queues and shares one with a lane, whose kernels then
        # wait behind the gather (and the other way round).  Three lanes leave it a queue of
        # its own: 51.5 against 58 us per step in the short run with a gather every 5 steps
        # (one rank, tools/archive/r02_forced_comm.sh); with a gather every 32 steps four lanes stay
        # ahead, and so they do with the likelihood payload (16 B per draw: the gather kernel
        # is short; 49.4-52.0 against 53.2-53.5 us per step, tools/archive/r03_forced_comm.sh).
        lanes_used = 3
        _lib.check(lib.tc_table_set_option(timer_handle, b'lanes', lanes_used))

    def synchronize():
        if interp_mode:
            _lib.check(lib.tc_interp_synchronize(handle))
        else:
            _lib.check(lib.tc_table_synchronize(handle))

    def drain():
        synchronize()
        comm.synchronize()
        _lib.check(lib.tc_device_synchronize())

    # ---- dominant kernel, serialised: per-launch start / stop events ----------------------
    # Before the timed region, in the state rocprofv3's serialised trace of this script sees
    # (profiles/*_kernel_stats_lanes1.csv): after the long pipelined region the chip is a few
    # per cent slower for a while (round 2: 38.7 us there against 37.5 here and 36.5 in the
    # trace).  No collectives inside: every rank does the same on its own.
    _lib.check(lib.tc_table_set_option(timer_handle, b'pipeline', 0))
    isolated_ms, n_launch, serial_step_ms = kernel_time(
        lib, _lib, timer_handle, lambda: predict(0), synchronize)
    _lib.check(lib.tc_table_set_option(timer_handle, b'pipeline', 1))

    # Untimed: load the chip until its power management has settled (tools/archive/ramp.py: a
    # region started from idle runs 10-20 % slower for the first tens of milliseconds),
    # then the W warm-up steps.  The driver's short runs (--steps 20) would otherwise time
    # the ramp, not the path.
    settle_steps = 0
    if args.settle_seconds > 0:
        probe = 100 if not interp_mode else 5
        t_probe = time.perf_counter()
        for index in range(probe):
            step(index)
        flush(probe)
        drain()
        per_step = max((time.perf_counter() - t_probe) / probe, 1e-6)
        # (the same count on every rank: the gathers inside step() are collectives)
        settle_steps = int(comm.max(int(args.settle_seconds / per_step) + 1))
        # in chunks with a drain after each: the runtime retires finished commands lazily, and
        # thousands of them left over from one long burst make later launches stall
        # (tools/archive/stall.py: 60-100 us per launch for a while, as long as 10 steps)
        chunk = 4 * n_slots
        for begin in range(0, settle_steps, chunk):
            count = min(chunk, settle_steps - begin)
            for index in range(count):
                step(index)
            flush(count)
            drain()
        settle_steps += probe
    for index in range(args.warmup):
        step(index)
    flush(args.warmup)
    drain()

    # ---- timed region: exactly `steps` steps between barrier + device sync -----------
    # (no garbage collection inside it: a full collection pass of the interpreter stalls
    # the enqueueing thread for hundreds of microseconds, longer than 10 steps)
    import gc

    # Per-launch start / stop events INSIDE the timed region when it is short (the driver's
    # --steps 20): the mean launch duration of the roofline record then belongs to the very
    # launches `value` times.  (20 000 event pairs in the default run would cost more than they
    # tell: there the same stream of calls is sampled right after the region.)  The event pairs
    # are created beforehand.  ONE launch in `lanes_used` carries a pair -- the launches of one
    # lane: a pair costs its launch ~1.5 us on the queue, and pairs on all 20 launches of the
    # driver's --steps 20 made that region 1.3-2.0 us per step longer than the same 20 steps
    # without them (42.2 against 40.5 us, four runs each in alternation on one box).
    events_in_region = args.steps <= 1024

    def region_events(begin):
        if not events_in_region:
            return None
        if begin:
            _lib.check(lib.tc_table_timer_begin(timer_handle, max(1, lanes_used)))
            return None
        ms, count, mean = ctypes.c_float(), ctypes.c_int(), ctypes.c_float()
        _lib.check(lib.tc_table_timer_end(timer_handle, ctypes.byref(ms)))
        _lib.check(lib.tc_table_kernel_time(timer_handle, ctypes.byref(count),
                                            ctypes.byref(mean)))
        return (mean.value, count.value) if count.value else None

    def timed(payload):
        gc.disable()
        comm.barrier()
        drain()
        region_events(True)
        t0 = time.perf_counter()
        for index in range(args.steps):
            payload.step(index)
        t_queued = time.perf_counter()
        payload.flush(args.steps)
        drain()
        t_drained = time.perf_counter()
        comm.barrier()
        seconds = comm.max(time.perf_counter() - t0)
        gc.enable()
        launches = region_events(False)
        return seconds, {'enqueue': (t_queued - t0) * 1e6, 'drain': (t_drained - t_queued) * 1e6,
                         'barrier': (time.perf_counter() - t_drained) * 1e6}, launches

    if events_in_region:
        region_events(True)
        for index in range(args.steps):
            step(index)
        flush(args.steps)
        drain()
        region_events(False)
    elapsed, breakdown, region_launches = timed(region)

    # The other payload, same workload, same steps: a second region of the same run (default
    # for --gpus > 1, where what travels to rank 0 is the difference between the two).
    second = None
    want_second = (args.second_payload if args.second_payload is not None
                   else int(comm.world_size > 1))
    if want_second:
        other = Region(not chi2_mode)
        for index in range(args.warmup):
            other.step(index)
        other.flush(args.warmup)
        drain()
        second_elapsed, second_breakdown, _ = timed(other)
        second = {
            'gather_payload': 'ngal + chi2 (16 B per draw)' if other.chi2_mode
                              else 'ngal + xi (%d B per draw)' % (8 * (1 + N_R)),
            'value': comm.world_size * n_draws * args.steps / second_elapsed,
            'unit': 'calls/s', 'ms_per_step': second_elapsed / args.steps * 1e3,
            'steps': args.steps, 'warmup': args.warmup,
            'timed_region_breakdown_us': second_breakdown,
            'what': 'the same workload and steps with the other payload, timed right after '
                    'the region of `value` (barrier + device sync on both sides, max over '
                    'ranks)'}

    # What rank 0 holds at the end: every step of every rank that the ring's blocks still carry,
    # against rank 0's OWN evaluation of that rank's draws (the same seeds: a single-rank run of
    # that shard) -- bit for bit (the multi-rank path next to real kernels).
    gather_check = None
    if comm.dist is not None and args.check_gather:
        gather_check = check_gathered(region, comm, lib, _lib, dev, handle, interp_mode,
                                      chi2_mode, n_draws, N_R, N_GAUSS, data_p, precision_p,
                                      synthetic, interp, synchronize)

    # ---- dominant kernel: contraction, per-launch start / stop events -------------------
    n_bins = 2 * N_PRIM * N_SEC
    n_pairs = n_bins * (n_bins + 1) // 2
    flop_contract = n_draws * n_tables * pair_flops(n_bins, N_R)
    drain()
    # (the serialised pass ran before the timed region, see above) ... and stretched by the
    # kernels of neighbouring batches in the overlapped regime of the timed region
    overlapped_ms, _, overlapped_wall_ms = kernel_time(
        lib, _lib, timer_handle, lambda: predict(0), synchronize)
    launch = [ctypes.c_int() for _ in range(4)]
    lib.tc_table_last_launch(timer_handle, *[ctypes.byref(v) for v in launch])
    # One launch per step (predict_fused_kernel: occupation -> quadratic form -> results inside a
    # workgroup) when the library chose it for this batch size: the dominant kernel of the timed
    # region is then that one, and its launches overlap by design (one per lane).  Also timed:
    # the same kernel alone on the chip.
    fused_active = not interp_mode and launch[1].value == 8 and launch[2].value == 0
    fused_alone_ms = None
    if fused_active:
        user_fused = dict(o.split('=') for o in args.option).get('fused', '1')
        _lib.check(lib.tc_table_set_option(timer_handle, b'pipeline', 0))
        _lib.check(lib.tc_table_set_option(timer_handle, b'fused', 2))
        fused_alone_ms, _, _ = kernel_time(lib, _lib, timer_handle, lambda: predict(0),
                                           synchronize, n_launches=300, max_seconds=0.3)
        _lib.check(lib.tc_table_set_option(timer_handle, b'fused', int(user_fused)))
        _lib.check(lib.tc_table_set_option(timer_handle, b'pipeline', 1))

    result = None
    if comm.is_root:
        from oracle import tabcorr_oracle as oracle
        # spot check against the CPU oracle (4 draws)
        predict(0)
        drain()
        host = dev.download(out_ptr(0), n_out)
        if interp_mode:
            setup = oracle.interpolator_setup(tables, points)
            expect = oracle.interpolator_predict_zheng07_batch(tables, setup, theta[:2], x[:2])
            if chi2_mode:
                delta = expect[1] - data_vector
                chi2 = np.einsum('bi,ij,bj->b', delta, precision, delta)
                parity = float(max(np.max(np.abs(host[:2] / expect[0] - 1)),
                                   np.max(np.abs(host[n_draws:n_draws + 2] / chi2 - 1))))
            else:
                xi = host[n_draws:].reshape(n_draws, N_R)
                parity = float(max(np.max(np.abs(host[:2] / expect[0] - 1)),
                                   np.max(np.abs(xi[:2] / expect[1] - 1))))
        else:
            expect = oracle.predict_zheng07_batch(table, theta[:4])
            if chi2_mode:
                delta = expect[1] - data_vector
                chi2 = np.einsum('bi,ij,bj->b', delta, precision, delta)
                parity = float(max(np.max(np.abs(host[:4] / expect[0] - 1)),
                                   np.max(np.abs(host[n_draws:n_draws + 4] / chi2 - 1))))
            else:
                xi = host[n_draws:].reshape(n_draws, N_R)
                parity = float(max(np.max(np.abs(host[:4] / expect[0] - 1)),
                                   np.max(np.abs(xi[:4] / expect[1] - 1))))

        three_kernel_name = ('tc::contract_quad_kernel<5, true>' if interp_mode
                             else 'tc::contract_quad_kernel<5, false>')
        # (the one-launch kernel's instance from what the library reports about its last launch:
        # draws per workgroup 64 -- the throughput form --, 32, or 40 -- the latency form, which
        # calls that have the chip to themselves take; the deferring instance follows the
        # option "fused_defer", 2 by default)
        fused_draws = next((d for d in (64, 40, 32)
                            if launch[0].value == (n_draws + d - 1) // d), 64)
        fused_defer = dict(o.split('=') for o in args.option).get('fused_defer', '2')
        kernel_name = ('tc::predict_fused_kernel<10, 5, false, false, false, 8, %d, false, %s>'
                       % (fused_draws, fused_defer if fused_draws != 32 else '0')
                       if fused_active else three_kernel_name)
        headline_traffic = (pmc_traffic(kernel_name, 'interp5x5' if interp_mode else '')
                            if n_draws == (100000 if interp_mode else 10000) // (
                                comm.world_size if interp_mode else 1)
                            else (None, None))
        step_seconds = elapsed / args.steps
        # The dominant kernel's launch durations: events inside the timed region when it is
        # short, else a sample of the same stream of calls right after it (one launch per step:
        # predict_fused_kernel, launches of different lanes sharing the chip) or -- three
        # kernels per step -- the contraction kernel serialised before the region
        sample_concurrency = overlapped_ms / overlapped_wall_ms
        if region_launches is not None:
            launch_ms, n_launch = region_launches
            launch_source = ('events on %d of the %d launches of the timed region (one lane\'s: '
                             'every %s launch)' % (n_launch, args.steps,
                                                   {1: '', 2: 'second', 3: 'third', 4: 'fourth'}.get(lanes_used, '%d-th' % lanes_used)))
        elif fused_active:
            launch_ms = overlapped_ms
            launch_source = ('events on %d launches of the same stream of calls right after '
                             'the timed region (its %d steps carry none)' % (n_launch, args.steps))
        else:
            launch_ms = isolated_ms
            launch_source = ('kernels serialised (pipeline off) before the timed region: %d '
                             'launches with events' % n_launch)
        concurrency = launch_ms / (step_seconds * 1e3)
        # THE fraction: algorithmic flop of one step over the step time of the timed region
        achieved = flop_contract / step_seconds / 1e12
        total_draws = comm.world_size * n_draws * args.steps
        if interp_mode:
            workload = ('BASELINE configs[3]: Interpolator.predict() over a 5 x 5 grid of '
                        'synthetic auto tables (each 50 mass bins x {cen,sat}, G=100, P=5050, 19 '
                        'rp bins), Zheng07, n_gauss_prim=10; %d draws per step sharded '
                        'round-robin over %d GPU(s), draws resident in HBM, results gathered on '
                        'rank 0' % (n_draws * comm.world_size, comm.world_size))
        else:
            workload = ('BASELINE configs[1]: Zheng07 predict(), synthetic auto '
                        'table 50 mass bins x {cen,sat} (G=100, P=5050), 19 rp '
                        'bins, n_gauss_prim=10, batch of %d draws per GPU per '
                        'step, draws and results resident in HBM' % n_draws)
        result = {
            'metric': 'predict_calls_per_sec',
            'value': total_draws / elapsed,
            'value_definition': 'device-resident',
            'value_is': 'device-resident rate (draws and results in HBM; multi-GPU: results '
                        'gathered on rank 0); the host-to-host rate of SURVEY.md 8d is '
                        'value_host_to_host (= host_to_host_pipelined.value)',
            'unit': 'calls/s',
            'n_gpus': comm.world_size,
            'steps': args.steps,
            'warmup': args.warmup,
            'settle_steps': settle_steps,
            'timed_region_breakdown_us': breakdown,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'strong' if interp_mode else 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic',
            'config': {
                'workload': workload,
                'draws_per_gpu_per_step': n_draws,
                'n_bins': n_bins, 'n_pairs': n_pairs, 'n_r': N_R, 'n_tables': n_tables,
                'parallelism': 'draws sharded over %d GPU(s), table%s replicated' %
                               (comm.world_size, 's' if interp_mode else ''),
                'gather': comm.gather_backend,
                'rccl_ranks': comm.world_size if comm.comm is not None else 0,
                'gather_payload': 'ngal + chi2 (16 B per draw)' if chi2_mode
                                  else 'ngal + xi (%d B per draw)' % (8 * (1 + N_R)),
                'gather_payload_options': {
                    'chi2': '16 B per draw = %.2f MB per rank per gather' %
                            (16e-6 * n_draws * every),
                    'full': '%d B per draw = %.2f MB per rank per gather' %
                            (8 * (1 + N_R), 8e-6 * (1 + N_R) * n_draws * every)},
                'gather_every_steps': every,
                'lanes': lanes_used,
            },
            'roofline': {
                'kernel': kernel_name,
                'bound': 'mfma',
                'achieved': achieved,
                'peak': FP64_PEAK_TFLOPS,
                'unit': 'TFLOP/s',
                'frac': achieved / FP64_PEAK_TFLOPS,
                'frac_method': 'flop_per_launch x launches per step (1) / (elapsed / steps) / '
                               'peak, of the timed region: = flop_per_launch / ms_per_step / peak',
                'frac_by_duration': flop_contract / (launch_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                'frac_by_duration_method': 'flop_per_launch / mean_launch_ms / peak: by the raw '
                                           'duration of a launch (what rocprofv3 reports per '
                                           'dispatch), whatever else runs beside it',
                'traffic': headline_traffic[0],
                'traffic_source': '%s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
                                  'of this script, tools/profile_round.sh; FETCH_SIZE x 2 per '
                                  'the gfx950 correction)' % headline_traffic[1]
                                  if headline_traffic[1] else None,
                'flop_per_launch': flop_contract,
                'mean_launch_ms': launch_ms,
                'launch_ms_source': launch_source,
                'concurrent_launches': concurrency,
                'launches_timed': n_launch,
                'sample_after_region': {
                    'mean_launch_ms': overlapped_ms, 'wall_ms_per_launch': overlapped_wall_ms,
                    'concurrent_launches': sample_concurrency,
                    'frac_share_of_chip': flop_contract / (overlapped_wall_ms * 1e-3) / 1e12
                                          / FP64_PEAK_TFLOPS,
                    'note': '1000 launches of the same stream of calls with per-launch events, '
                            'right after the timed region (rounds 3-4 quoted this as frac)'},
                'alone': {
                    'mean_launch_ms': fused_alone_ms,
                    'frac': flop_contract / (fused_alone_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    'note': 'the same kernel with nothing else on the chip (pipeline off): '
                            '%d workgroups, one per CU on %d of the 256 CUs, the occupation '
                            'phase not overlapped by a neighbour\'s matrix phase -- not how it '
                            'is used; calls that run alone (synchronous host API) keep the '
                            'three kernels' % (launch[0].value, min(256, launch[0].value))}
                if fused_active else None,
                'three_kernel_path': {
                    'kernel': three_kernel_name,
                    'mean_launch_ms': isolated_ms,
                    'frac': flop_contract / (isolated_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                    'serialised_step_ms': serial_step_ms,
                    'note': 'contraction kernel of the three-kernel path (option fused=0, and '
                            'every call that runs alone on its lane), kernels serialised, '
                            'measured before the timed region'},
                'matrix_pipe_busy': matrix_pipe_busy(kernel_name, step_seconds)
                                    if fused_active else None,
                'workgroups': launch[0].value,
                'waves_per_workgroup': launch[1].value,
                'lds_bytes': launch[3].value,
            },
            'parity_max_rel_vs_oracle': parity,
            'device': _lib.device_name(),
        }
        if comm.rccl_error:
            result['config']['rccl_error'] = comm.rccl_error
        if gather_check is not None:
            result['gather_check'] = gather_check
        if second is not None:
            result['second_payload'] = second

    # ---- CPU baseline (the headline line needs it), then the secondary legs (1 GPU) ----------
    if result is not None and comm.world_size == 1 and args.cpu_seconds > 0 and not interp_mode:
        result['cpu_baseline'] = cpu_baseline(table, args.cpu_seconds)
        if cpu_all is not None:
            result['cpu_baseline']['all_cores'] = cpu_all

    def leg(name, measure):
        """A secondary measurement: its record (or its failure) goes to the sidecar; it never
        costs the headline line."""
        try:
            value = measure()
        except Exception as error:   # noqa: BLE001
            import traceback
            result.setdefault('failed_legs', {})[name] = '%s: %s' % (type(error).__name__, error)
            traceback.print_exc(file=sys.stderr)
            return None
        if isinstance(value, dict) and name is None:
            result.update(value)
        elif name is not None:
            result[name] = value
        return value

    if comm.world_size == 1 and not interp_mode and args.detail:
        ngal_host = np.empty(n_draws)
        xi_host = np.empty((n_draws, N_R))

        def host_to_host():
            def host_call():
                _lib.check(lib.tc_predict_zheng07_batch(
                    handle, _lib.as_double_p(theta), 5, n_draws, N_GAUSS, 0,
                    _lib.as_double_p(ngal_host), _lib.as_double_p(xi_host)))
            seconds = time_calls(host_call, seconds=0.5, warm=5)
            return {
                'value': n_draws / seconds, 'unit': 'calls/s', 'ms_per_call': seconds * 1e3,
                'what': 'tc_predict_zheng07_batch: %d draws in pageable host memory -> (ngal, '
                        'xi) in host memory, synchronous, PCIe included (SURVEY.md 8d)' % n_draws}
        leg('host_to_host', host_to_host)
        if leg(None, lambda: host_pipelined(lib, _lib, handle, table, n_draws, data_vector,
                                            precision)):
            result['value_host_to_host'] = result['host_to_host_pipelined']['value']

        def batch_sizes():
            # other batch sizes of the same table (an ensemble sampler's 10^2 ... 10^4 walkers
            # per step), device-resident, the form the library chooses and -- by option -- the
            # three kernels; first draws of every batch against the oracle
            sizes = {}
            from oracle import tabcorr_oracle as oracle
            for size in (256, 1024, 4096):
                row = {}
                for name, fused in (('chosen', 1), ('three_kernels', 0)):
                    _lib.check(lib.tc_table_set_option(handle, b'fused', fused))
                    seconds = sustained(
                        lambda: lib.tc_predict_zheng07_batch_device(
                            handle, d_theta, 5, size, N_GAUSS, 0, out_ptr(0), out_ptr(0, size)),
                        synchronize, seconds=0.15, warm_seconds=0.05)
                    row[name + '_us_per_call'] = seconds * 1e6
                    row[name + '_calls_per_sec'] = size / seconds
                    if fused:
                        shape = [ctypes.c_int() for _ in range(4)]
                        lib.tc_table_last_launch(handle, *[ctypes.byref(v) for v in shape])
                        row['chosen_form'] = (
                            'one launch: %d workgroups of %d waves'
                            % (shape[0].value, shape[1].value)
                            if shape[2].value == 0 else 'three kernels')
                        host = dev.download(out_ptr(0), size * (1 + N_R))
                        expect = oracle.predict_zheng07_batch(table, theta[:3])
                        row['parity_max_rel_vs_oracle'] = float(max(
                            np.max(np.abs(host[:3] / expect[0] - 1)),
                            np.max(np.abs(host[size:size + 3 * N_R].reshape(3, N_R)
                                          / expect[1] - 1))))
                sizes[str(size)] = row
            _lib.check(lib.tc_table_set_option(handle, b'fused', 1))
            return sizes
        leg('batch_sizes', batch_sizes)
        leg('unbatched_us', lambda: unbatched(make, table, synthetic, Interpolator))
        leg('tabulation', lambda: tabulation(args.cpu_seconds))
        if args.other_configs:
            leg('other_configs', lambda: other_configs(
                lib, _lib, make, synthetic, Interpolator, args.cpu_seconds))
    if result is not None:
        emit(result)

    dev.free_all()
    comm.barrier()
    comm.close()


if __name__ == '__main__':
    main()
