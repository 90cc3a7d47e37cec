"""Sharding of parameter draws over the GPUs of one node.

The draws of an MCMC ensemble are independent, so the path shards without any
data-path collective: every rank holds a full replica
of the table (at most ~120 MB), evaluates the draws ``rank, rank + world,
rank + 2 world, ...`` (round-robin) and the per-rank results are collected on
the root by ONE gather.  One process per GPU, launched by ``torchrun`` /
``python -m torch.distributed.run``.

Control plane (rendezvous, barrier, scalar reductions, exchange of the RCCL
unique id): ``torch.distributed`` with the ``gloo`` backend on CPU tensors --
PyTorch never touches the GPU here.  Data plane: ``ncclGather`` of RCCL over
xGMI through the C ABI (``tc_comm_*``), on device buffers owned by this
library, for every sharded variant (single table or interpolator, total or
separated by galaxy type).  When RCCL is unavailable (e.g. the CPU-only
tests) the gather falls back to gloo on host arrays; the result is identical.
"""

import ctypes
import os

import numpy as np

from . import _lib


def round_robin_indices(n_draws, rank, world_size):
    """Indices of the draws rank ``rank`` evaluates."""
    return np.arange(rank, n_draws, world_size)


def shard_size(n_draws, world_size):
    """Per-rank buffer length (the same on every rank; short shards are padded
    by repeating their last draw so that the collective is regular)."""
    return (n_draws + world_size - 1) // world_size


def local_shard(theta, rank, world_size):
    """Rows of ``theta`` for this rank, padded to `shard_size` rows."""
    theta = np.asarray(theta)
    index = round_robin_indices(len(theta), rank, world_size)
    size = shard_size(len(theta), world_size)
    if len(index) == 0:
        # more ranks than draws: evaluate draw 0 and discard it on assembly
        index = np.zeros(1, dtype=int)
    index = np.concatenate([index, np.repeat(index[-1:], size - len(index))])
    return np.ascontiguousarray(theta[index])


def assemble(parts, n_draws):
    """Undo the round-robin split: ``parts[rank][k]`` is draw
    ``rank + k * world``.  ``parts`` is rank-major, each of `shard_size` rows.
    """
    world_size = len(parts)
    first = np.asarray(parts[0])
    out = np.empty((n_draws, ) + first.shape[1:], dtype=first.dtype)
    for rank, part in enumerate(parts):
        index = round_robin_indices(n_draws, rank, world_size)
        out[index] = np.asarray(part)[:len(index)]
    return out


class Communicator:
    """Process group of one-rank-per-GPU workers."""

    def __init__(self, rank=0, world_size=1, local_rank=0, use_rccl=True):
        self.rank = rank
        self.world_size = world_size
        self.local_rank = local_rank
        self.dist = None
        self.comm = None          # tc_comm handle (RCCL) or None
        self.rccl_error = None
        force = os.environ.get('TABCORR_AMD_FORCE_COMM', '0') == '1'
        if world_size > 1 or force:
            # The HIP library must be in the process before torch so that it
            # keeps the system ROCm runtime (see _lib.load).
            try:
                _lib.load()
            except _lib.TabCorrHipError:
                pass
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                dist.init_process_group('gloo', rank=rank,
                                        world_size=world_size)
            self.dist = dist
            if use_rccl:
                self._init_rccl()

    @classmethod
    def from_env(cls, use_rccl=True):
        return cls(int(os.environ.get('RANK', '0')),
                   int(os.environ.get('WORLD_SIZE', '1')),
                   int(os.environ.get('LOCAL_RANK', '0')), use_rccl)

    @property
    def is_root(self):
        return self.rank == 0

    @property
    def gather_backend(self):
        if self.dist is None:
            return 'none'
        return 'rccl' if self.comm is not None else 'gloo'

    def _init_rccl(self):
        """Create the RCCL communicator, or fall back to gloo on EVERY rank.

        Every local failure (library missing, no device, librccl not
        loadable) becomes this rank's ``ok = 0`` contribution to the first
        all_reduce, so that all ranks run the same sequence of gloo
        collectives and agree on the outcome -- a rank that cannot use RCCL
        must not leave the others waiting in ncclCommInitRank.
        """
        import torch
        lib = None
        error = None
        buffer = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        try:
            lib = _lib.load()
            if _lib.device_count() < 1:
                raise _lib.TabCorrHipError('no HIP device')
            # (more ranks than devices: round-robin -- RCCL then refuses two ranks on one
            # device below, and every rank falls back to gloo)
            _lib.check(lib.tc_set_device(self.local_rank % _lib.device_count()))
            _lib.check(lib.tc_comm_unique_id(buffer))
        except Exception as exc:   # noqa: BLE001 -- every failure is a vote
            # (also e.g. subprocess.CalledProcessError of an on-demand build:
            # whatever goes wrong here, this rank must still reach the
            # all_reduce below or the others would wait for it)
            error = '{}: {}'.format(type(exc).__name__, exc)
        ok = torch.tensor([0 if error else 1], dtype=torch.int32)
        self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
        unique = torch.zeros(_lib.UNIQUE_ID_BYTES, dtype=torch.uint8)
        if self.rank == 0 and error is None:
            unique = torch.frombuffer(
                bytearray(buffer.raw), dtype=torch.uint8).clone()
        self.dist.broadcast(unique, 0)
        if ok.item() == 0:
            self.rccl_error = error or 'RCCL is not usable on every rank'
            return
        handle = ctypes.c_void_p()
        raw = bytes(unique.numpy().tobytes())
        status = lib.tc_comm_create(raw, self.world_size, self.rank,
                                    ctypes.byref(handle))
        if status != _lib.TC_OK:
            error = lib.tc_last_error().decode(errors='replace')
        good = torch.tensor([1 if status == _lib.TC_OK else 0],
                            dtype=torch.int32)
        self.dist.all_reduce(good, op=self.dist.ReduceOp.MIN)
        if good.item() == 0:
            if status == _lib.TC_OK:
                lib.tc_comm_destroy(handle)
            self.rccl_error = ('ncclCommInitRank failed on some rank' +
                               (': ' + error if error else ''))
            return
        self.comm = handle

    # -- control plane -----------------------------------------------------

    def barrier(self):
        """All ranks: over RCCL when the communicator exists (one all-reduce of a double on
        its stream, tens of microseconds), else gloo (a TCP round trip, 0.2-0.3 ms -- a
        quarter of a 20-step timed region)."""
        if self.comm is not None:
            _lib.check(_lib.load().tc_comm_barrier(self.comm))
        elif self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        tensor = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.MAX)
        return tensor.item()

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        tensor = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM)
        return tensor.item()

    # -- data plane -----------------------------------------------------------

    def gather_host(self, array):
        """Gather equal-shaped host arrays on the root (list, rank-major);
        ``None`` elsewhere.  gloo; used when RCCL is not available."""
        array = np.ascontiguousarray(array)
        if self.dist is None:
            return [array]
        import torch
        tensor = torch.from_numpy(array)
        if self.is_root:
            parts = [torch.empty_like(tensor) for _ in range(self.world_size)]
            self.dist.gather(tensor, parts, dst=0)
            return [part.numpy() for part in parts]
        self.dist.gather(tensor, None, dst=0)
        return None

    def gather_device(self, table_handle, send_ptr, recv_ptr, count, slot=0,
                      interp_handle=None):
        """ncclGather of ``count`` doubles per rank from ``send_ptr`` into
        ``recv_ptr`` (root), queued behind the work on the table's (or the
        interpolator's) stream."""
        lib = _lib.load()
        if interp_handle is not None:
            _lib.check(lib.tc_comm_gather_interp(
                self.comm, interp_handle, send_ptr, recv_ptr, count, 0, slot))
        else:
            _lib.check(lib.tc_comm_gather(
                self.comm, table_handle, send_ptr, recv_ptr, count, 0, slot))

    def release(self, table_handle, slot=0, interp_handle=None):
        """Device-side wait of the producer's stream for the gather of
        ``slot``."""
        lib = _lib.load()
        if interp_handle is not None:
            _lib.check(lib.tc_comm_release_interp(self.comm, interp_handle,
                                                  slot))
        else:
            _lib.check(lib.tc_comm_release(self.comm, table_handle, slot))

    def synchronize(self):
        if self.comm is not None:
            _lib.check(_lib.load().tc_comm_synchronize(self.comm))

    def close(self):
        if self.comm is not None:
            _lib.load().tc_comm_destroy(self.comm)
            self.comm = None


def packed_count(n_local, n_ngal, n_comp, n_r):
    """Doubles one rank contributes to the gather: ``[ngal (n_local, n_ngal) |
    xi (n_local, n_comp, n_r)]``."""
    return n_local * (n_ngal + n_comp * n_r)


def unpack_gathered(flat, world_size, n_local, n_ngal, n_comp, n_r, n_draws):
    """Undo the gather of `packed_count` blocks: ``flat`` holds the blocks of
    all ranks, rank-major, as ``ncclGather`` delivers them; returns ``ngal
    (n_draws, n_ngal)`` and ``xi (n_draws, n_comp, n_r)`` in the caller's
    draw order (round-robin undone, padding rows dropped)."""
    count = packed_count(n_local, n_ngal, n_comp, n_r)
    flat = np.asarray(flat).reshape(world_size, count)
    ngal = assemble([part[:n_local * n_ngal].reshape(n_local, n_ngal)
                     for part in flat], n_draws)
    xi = assemble([part[n_local * n_ngal:].reshape(n_local, n_comp, n_r)
                   for part in flat], n_draws)
    return ngal, xi


class ResultRing:
    """Bookkeeping of a ring of per-step result slots that is drained to the
    root block by block (``bench.py --gpus N``; any producer that keeps the
    results of many steps on the device and gathers them in the background).

    The ring holds ``n_blocks`` blocks of ``every`` slots of ``n_out`` doubles.
    Step ``k`` writes slot ``k % (n_blocks * every)``.  When the last slot of
    a block has been written the block is gathered (``gather`` -- injectable:
    RCCL on device buffers in production, gloo on host arrays in the CPU
    tests); before the first slot of a block is written again the producer
    waits for that block's previous gather (``release``).  `flush` gathers a
    trailing, partly filled block.

    ``gather(block, send_offset, recv_offset, count)`` must collect ``count``
    doubles starting at element ``send_offset`` of every rank's ring into the
    root's receive buffer at ``recv_offset``, rank-major (rank ``r`` at
    ``recv_offset + r * count``): the layout of ``ncclGather``.  The receive
    buffer has ``n_blocks * world_size * every * n_out`` elements; block ``b``
    owns ``[b, b + 1) * world_size * every * n_out``.

    Every gather is recorded in ``self.log`` as ``(block, first_step,
    n_steps, recv_offset, count)`` so that a consumer on the root knows which
    steps a region holds (`steps_in`).
    """

    def __init__(self, n_out, every, world_size, gather, release=None,
                 n_blocks=4):
        if every < 1 or n_blocks < 1:
            raise ValueError('every and n_blocks must be positive.')
        self.n_out = n_out
        self.every = every
        self.world_size = world_size
        self.n_blocks = n_blocks
        self.n_slots = n_blocks * every
        self._gather = gather
        self._release = release
        self.log = []
        self._gathers_of_block = [0] * n_blocks

    @property
    def ring_elements(self):
        return self.n_slots * self.n_out

    @property
    def recv_elements(self):
        return self.n_blocks * self.world_size * self.every * self.n_out

    def slot(self, index):
        return index % self.n_slots

    def slot_offset(self, index):
        """Element offset of the slot step ``index`` writes."""
        return self.slot(index) * self.n_out

    def recv_offset(self, block):
        return block * self.world_size * self.every * self.n_out

    def _gather_block(self, block, first_step, n_steps):
        count = n_steps * self.n_out
        self._gather(block, block * self.every * self.n_out,
                     self.recv_offset(block), count)
        self._gathers_of_block[block] += 1
        self.log.append((block, first_step, n_steps, self.recv_offset(block),
                         count))

    def before_step(self, index):
        """Call before the producer of step ``index`` is queued."""
        slot = self.slot(index)
        block = slot // self.every
        if (slot % self.every == 0 and self._release is not None and
                self._gathers_of_block[block] > 0):
            # the block's previous gather is done before it is overwritten
            self._release(block)

    def after_step(self, index):
        """Call after the producer of step ``index`` is queued."""
        slot = self.slot(index)
        if slot % self.every == self.every - 1:
            self._gather_block(slot // self.every, index - self.every + 1,
                               self.every)

    def flush(self, n_steps):
        """Gather the trailing, partly filled block after ``n_steps`` steps
        (counted from a multiple of the ring size)."""
        rest = n_steps % self.every
        if rest:
            self._gather_block(self.slot(n_steps - 1) // self.every,
                               n_steps - rest, rest)

    def steps_in(self, entry, rank):
        """For a log entry: ``[(step, offset)]`` of rank ``rank``'s steps inside
        the root's receive buffer."""
        block, first_step, n_steps, recv_offset, count = entry
        return [(first_step + k, recv_offset + rank * count + k * self.n_out)
                for k in range(n_steps)]


class _DeviceArray:
    """A device allocation of doubles owned through the C ABI."""

    def __init__(self, count):
        self.lib = _lib.load()
        self.ptr = ctypes.c_void_p()
        self.count = count
        _lib.check(self.lib.tc_device_malloc(ctypes.byref(self.ptr),
                                             max(count, 1) * 8))

    def upload(self, array):
        array = _lib.contiguous(array)
        _lib.check(self.lib.tc_memcpy_h2d(
            self.ptr, array.ctypes.data_as(ctypes.c_void_p), array.nbytes))

    def download(self, count=None):
        out = np.empty(self.count if count is None else count)
        _lib.check(self.lib.tc_memcpy_d2h(
            out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
        return out

    def offset(self, count):
        return ctypes.c_void_p(self.ptr.value + count * 8)

    def __del__(self):
        if getattr(self, 'ptr', None) is not None and self.ptr.value:
            try:
                self.lib.tc_device_free(self.ptr)
            except Exception:
                pass
            self.ptr = ctypes.c_void_p()


def _predict_sharded_rccl(predictor, theta, communicator, x=None,
                          separate_gal_type=False, n_gauss_prim=10,
                          modulate_with_cenocc=False, assembias=False,
                          extrapolate=False, family='zheng07'):
    """Device buffers end to end, one ``ncclGather``: every rank evaluates its
    round-robin share (``TabCorr`` or, with ``x``, ``Interpolator``; total or
    per-galaxy-type prediction) into ``[ngal | xi]`` on its GPU, RCCL collects
    the blocks on the root, the root downloads once and undoes the
    round-robin split."""
    from .tabcorr import _flags
    n_draws = len(theta)
    rank, world = communicator.rank, communicator.world_size
    shard = local_shard(theta, rank, world)
    device = predictor.to_device()
    lib = device.lib
    if x is not None:
        predictor._check_range(x, extrapolate)     # same outcome on every rank
        x_shard = local_shard(x, rank, world)
        table = device.tables[0]
        first = predictor.tabcorr_list[0]
    else:
        table = device
        first = predictor
    n_local, n_r = len(shard), table.n_r
    n_ngal = 2 if separate_gal_type else 1
    n_comp = table.n_components if separate_gal_type else 1
    count = packed_count(n_local, n_ngal, n_comp, n_r)
    flags = _flags(separate_gal_type, modulate_with_cenocc, assembias, family)
    d_theta = _DeviceArray(shard.size)
    d_theta.upload(shard)
    d_out = _DeviceArray(count)
    d_recv = _DeviceArray(count * world if communicator.is_root else 0)
    if x is not None:
        d_x = _DeviceArray(x_shard.size)
        d_x.upload(x_shard)
        _lib.check(lib.tc_interp_predict_zheng07_batch_device(
            device.handle, d_theta.ptr, shard.shape[1], d_x.ptr, n_local,
            n_gauss_prim, flags, d_out.ptr, d_out.offset(n_local * n_ngal)))
        communicator.gather_device(
            None, d_out.ptr, d_recv.ptr if communicator.is_root else None,
            count, 0, interp_handle=device.handle)
    else:
        _lib.check(lib.tc_predict_zheng07_batch_device(
            device.handle, d_theta.ptr, shard.shape[1], n_local, n_gauss_prim,
            flags, d_out.ptr, d_out.offset(n_local * n_ngal)))
        communicator.gather_device(
            device.handle, d_out.ptr,
            d_recv.ptr if communicator.is_root else None, count, 0)
    communicator.synchronize()
    if not communicator.is_root:
        return None
    ngal, xi = unpack_gathered(d_recv.download(), world, n_local, n_ngal,
                               n_comp, n_r, n_draws)
    return first._package(ngal, xi, separate_gal_type)


def predict_batch_sharded(halotab, theta, communicator, x=None, **kwargs):
    """``TabCorr.predict_batch`` (or ``Interpolator.predict_batch`` when the extra
    parameters ``x`` are given) with the draws sharded round-robin over the
    ranks of ``communicator``.  Every rank passes the same ``theta`` (and
    ``x``); the root returns the assembled ``(ngal, xi)`` (or dicts), other
    ranks ``None``.  With an RCCL communicator the results of every variant
    (single table or interpolator, total or separated by galaxy type) stay on
    the devices until one ``ncclGather`` has collected them on the root; without
    RCCL (CPU-only tests, RCCL unusable) host arrays are gathered over gloo.
    """
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    n_draws = len(theta)
    if x is not None:
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
    if communicator.comm is not None and hasattr(halotab, 'to_device'):
        return _predict_sharded_rccl(halotab, theta, communicator, x=x,
                                     **kwargs)
    shard = local_shard(theta, communicator.rank, communicator.world_size)
    if x is not None:
        x_shard = local_shard(x, communicator.rank, communicator.world_size)
        ngal, xi = halotab.predict_batch(shard, x_shard, **kwargs)
    else:
        ngal, xi = halotab.predict_batch(shard, **kwargs)
    if isinstance(ngal, dict):
        keys_n, keys_x = list(ngal.keys()), list(xi.keys())
        packed = np.concatenate(
            [np.stack([ngal[k] for k in keys_n], axis=1)] +
            [xi[k].reshape(len(shard), -1) for k in keys_x], axis=1)
    else:
        packed = np.concatenate([ngal[:, np.newaxis],
                                 xi.reshape(len(shard), -1)], axis=1)
    parts = communicator.gather_host(packed)
    if parts is None:
        return None
    full = assemble(parts, n_draws)
    tpcf_shape = (halotab.tpcf_shape if hasattr(halotab, 'tpcf_shape') else
                  halotab.tabcorr_list[0].tpcf_shape)
    shape = (n_draws, ) + tuple(tpcf_shape)
    if isinstance(ngal, dict):
        n_r = int(np.prod(tpcf_shape))
        ngal_out = {k: full[:, i] for i, k in enumerate(keys_n)}
        xi_out = {}
        for i, k in enumerate(keys_x):
            lo = len(keys_n) + i * n_r
            xi_out[k] = full[:, lo:lo + n_r].reshape(shape)
        return ngal_out, xi_out
    return full[:, 0], full[:, 1:].reshape(shape)


def chi2_batch_sharded(predictor, theta, data, precision, communicator, x=None,
                       n_gauss_prim=10, modulate_with_cenocc=False,
                       assembias=False, extrapolate=False, family='zheng07'):
    """``chi2_batch`` of a ``TabCorr`` (or, with ``x``, an ``Interpolator``) with
    the draws sharded round-robin over the ranks: every rank evaluates
    ``(ngal, chi2)`` of its share on its GPU and 16 bytes per draw are gathered
    on the root (one ``ncclGather`` with an RCCL communicator, gloo otherwise).
    Every rank passes the same arguments; the root returns ``(ngal, chi2)``,
    each ``(n_draws, )``, the other ranks ``None``."""
    from .tabcorr import _flags
    theta = np.atleast_2d(np.asarray(theta, dtype=np.float64))
    n_draws = len(theta)
    rank, world = communicator.rank, communicator.world_size
    shard = local_shard(theta, rank, world)
    kwargs = dict(n_gauss_prim=n_gauss_prim, modulate_with_cenocc=modulate_with_cenocc,
                  assembias=assembias, family=family)
    if x is not None:
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        x_shard = local_shard(x, rank, world)
    if communicator.comm is None or not hasattr(predictor, 'to_device'):
        if x is not None:
            ngal, chi2 = predictor.chi2_batch(shard, x_shard, data, precision,
                                              extrapolate=extrapolate, **kwargs)
        else:
            ngal, chi2 = predictor.chi2_batch(shard, data, precision, **kwargs)
        parts = communicator.gather_host(np.stack([ngal, chi2], axis=1))
        if parts is None:
            return None
        full = assemble(parts, n_draws)
        return full[:, 0], full[:, 1]
    device = predictor.to_device()
    lib = device.lib
    n_r = (device.tables[0] if x is not None else device).n_r
    data = _lib.contiguous(np.ravel(data))
    precision = _lib.contiguous(precision)
    if data.shape != (n_r, ) or precision.shape != (n_r, n_r):
        raise ValueError('data must have {0} entries and precision shape '
                         '({0}, {0}).'.format(n_r))
    if x is not None:
        predictor._check_range(x, extrapolate)     # same outcome on every rank
    n_local = len(shard)
    count = 2 * n_local
    flags = _flags(False, modulate_with_cenocc, assembias, family)
    d_theta = _DeviceArray(shard.size)
    d_theta.upload(shard)
    d_out = _DeviceArray(count)
    d_recv = _DeviceArray(count * world if communicator.is_root else 0)
    if x is not None:
        d_x = _DeviceArray(x_shard.size)
        d_x.upload(x_shard)
        _lib.check(lib.tc_interp_chi2_zheng07_batch_device(
            device.handle, d_theta.ptr, shard.shape[1], d_x.ptr, n_local,
            n_gauss_prim, flags, _lib.as_double_p(data),
            _lib.as_double_p(precision), d_out.ptr, d_out.offset(n_local)))
        communicator.gather_device(
            None, d_out.ptr, d_recv.ptr if communicator.is_root else None,
            count, 0, interp_handle=device.handle)
    else:
        _lib.check(lib.tc_chi2_zheng07_batch_device(
            device.handle, d_theta.ptr, shard.shape[1], n_local, n_gauss_prim,
            flags, _lib.as_double_p(data), _lib.as_double_p(precision),
            d_out.ptr, d_out.offset(n_local)))
        communicator.gather_device(
            device.handle, d_out.ptr,
            d_recv.ptr if communicator.is_root else None, count, 0)
    communicator.synchronize()
    if not communicator.is_root:
        return None
    flat = d_recv.download().reshape(world, 2, n_local)
    return (assemble([part[0] for part in flat], n_draws),
            assemble([part[1] for part in flat], n_draws))

