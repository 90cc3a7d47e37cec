"""Page-locked host arrays and pending results of the asynchronous API.

The asynchronous entry points of the C ABI (``tc_predict_zheng07_batch_async``
and friends, ``include/tabcorr_amd.h``) move draws and results over PCIe with
copy commands queued on the same stream as the kernels, which requires
page-locked (pinned) host memory.  ``pinned_empty`` returns a NumPy array on
such memory; ``pin`` page-locks an array the caller already owns (e.g. the
coordinate array of an ensemble sampler) in place.

This is what replaces the reference's usage pattern -- a Python loop of
``predict`` calls (``README.md:72-75``) -- when the sampler proposes a whole
ensemble per step: enqueue step ``k + 1`` while step ``k`` is on its way back.
"""

import ctypes
import threading

import numpy as np

from . import _lib


class _PinnedBlock:
    """Owner of one ``tc_host_alloc`` allocation, exposed through the array
    interface so that NumPy views keep it alive."""

    def __init__(self, nbytes):
        self.lib = _lib.load()
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(max(nbytes, 1))
        _lib.check(self.lib.tc_host_alloc(ctypes.byref(self.ptr), self.nbytes))
        self.__array_interface__ = {
            'shape': (self.nbytes, ), 'typestr': '|u1',
            'data': (self.ptr.value, False), 'version': 3}

    def __del__(self):
        ptr = getattr(self, 'ptr', None)
        if ptr is not None and ptr.value:
            try:
                self.lib.tc_host_free(ptr)
            except Exception:  # interpreter shutdown
                pass
            self.ptr = ctypes.c_void_p()


def pinned_empty(shape, dtype=np.float64):
    """Uninitialised C-contiguous array in page-locked host memory."""
    dtype = np.dtype(dtype)
    shape = (shape, ) if np.isscalar(shape) else tuple(shape)
    count = int(np.prod(shape, dtype=np.int64))
    block = _PinnedBlock(count * dtype.itemsize)
    flat = np.asarray(block)[:count * dtype.itemsize].view(dtype)
    return flat.reshape(shape)


def pinned_array(values, dtype=np.float64):
    """Copy of ``values`` in page-locked host memory."""
    values = np.asarray(values, dtype=dtype)
    out = pinned_empty(values.shape, dtype)
    out[...] = values
    return out


def is_pinned(array):
    """Whether the whole of ``array`` (C-contiguous) lies in memory the library
    knows to be page-locked."""
    if not isinstance(array, np.ndarray) or not array.flags.c_contiguous:
        return False
    flag = ctypes.c_int(0)
    _lib.check(_lib.load().tc_host_is_pinned(
        ctypes.c_void_p(array.ctypes.data), array.nbytes, ctypes.byref(flag)))
    return bool(flag.value)


class pin:
    """Context manager / handle that page-locks an existing C-contiguous array
    in place (``tc_host_register``) and unlocks it on ``close``."""

    def __init__(self, array):
        if not isinstance(array, np.ndarray) or not array.flags.c_contiguous:
            raise ValueError('pin() needs a C-contiguous numpy array.')
        self.array = array
        self.lib = _lib.load()
        _lib.check(self.lib.tc_host_register(
            ctypes.c_void_p(array.ctypes.data), array.nbytes))
        self.open = True

    def close(self):
        if self.open:
            self.open = False
            _lib.check(self.lib.tc_host_unregister(
                ctypes.c_void_p(self.array.ctypes.data)))

    def __enter__(self):
        return self.array

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Pool:
    """Free list of pinned blocks by size, so that a loop of asynchronous
    calls on pageable arrays does not allocate page-locked memory per call."""

    def __init__(self):
        self.free = {}
        self.lock = threading.RLock()     # (re-entrant: PendingPrediction.__del__ may run
                                          # inside take / give on the same thread)

    def take(self, shape):
        count = int(np.prod(shape, dtype=np.int64))
        with self.lock:
            blocks = self.free.get(count)
            flat = blocks.pop() if blocks else None
        if flat is None:
            flat = pinned_empty(count)
        return flat.reshape(shape)

    def give(self, array):
        flat = array.reshape(-1)
        with self.lock:
            blocks = self.free.setdefault(flat.size, [])
            if len(blocks) < 8:
                blocks.append(flat)


POOL = _Pool()


class PendingPrediction:
    """Result of ``predict_batch_async`` / ``chi2_batch_async``: ``wait()``
    blocks until the results are in host memory and returns them in the form
    the synchronous call returns; ``done()`` tests without blocking.

    Arrays passed by the caller (``out=``, pinned ``theta``) must not be
    touched before ``wait()`` has returned.
    """

    def __init__(self, owner, wait_call, query_call, ticket, inputs, outputs,
                 pooled_inputs, pooled_outputs, package):
        self._owner = owner            # keeps the device handle alive
        self._wait_call = wait_call
        self._query_call = query_call
        self.ticket = ticket
        self._inputs = inputs
        self._outputs = outputs
        self._pooled_inputs = pooled_inputs
        self._pooled_outputs = pooled_outputs
        self._package = package
        self._result = None
        self._waited = False

    def done(self):
        if self._waited:
            return True
        if self._query_call is None:
            return False
        flag = ctypes.c_int(0)
        with self._owner.lock:
            _lib.check(self._query_call(self._owner.handle, self.ticket,
                                        ctypes.byref(flag)))
        return bool(flag.value)

    def wait(self):
        if not self._waited:
            with self._owner.lock:
                _lib.check(self._wait_call(self._owner.handle, self.ticket))
            self._waited = True
            outputs = self._outputs
            if self._pooled_outputs:
                outputs = [np.array(o) for o in outputs]
                for o in self._outputs:
                    POOL.give(o)
            if self._pooled_inputs:
                for a in self._inputs:
                    POOL.give(a)
            self._inputs = self._outputs = None
            self._result = self._package(*outputs)
        return self._result

    def __del__(self):
        # buffers of an abandoned ticket must outlive the copies into them
        if not self._waited and self._outputs is not None:
            try:
                self.wait()
            except Exception:
                pass


def stage_inputs(arrays):
    """Pinned versions of the input arrays: the arrays themselves when they
    are already page-locked, else copies in pooled pinned blocks."""
    if all(is_pinned(a) for a in arrays):
        return list(arrays), False
    staged = []
    for a in arrays:
        block = POOL.take(a.shape)
        block[...] = a
        staged.append(block)
    return staged, True


def caller_outputs(shapes, out):
    """The caller's ``out`` arrays (C-contiguous float64 of the right sizes,
    pinned or not), reshaped to ``shapes``."""
    out = list(out)
    if len(out) != len(shapes):
        raise ValueError('out must hold {} arrays.'.format(len(shapes)))
    for array, shape in zip(out, shapes):
        if (not isinstance(array, np.ndarray) or array.dtype != np.float64 or
                not array.flags.c_contiguous or
                array.size != int(np.prod(shape, dtype=np.int64))):
            raise ValueError(
                'out arrays must be C-contiguous float64 with {} elements.'
                .format(int(np.prod(shape, dtype=np.int64))))
    return [a.reshape(shape) for a, shape in zip(out, shapes)]


def stage_outputs(shapes, out):
    """Output arrays of an asynchronous call: the caller's (``out``, which
    must be pinned float64 C-contiguous arrays of the right shapes) or pooled
    pinned blocks."""
    if out is None:
        return [POOL.take(shape) for shape in shapes], True
    arrays = caller_outputs(shapes, out)
    if not all(is_pinned(array) for array in arrays):
        raise ValueError('out arrays must be page-locked: allocate them '
                         'with tabcorr_amd.pinned_empty.')
    return arrays, False
