"""``Interpolator`` with the reference's class surface, on an MI355X.

Mirrors ``tabcorr/interpolator.py`` of johannesulf/TabCorr v1.2.0: a list of
``TabCorr`` instances tabulated on a regular grid of extra parameters (e.g.
``log_eta``, ``alpha_s``, ``alpha_c``) is interpolated with a tensor-product
not-a-knot cubic spline at ``model.param_dict[key]``.

For Zheng07-family models everything runs in one pass on the device: the
occupations once per class of identical halo tables, the spline weight of
every table for every draw, and one contraction that accumulates all tables
(the interpolation is linear in the per-table ``xi``; each table's weight is
divided by that table's total pair weight beforehand because ``xi`` is a
ratio).  ``predict_batch`` exposes the batched form; ``predict`` keeps the
reference's scalar signature.
"""

import ctypes
import threading

import numpy as np

from . import _lib
from . import pinned
from .galtable import GalTypeTable
from .models import device_spec
from .tabcorr import TabCorr, XI_KEYS, NGAL_KEYS, _flags, _unbatch

OUT_OF_RANGE = ('The x-coordinates are outside of the interpolation ' +
                'range and extrapolation is turned off.')


def _columns(param_dict_table):
    """Column names and values of a parameter table given as an astropy
    Table, a dict of columns or a NumPy structured array."""
    if hasattr(param_dict_table, 'colnames'):
        names = list(param_dict_table.colnames)
        return names, [np.asarray(param_dict_table[n], dtype=np.float64)
                       for n in names]
    if isinstance(param_dict_table, dict):
        names = list(param_dict_table.keys())
        return names, [np.asarray(param_dict_table[n], dtype=np.float64)
                       for n in names]
    array = np.asarray(param_dict_table)
    if array.dtype.names is None:
        raise TypeError('param_dict_table must have named columns.')
    names = list(array.dtype.names)
    return names, [np.asarray(array[n], dtype=np.float64) for n in names]


class _HandleLocks:
    """The lock of an interpolator handle: its own and those of its tables.

    ``tc_interp_*`` calls use the table handles they were created from (the
    quadrature and schedule caches, the fused-likelihood state and the kernel
    timer of the first table, the lanes of every class representative), so a
    call on the interpolator must exclude calls on any of its tables
    (``include/tabcorr_amd.h``).  The table locks are taken in a fixed global
    order, so that two interpolators sharing tables cannot deadlock."""

    def __init__(self, tables):
        unique = {id(t.lock): t.lock for t in tables}
        self._locks = [threading.RLock()] + [unique[k] for k in sorted(unique)]

    def __enter__(self):
        for lock in self._locks:
            lock.acquire()
        return self

    def __exit__(self, *exc):
        for lock in reversed(self._locks):
            lock.release()
        return False

    def acquire(self, blocking=True):
        taken = []
        for lock in self._locks:
            if not lock.acquire(blocking):
                for held in reversed(taken):
                    held.release()
                return False
            taken.append(lock)
        return True

    def release(self):
        for lock in reversed(self._locks):
            lock.release()


class _DeviceInterpolator:

    def __init__(self, interpolator):
        lib = _lib.load()
        devices = [t.to_device() for t in interpolator.tabcorr_list]
        handles = (ctypes.c_void_p * len(devices))(
            *[d.handle for d in devices])
        points = _lib.contiguous(interpolator.points)
        handle = ctypes.c_void_p()
        _lib.check(lib.tc_interp_create(
            ctypes.cast(handles, _lib.c_void_pp), len(devices),
            points.shape[1], _lib.as_double_p(points), ctypes.byref(handle)))
        self.handle = handle
        self.lib = lib
        self.tables = devices          # keep the table handles alive
        # one host thread at a time per handle (see tabcorr._DeviceTable),
        # the handles of the tables included
        self.lock = _HandleLocks(devices)
        # scratch of the un-batched predict(model) path (see
        # tabcorr._DeviceTable.predict_one)
        self._one_theta = np.zeros(16)
        self._one_x = np.zeros(points.shape[1])
        self._one_ngal = np.zeros(1)
        self._one_xi = np.zeros(devices[0].n_r)
        self._one_pointers = tuple(_lib.as_double_p(a) for a in (
            self._one_theta, self._one_x, self._one_ngal, self._one_xi))
        self._one_call = lib.tc_interp_predict_zheng07_batch

    def predict_one(self, theta, x, n_gauss_prim, flags):
        n_theta = len(theta)
        with self.lock:
            self._one_theta[:n_theta] = theta
            self._one_x[:] = x
            p_theta, p_x, p_ngal, p_xi = self._one_pointers
            status = self._one_call(self.handle, p_theta, n_theta, p_x, 1,
                                    n_gauss_prim, flags, p_ngal, p_xi)
            if status:
                _lib.check(status)
            return self._one_ngal[0], self._one_xi.copy()

    def __del__(self):
        handle = getattr(self, 'handle', None)
        if handle is not None and handle.value is not None:
            try:
                self.lib.tc_interp_destroy(handle)
            except Exception:
                pass
            self.handle = None


class Interpolator:
    """Interpolation of multiple `TabCorr` instances."""

    def __init__(self, tabcorr_list, param_dict_table):
        """
        Parameters
        ----------
        tabcorr_list : list of TabCorr
        param_dict_table : table
            Keys and values of the extra parameters of each instance, same
            length and order as ``tabcorr_list``.

        Raises
        ------
        ValueError
            If ``param_dict_table`` does not describe a grid
            (``tabcorr/interpolator.py:32-57``).
        """
        names, columns = _columns(param_dict_table)
        names_columns = [(n, c) for n, c in zip(names, columns)
                         if n != 'tabcorr_index']
        self.keys = [n for n, c in names_columns]
        if len(tabcorr_list) != len(names_columns[0][1]):
            raise ValueError("The number of TabCorr instances does not match" +
                             " the number of entries in 'param_dict_table'.")
        self.tabcorr_list = list(tabcorr_list)
        self.points = np.stack([c for n, c in names_columns], axis=-1)
        self.xp = [np.sort(np.unique(c)) for n, c in names_columns]
        for xp in self.xp:
            if len(xp) < 4:
                raise ValueError('Cannot perform spline interpolation with ' +
                                 'less than 4 values.')
        if (np.prod([len(xp) for xp in self.xp]) != len(self.points) or
                len(np.unique(self.points, axis=0)) != len(self.points)):
            raise ValueError(
                "The 'param_dict_table' does not describe a grid.")
        # Grid order (lexicographic in the keys), as the sorted table with its
        # tabcorr_index column in the reference (interpolator.py:59-61).
        self.order = np.lexsort(self.points.T[::-1])
        columns = {key: self.points[self.order, d]
                   for d, key in enumerate(self.keys)}
        columns['tabcorr_index'] = self.order.copy()
        # (a column table with ``colnames`` / ``[name]`` / ``len`` like the
        # astropy Table of the reference)
        self.param_dict_table = GalTypeTable(columns)
        self._device = None
        self._a = None
        self._checked = None       # signature of the last consistent model

    # -- I/O ------------------------------------------------------------------

    @classmethod
    def read(cls, fname):
        """Read an interpolator written by the reference or by `write`
        (``tabcorr/interpolator.py:72-96``)."""
        from . import io
        return io.read_interpolator(cls, fname)

    def write(self, fname, overwrite=False, max_args_size=1000000,
              matrix_dtype=np.float32):
        """Write in the reference's layout (``tabcorr/interpolator.py:98-122``:
        dataset ``param_dict_table`` plus one group ``tabcorr_{i}`` per
        instance)."""
        from . import io
        io.write_interpolator(self, fname, overwrite=overwrite,
                              max_args_size=max_args_size,
                              matrix_dtype=matrix_dtype)

    # -- device ----------------------------------------------------------------

    def to_device(self):
        if self._device is None:
            self._device = _DeviceInterpolator(self)
        return self._device

    def set_deterministic(self, level=True):
        """``TabCorr.set_deterministic`` for every table of the grid (the
        interpolator's calls read the option of the first one).  ``True``:
        batch-invariant results where a one-launch form serves the grid (mode
        cross with up to 128 rows, e.g. the reference's AbacusSummit
        interpolator); grids in mode auto run three kernels whose sums are cut
        where the batch size puts them -- there the option only refuses the
        measured dispatch.  Returns whether the first table reports a
        batch-invariant form."""
        device = self.to_device()
        invariant = [halotab.set_deterministic(level)
                     for halotab in self.tabcorr_list]
        del device
        return invariant[0]

    def _x_model(self, model):
        x = np.empty(len(self.keys))
        for i, key in enumerate(self.keys):
            try:
                x[i] = model.param_dict[key]
            except KeyError:
                raise ValueError(
                    'The key {} is not present in the parameter '.format(key) +
                    'dictionary of the model.')
        return x

    def _check_range(self, x, extrapolate):
        if extrapolate:
            return
        x = np.atleast_2d(x)
        for d, xp in enumerate(self.xp):
            # (NaN fails the test, as np.digitize puts it past the last node:
            # interpolator.py:318-326)
            if not np.all((x[:, d] >= xp[0]) & (x[:, d] <= xp[-1])):
                raise ValueError(OUT_OF_RANGE)

    # -- predict -------------------------------------------------------------------

    def predict(self, model, separate_gal_type=False, n_gauss_prim=10,
                extrapolate=False, check_consistency=True, **occ_kwargs):
        """Interpolate the predictions of the instances
        (``tabcorr/interpolator.py:124-216``).

        Raises
        ------
        ValueError
            If a key is missing from ``model.param_dict`` or a value lies
            outside the grid and ``extrapolate`` is False.
        """
        x = self._x_model(model)
        if check_consistency:
            # the checks of tabcorr.py:496-535 read a handful of model
            # attributes; they are repeated for every table only when those
            # attributes (or the tables) have changed since the last call
            components = model._input_model_dictionary
            signature = (
                id(self.tabcorr_list[0]), len(self.tabcorr_list),
                tuple(model.gal_types), model.redshift,
                tuple((components[name].prim_haloprop_key,
                       getattr(components[name], 'sec_haloprop_key', None))
                      for name in ('centrals_occupation',
                                   'satellites_occupation')))
            if signature != self._checked:
                for halotab in self.tabcorr_list:
                    halotab._check_consistency(model)
                self._checked = signature
        spec = None if occ_kwargs else device_spec(model)
        if spec is None:
            return self._predict_generic(model, x, separate_gal_type,
                                         n_gauss_prim, extrapolate,
                                         **occ_kwargs)
        if not separate_gal_type:
            self._check_range(x, extrapolate)
            ngal, xi = self.to_device().predict_one(
                spec.theta, x, n_gauss_prim,
                _flags(False, spec.modulate_with_cenocc, spec.assembias,
                       spec.family))
            return ngal, xi.reshape(self.tabcorr_list[0].tpcf_shape)
        return _unbatch(*self.predict_batch(
            spec.theta[np.newaxis], x[np.newaxis],
            separate_gal_type=separate_gal_type, n_gauss_prim=n_gauss_prim,
            extrapolate=extrapolate,
            modulate_with_cenocc=spec.modulate_with_cenocc,
            assembias=spec.assembias, family=spec.family))

    def predict_batch(self, theta, x, separate_gal_type=False,
                      n_gauss_prim=10, extrapolate=False,
                      modulate_with_cenocc=False, assembias=False,
                      family='zheng07', out=None):
        """`predict` for ``(n_draws, 5 | 7)`` Zheng07 parameters ``theta`` and
        ``(n_draws, n_dim)`` values ``x`` of the extra parameters (columns in
        the order of ``self.keys``).  ``out=(ngal, xi)``: page-locked arrays
        that receive the results (see `TabCorr.predict_batch`)."""
        if out is not None:
            return self.predict_batch_async(
                theta, x, separate_gal_type=separate_gal_type,
                n_gauss_prim=n_gauss_prim, extrapolate=extrapolate,
                modulate_with_cenocc=modulate_with_cenocc,
                assembias=assembias, family=family, out=out).wait()
        theta = _lib.contiguous(np.atleast_2d(theta))
        x = _lib.contiguous(np.atleast_2d(x))
        if x.shape != (len(theta), len(self.keys)):
            raise ValueError('x must have shape (n_draws, {}).'.format(
                len(self.keys)))
        self._check_range(x, extrapolate)
        device = self.to_device()
        table = device.tables[0]
        flags = _flags(separate_gal_type, modulate_with_cenocc, assembias,
                       family)
        n_draws = len(theta)
        n_comp = table.n_components if separate_gal_type else 1
        ngal = np.empty((n_draws, 2 if separate_gal_type else 1))
        xi = np.empty((n_draws, n_comp, table.n_r))
        with device.lock:
            _lib.check(device.lib.tc_interp_predict_zheng07_batch(
                device.handle, _lib.as_double_p(theta), theta.shape[1],
                _lib.as_double_p(x), n_draws, n_gauss_prim, flags,
                _lib.as_double_p(ngal), _lib.as_double_p(xi)))
        return self.tabcorr_list[0]._package(ngal, xi, separate_gal_type)

    def _async_inputs(self, theta, x, extrapolate):
        theta = _lib.contiguous(np.atleast_2d(theta))
        x = _lib.contiguous(np.atleast_2d(x))
        if x.shape != (len(theta), len(self.keys)):
            raise ValueError('x must have shape (n_draws, {}).'.format(
                len(self.keys)))
        self._check_range(x, extrapolate)
        return theta, x

    def predict_batch_async(self, theta, x, separate_gal_type=False,
                            n_gauss_prim=10, extrapolate=False,
                            modulate_with_cenocc=False, assembias=False,
                            family='zheng07', out=None):
        """`predict_batch` without waiting for the device (see
        `TabCorr.predict_batch_async`): returns a
        `tabcorr_amd.pinned.PendingPrediction`."""
        theta, x = self._async_inputs(theta, x, extrapolate)
        device = self.to_device()
        table = device.tables[0]
        flags = _flags(separate_gal_type, modulate_with_cenocc, assembias,
                       family)
        n_draws = len(theta)
        n_comp = table.n_components if separate_gal_type else 1
        shapes = [(n_draws, 2 if separate_gal_type else 1),
                  (n_draws, n_comp, table.n_r)]
        (ngal, xi), pooled_out = pinned.stage_outputs(shapes, out)
        (theta_p, x_p), pooled_in = pinned.stage_inputs([theta, x])
        ticket = ctypes.c_int64(-1)
        with device.lock:
            _lib.check(device.lib.tc_interp_predict_zheng07_batch_async(
                device.handle, _lib.as_double_p(theta_p), theta.shape[1],
                _lib.as_double_p(x_p), n_draws, n_gauss_prim, flags,
                _lib.as_double_p(ngal), _lib.as_double_p(xi),
                ctypes.byref(ticket)))
        first = self.tabcorr_list[0]
        return pinned.PendingPrediction(
            device, device.lib.tc_interp_wait, device.lib.tc_interp_query,
            ticket.value,
            [theta_p, x_p], [ngal, xi], pooled_in, pooled_out,
            lambda n, v: first._package(n, v, separate_gal_type))

    def chi2_batch_async(self, theta, x, data, precision, n_gauss_prim=10,
                         extrapolate=False, modulate_with_cenocc=False,
                         assembias=False, family='zheng07', out=None):
        """`chi2_batch` without waiting for the device."""
        theta, x = self._async_inputs(theta, x, extrapolate)
        device = self.to_device()
        n_r = device.tables[0].n_r
        data = _lib.contiguous(np.ravel(data))
        precision = _lib.contiguous(precision)
        if data.shape != (n_r, ) or precision.shape != (n_r, n_r):
            raise ValueError('data must have {0} entries and precision shape '
                             '({0}, {0}).'.format(n_r))
        n_draws = len(theta)
        (ngal, chi2), pooled_out = pinned.stage_outputs(
            [(n_draws, ), (n_draws, )], out)
        (theta_p, x_p), pooled_in = pinned.stage_inputs([theta, x])
        ticket = ctypes.c_int64(-1)
        with device.lock:
            _lib.check(device.lib.tc_interp_chi2_zheng07_batch_async(
                device.handle, _lib.as_double_p(theta_p), theta.shape[1],
                _lib.as_double_p(x_p), n_draws, n_gauss_prim,
                _flags(False, modulate_with_cenocc, assembias, family),
                _lib.as_double_p(data), _lib.as_double_p(precision),
                _lib.as_double_p(ngal), _lib.as_double_p(chi2),
                ctypes.byref(ticket)))
        return pinned.PendingPrediction(
            device, device.lib.tc_interp_wait, device.lib.tc_interp_query,
            ticket.value,
            [theta_p, x_p], [ngal, chi2], pooled_in, pooled_out,
            lambda n, c: (n, c))

    def chi2_batch(self, theta, x, data, precision, n_gauss_prim=10,
                   extrapolate=False, modulate_with_cenocc=False,
                   assembias=False, family='zheng07'):
        """Gaussian ``chi^2 = (xi - data)^T precision (xi - data)`` of every
        draw of `predict_batch`, evaluated on the device right behind the
        interpolated prediction (extension, as `TabCorr.chi2_batch`: the
        reference leaves the likelihood to the user, ``README.md:7``).

        Returns
        -------
        ngal, chi2 : numpy.ndarray ``(n_draws, )``
        """
        theta = _lib.contiguous(np.atleast_2d(theta))
        x = _lib.contiguous(np.atleast_2d(x))
        if x.shape != (len(theta), len(self.keys)):
            raise ValueError('x must have shape (n_draws, {}).'.format(
                len(self.keys)))
        self._check_range(x, extrapolate)
        device = self.to_device()
        n_r = device.tables[0].n_r
        data = _lib.contiguous(np.ravel(data))
        precision = _lib.contiguous(precision)
        if data.shape != (n_r, ) or precision.shape != (n_r, n_r):
            raise ValueError('data must have {0} entries and precision shape '
                             '({0}, {0}).'.format(n_r))
        ngal = np.empty(len(theta))
        chi2 = np.empty(len(theta))
        with device.lock:
            _lib.check(device.lib.tc_interp_chi2_zheng07_batch(
                device.handle, _lib.as_double_p(theta), theta.shape[1],
                _lib.as_double_p(x), len(theta), n_gauss_prim,
                _flags(False, modulate_with_cenocc, assembias, family),
                _lib.as_double_p(data), _lib.as_double_p(precision),
                _lib.as_double_p(ngal), _lib.as_double_p(chi2)))
        return ngal, chi2

    # -- generic models: host callbacks + device contraction per table -----------------

    def _spline_matrices(self):
        if self._a is None:
            lib = _lib.load()
            self._a = []
            for xp in self.xp:
                xp = _lib.contiguous(xp)
                a = np.zeros((len(xp) - 1, 4, len(xp)))
                _lib.check(lib.tc_spline_interpolation_matrix(
                    len(xp), _lib.as_double_p(xp), _lib.as_double_p(a)))
                self._a.append(a)
        return self._a

    def _predict_generic(self, model, x, separate_gal_type, n_gauss_prim,
                         extrapolate, **occ_kwargs):
        self._check_range(x, extrapolate)
        # occupations once per distinct halo table (interpolator.py:63-70,
        # 181-184)
        cache = []
        results = []
        for k in self.order:
            halotab = self.tabcorr_list[k]
            raw = halotab.gal_type.as_array()
            occupation = None
            for other, value in cache:
                if other.dtype == raw.dtype and np.array_equal(other, raw):
                    occupation = value
                    break
            if occupation is None:
                occupation = halotab._host_mean_occupation(
                    model, n_gauss_prim, **occ_kwargs)
                cache.append((raw, occupation))
            results.append(halotab.predict(
                occupation, separate_gal_type=separate_gal_type))
        shape = [len(xp) for xp in self.xp]
        a = self._spline_matrices()
        output = []
        for i in range(2):
            if separate_gal_type:
                output.append({})
                for key in results[0][i]:
                    data = np.array([r[i][key] for r in results])
                    data = data.reshape(shape + list(data.shape[1:]))
                    output[-1][key] = spline_interpolate(
                        x, self.xp, a, data, extrapolate=extrapolate)
            else:
                data = np.array([r[i] for r in results])
                data = data.reshape(shape + list(data.shape[1:]))
                output.append(spline_interpolate(
                    x, self.xp, a, data, extrapolate=extrapolate))
        return tuple(output)


def spline_interpolation_matrix(xp):
    """Matrix ``a`` of shape ``(n - 1, 4, n)`` such that
    ``np.einsum('ij,j,i', a[i], y, x0**np.arange(4))`` is the ``i``-th segment
    of the not-a-knot cubic spline through ``(xp, y)`` at ``x0``
    (``tabcorr/interpolator.py:219-272``; computed by the C library).

    Raises
    ------
    ValueError
        If ``xp`` has fewer than 4 entries.
    """
    xp = _lib.contiguous(xp)
    a = np.zeros((max(len(xp) - 1, 0), 4, len(xp)))
    _lib.check(_lib.load().tc_spline_interpolation_matrix(
        len(xp), _lib.as_double_p(xp), _lib.as_double_p(a)))
    return a


def spline_interpolate(x, xp, a, yp, extrapolate=False):
    """Evaluate the tensor-product spline along the first ``len(x)`` axes of
    ``yp`` (``tabcorr/interpolator.py:275-331``): segment search with
    ``np.digitize`` (right edge included), ``ValueError`` or clamping outside
    the grid, cubic polynomial per axis."""
    xp = xp if isinstance(xp, list) else [xp]
    a = a if isinstance(a, list) else [a]
    for value, matrix, nodes in zip(np.atleast_1d(x), a, xp):
        segment = int(np.searchsorted(nodes, value, side='right')) - 1
        if value == nodes[-1]:
            segment = len(nodes) - 2
        if not 0 <= segment <= len(nodes) - 2:
            if not extrapolate:
                raise ValueError(OUT_OF_RANGE)
            segment = min(max(segment, 0), len(nodes) - 2)
        weights = matrix[segment].T @ value**np.arange(4)
        yp = np.tensordot(weights, yp, axes=(0, 0))
    return yp
