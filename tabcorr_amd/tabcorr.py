"""``TabCorr`` with the reference's class surface, computing on an MI355X.

The public methods mirror ``tabcorr/tabcorr.py`` of johannesulf/TabCorr v1.2.0
(``tabulate`` / ``read`` / ``write`` / ``mean_occupation`` / ``predict``, same
arguments, return types and exception types) so that an existing likelihood
can switch imports.  The arithmetic of ``predict`` runs in hand-written HIP
kernels behind a C ABI (``include/tabcorr_amd.h``); there is no CPU fallback.

Extensions beyond the reference (the reason to use a GPU at all):

``predict_batch(theta)``
    thousands of Zheng07 parameter vectors against the resident table in one
    call -- the reference's usage pattern is a Python loop of ``predict``
    calls (``README.md:72-75``).
``predict(ndarray)`` also accepts a 2-D ``(n_draws, n_bins)`` array of mean
    occupations (batched form of the operator seam at
    ``tabcorr/tabcorr.py:616-621``).
"""

import ctypes
import sys
import threading

import numpy as np

from . import _lib
from . import pinned
from .galtable import GalTypeTable
from .models import device_spec

ATTR_KEYS = ['tpcf', 'mode', 'simname', 'redshift', 'Num_ptcl_requirement',
             'prim_haloprop_key', 'sec_haloprop_key']

XI_KEYS = {'auto': ['centrals-centrals', 'centrals-satellites',
                    'satellites-satellites'],
           'cross': ['centrals', 'satellites']}
NGAL_KEYS = ['centrals', 'satellites']


class _DeviceTable:
    """Owner of a ``tc_table`` handle."""

    def __init__(self, halotab, compute_dtype):
        lib = _lib.load()
        _lib.require_device()
        gal_type = halotab.gal_type
        mode = halotab.attrs['mode']
        if mode not in _lib.MODE:
            raise ValueError("Unknown mode '{}'.".format(mode))
        matrix = np.asarray(halotab.tpcf_matrix)
        if matrix.ndim != 2:
            raise ValueError('tpcf_matrix must be two-dimensional.')
        if matrix.dtype == np.float32:
            matrix = np.ascontiguousarray(matrix)
            matrix_dtype = _lib.DTYPE_F32
        else:
            matrix = _lib.contiguous(matrix)
            matrix_dtype = _lib.DTYPE_F64
        columns = {}
        for name in ['n_h', 'log_prim_haloprop_min', 'log_prim_haloprop_max',
                     'sec_haloprop_percentile']:
            columns[name] = _lib.contiguous(gal_type[name])
        if 'prim_haloprop_dist_index' in gal_type.colnames:
            dist_index = _lib.contiguous(gal_type['prim_haloprop_dist_index'])
            dist_index_p = _lib.as_double_p(dist_index)
        else:
            # Tables of earlier TabCorr versions (tabcorr/tabcorr.py:571-574).
            dist_index_p = None
        is_central = np.ascontiguousarray(gal_type.is_centrals(),
                                          dtype=np.uint8)
        handle = ctypes.c_void_p()
        _lib.check(lib.tc_table_create(
            _lib.MODE[mode], len(gal_type), matrix.shape[0], matrix.shape[1],
            matrix.ctypes.data_as(ctypes.c_void_p), matrix_dtype,
            _lib.as_double_p(columns['n_h']),
            _lib.as_double_p(columns['log_prim_haloprop_min']),
            _lib.as_double_p(columns['log_prim_haloprop_max']),
            _lib.as_double_p(columns['sec_haloprop_percentile']),
            dist_index_p, is_central.ctypes.data_as(_lib.c_uint8_p),
            _lib.DTYPE_F32 if compute_dtype == 'float32' else _lib.DTYPE_F64,
            ctypes.byref(handle)))
        self.handle = handle
        self.lib = lib
        self.n_bins = len(gal_type)
        self.n_r = matrix.shape[0]
        self.n_components = 3 if mode == 'auto' else 2
        self.compute_dtype = compute_dtype
        # A handle serves one host thread at a time (include/tabcorr_amd.h);
        # ctypes releases the GIL during a call, so every call on this handle
        # -- and the scratch arrays of predict_one -- sits behind this lock
        # (re-entrant: a pending asynchronous call finalised by the garbage
        # collector inside a locked region waits on the same thread).
        self.lock = threading.RLock()
        # scratch of the un-batched predict(model) path: one draw in, one
        # (ngal, xi) out, with the ctypes pointers made once
        self._one_theta = np.zeros(16)
        self._one_ngal = np.zeros(1)
        self._one_xi = np.zeros(self.n_r)
        self._one_pointers = (_lib.as_double_p(self._one_theta),
                              _lib.as_double_p(self._one_ngal),
                              _lib.as_double_p(self._one_xi))
        self._one_call = lib.tc_predict_zheng07_batch

    def predict_one(self, theta, n_gauss_prim, flags):
        """Total prediction of ONE parameter vector (the reference's usage:
        one ``predict`` per MCMC step, ``README.md:72-75``) with as little
        Python between the caller and the C ABI as possible."""
        n_theta = len(theta)
        with self.lock:
            self._one_theta[:n_theta] = theta
            p_theta, p_ngal, p_xi = self._one_pointers
            status = self._one_call(self.handle, p_theta, n_theta, 1,
                                    n_gauss_prim, flags, p_ngal, p_xi)
            if status:
                _lib.check(status)
            return self._one_ngal[0], self._one_xi.copy()

    def __del__(self):
        handle = getattr(self, 'handle', None)
        if handle is not None and handle.value is not None:
            try:
                self.lib.tc_table_destroy(handle)
            except Exception:  # interpreter shutdown
                pass
            self.handle = None


class TabCorr:
    """Tabulated halo correlation functions convolved with an HOD on a GPU."""

    def __init__(self):
        self.attrs = {}
        self.tpcf_args = ()
        self.tpcf_kwargs = {}
        self.tpcf_shape = None
        self.tpcf_matrix = None
        self._gal_type = None
        self._device = None
        self._checked_model = None
        self.compute_dtype = 'float64'

    # -- construction -------------------------------------------------------

    @property
    def gal_type(self):
        return self._gal_type

    @gal_type.setter
    def gal_type(self, value):
        self._gal_type = None if value is None else (
            value if isinstance(value, GalTypeTable) else GalTypeTable(value))
        self._device = None

    @classmethod
    def from_arrays(cls, gal_type, tpcf_matrix, tpcf_shape, attrs,
                    tpcf_args=(), tpcf_kwargs=None, compute_dtype='float64'):
        """Build an instance from in-memory arrays (e.g. `synthetic_table`).
        """
        halotab = cls()
        halotab.gal_type = gal_type
        halotab.tpcf_matrix = np.asarray(tpcf_matrix)
        halotab.tpcf_shape = tuple(int(s) for s in tpcf_shape)
        halotab.attrs = dict(attrs)
        halotab.tpcf_args = tuple(tpcf_args)
        halotab.tpcf_kwargs = dict(tpcf_kwargs or {})
        halotab.compute_dtype = compute_dtype
        return halotab

    @classmethod
    def tabulate(cls, halocat, tpcf, *tpcf_args, **kwargs):
        """Tabulate halo correlation functions (``tabcorr/tabcorr.py:23-372``).

        Populating the mock and binning the halos is halotools-bound and
        delegated to the reference package when that (and halotools) is
        installed; with ``tpcf=tabcorr_amd.corrfunc.wp``, ``s_mu_tpcf`` or
        ``mean_delta_sigma`` the pair counting inside it (``compute_tpcf_matrix``,
        ``tabcorr/tabcorr.py:846-922``) runs on the GPU, all bin pairs in one
        pass.
        """
        try:
            import tabcorr as reference
        except ImportError as error:
            raise NotImplementedError(
                'TabCorr.tabulate needs the reference `tabcorr` package and '
                'halotools; tabcorr_amd accelerates predict() only.'
            ) from error
        from . import corrfunc
        if tpcf in (corrfunc.wp, corrfunc.s_mu_tpcf,
                    corrfunc.mean_delta_sigma):
            # the pair counts of ALL bin pairs in one pass on the GPU instead of
            # one two-point function call per pair of bins from a process pool
            # (tabcorr/tabcorr.py:846-922; a pool would also fork a process
            # that has initialised the GPU)
            module = sys.modules[reference.TabCorr.__module__]
            original = module.compute_tpcf_matrix
            module.compute_tpcf_matrix = corrfunc.reference_compute_tpcf_matrix
            try:
                tabulated = reference.TabCorr.tabulate(
                    halocat, tpcf, *tpcf_args, **kwargs)
            finally:
                module.compute_tpcf_matrix = original
        else:
            tabulated = reference.TabCorr.tabulate(
                halocat, tpcf, *tpcf_args, **kwargs)
        return cls.from_arrays(
            tabulated.gal_type, tabulated.tpcf_matrix, tabulated.tpcf_shape,
            tabulated.attrs, tabulated.tpcf_args, tabulated.tpcf_kwargs)

    @classmethod
    def read(cls, fname):
        """Read a table written by the reference (HDF5 layout of
        ``tabcorr/tabcorr.py:438-463``) or by `write`.

        Parameters
        ----------
        fname : str or h5py.Group
        """
        from . import io
        return io.read_tabcorr(cls, fname)

    def write(self, fname, overwrite=False, max_args_size=1000000,
              matrix_dtype=np.float32):
        """Write the table in the reference's HDF5 layout
        (``tabcorr/tabcorr.py:418-463``)."""
        from . import io
        io.write_tabcorr(self, fname, overwrite=overwrite,
                         max_args_size=max_args_size,
                         matrix_dtype=matrix_dtype)

    # -- device residency ----------------------------------------------------

    def to_device(self, compute_dtype=None):
        """Upload the table (done lazily by the first prediction).  The table
        must not be mutated afterwards; call `invalidate` if it is."""
        if compute_dtype is not None and compute_dtype != self.compute_dtype:
            self.compute_dtype = compute_dtype
            self._device = None
        if self._device is None:
            if self.compute_dtype not in ('float64', 'float32'):
                raise ValueError('compute_dtype must be float64 or float32.')
            self._device = _DeviceTable(self, self.compute_dtype)
        return self._device

    def autotune(self, separate_gal_type=False, modulate_with_cenocc=False,
                 assembias=False, family='zheng07', measure=True):
        """Measure which form of the batched path -- three kernels, or one
        launch with 64-draw / 32-draw workgroups -- serves this table fastest
        at batch sizes 256 ... 65536, and let the pipelined and asynchronous
        calls with these options take the measured form (about half a second;
        ``tc_table_set_option "autotune"``).  Without it the library estimates
        the crossovers from the table's shape.  (Option ``"autotune_after"``
        = N makes the N-th pipelined or asynchronous call with one combination
        of options run this measurement by itself; off by default, because
        that call then blocks for half a second and the measured choice --
        hence the last bits of the results -- may differ from run to run.)
        ``measure=False`` only reads the stored result (``None`` if there is
        none yet).

        Returns
        -------
        result : dict
            ``sizes`` (draws per call), ``forms`` (0: three kernels, 64 / 32:
            draws per workgroup of the one-launch form) and ``us_per_call``
            ``(n_sizes, 3)`` for (three kernels, 64 draws, 32 draws; 0 where
            a form is not available).
        """
        device = self.to_device()
        flags = _flags(separate_gal_type, modulate_with_cenocc, assembias,
                       family)
        sizes = np.zeros(16, dtype=np.int64)
        forms = np.zeros(16, dtype=np.int32)
        us = np.zeros((16, 3), dtype=np.float32)
        with device.lock:
            if measure:
                _lib.check(device.lib.tc_table_set_option(
                    device.handle, b'autotune', flags))
            count = ctypes.c_int(0)
            status = device.lib.tc_table_autotune_result(
                device.handle, flags, 16, ctypes.byref(count),
                sizes.ctypes.data_as(_lib.c_int64_p),
                forms.ctypes.data_as(_lib.c_int_p),
                us.ctypes.data_as(_lib.c_float_p))
            if status != 0 and not measure:
                return None
            _lib.check(status)
        count = count.value
        return {'sizes': sizes[:count].copy(), 'forms': forms[:count].copy(),
                'us_per_call': us[:count].astype(float)}

    def invalidate(self):
        self._device = None

    def set_resident(self, enabled=True, idle_us=None):
        """Serve un-batched ``predict(model)`` calls (one per MCMC step,
        ``README.md:72-75``) from ONE resident kernel launch instead of one
        launch per call: the launch path is 10 of the 15 us of such a call.
        The kernel takes every draw from a mailbox in page-locked memory and
        leaves by itself when no call has arrived for ``idle_us`` microseconds
        (default 2000; the next call launches it again) or when any other kind
        of call is made on this table.  Results are bit-identical to the
        one-launch-per-call path.

        Ensembles are served the same way: ``predict_batch`` with 24 to 256
        draws in host arrays (fewer: one launch is faster; option
        ``"resident_min_walkers"``) goes to a second resident kernel (one workgroup
        per CU: the walkers' occupations, slices of the table and the rows of
        the results in three phases that hand their data on through device
        memory) -- 64 walkers 34 -> 25 us, 256 walkers 45 -> 33 us per call on
        the reference's 60-bin table; a walker's result does not depend on the
        size of the ensemble or its place in it, and equals the batched path's
        to rounding (1e-14).

        ``enabled='auto'`` is the default state of every table: a loop of
        un-batched ``predict(model)`` calls is moved to the resident kernel by
        the library itself from the eighth call on that follows its
        predecessor within 300 us, with an idle time of 250 us (a
        ``hipDeviceSynchronize`` of the caller never waits longer for it); a
        caller whose pauses or device-wide synchronisations keep ending the
        launch is served by one launch per call again.  ``False`` switches
        that off as well."""
        device = self.to_device()
        value = 2 if enabled == 'auto' else (1 if enabled else 0)
        with device.lock:
            if idle_us is not None:
                _lib.check(device.lib.tc_table_set_option(
                    device.handle, b'resident_idle_us', int(idle_us)))
            _lib.check(device.lib.tc_table_set_option(
                device.handle, b'resident', value))

    def set_deterministic(self, level=True):
        """Reproducible bits on request (``tc_table_set_option
        "deterministic"``).  Every kernel form agrees with the reference to
        ~1e-14, but forms differ in their last bits, and by default the library
        picks the fastest form per call -- from the table, the options, the
        entry point and the batch size; never from timing, so a fixed-seed
        chain is reproducible from run to run as it is.

        ``level=True`` (2) makes the results *batch-invariant*: one kernel
        form per table and combination of options for ``predict(model)``,
        ``predict_batch`` of any size and the asynchronous calls, so that a
        draw's ``(ngal, xi)`` depends on the draw alone.  It costs latency:
        ``predict(model)`` ~70 us instead of ~10-18, batches below ~8000 draws
        up to 2x; batches of 10^4 draws nothing.  ``level=1`` only refuses the
        measured dispatch (``autotune``); ``False`` is the default behaviour.

        Returns whether calls with default options are batch-invariant now
        (``False`` for tables no one-launch form serves: float32, more than
        248 bins, several r tiles)."""
        device = self.to_device()
        value = 2 if level is True else int(level)
        with device.lock:
            _lib.check(device.lib.tc_table_set_option(
                device.handle, b'deterministic', value))
            out = ctypes.c_int(0)
            _lib.check(device.lib.tc_table_batch_invariant(
                device.handle, 10, 0, ctypes.byref(out)))
        return bool(out.value)

    # -- consistency checks ---------------------------------------------------

    def _check_consistency(self, model):
        """Same checks and messages as ``tabcorr/tabcorr.py:496-535``."""
        if sorted(model.gal_types) != ['centrals', 'satellites']:
            raise ValueError(
                'The model instance must only have centrals and ' +
                'satellites as galaxy types. Check the `gal_types` ' +
                'attribute of the model instance.')
        components = model._input_model_dictionary
        for name in ['centrals_occupation', 'satellites_occupation']:
            if (components[name].prim_haloprop_key !=
                    self.attrs['prim_haloprop_key']):
                raise ValueError('Mismatch in the primary halo properties ' +
                                 'of the model and the TabCorr instance.')
        for name in ['centrals_occupation', 'satellites_occupation']:
            if hasattr(components[name], 'sec_haloprop_key'):
                if (components[name].sec_haloprop_key !=
                        self.attrs['sec_haloprop_key']):
                    raise ValueError(
                        'Mismatch in the secondary halo properties ' +
                        'of the model and the TabCorr instance.')
        # (builtin abs: the ufunc dispatch of np.abs costs 0.7 us of an 18 us call)
        if not abs(model.redshift - self.attrs['redshift']) < 0.05:
            raise ValueError('Mismatch in the redshift of the model and ' +
                             'the TabCorr instance.')

    def _check_consistency_cached(self, model):
        """`_check_consistency`, skipped while nothing it reads has changed:
        the same model, component dictionary and ``gal_types`` objects, and
        equal VALUES of everything the checks compare -- the components'
        halo-property keys, the model's redshift and this table's three
        attributes -- so that an in-place change of any of them after a first
        successful call is checked again, as the reference does on every call
        (``tabcorr/tabcorr.py:496-535``)."""
        components = model._input_model_dictionary
        try:
            cen = components['centrals_occupation']
            sat = components['satellites_occupation']
            attrs = self.attrs
            values = (model.redshift, cen.prim_haloprop_key, sat.prim_haloprop_key,
                      getattr(cen, 'sec_haloprop_key', None),
                      getattr(sat, 'sec_haloprop_key', None),
                      attrs['prim_haloprop_key'], attrs.get('sec_haloprop_key'),
                      attrs['redshift'])
        except (KeyError, AttributeError, TypeError):
            values = None
        last = self._checked_model
        if (last is not None and values is not None and last[0] is model and
                last[1] is components and last[2] is model.gal_types and
                last[3] == values):
            return
        self._check_consistency(model)
        self._checked_model = (model, components, model.gal_types, values)

    # -- mean occupation ---------------------------------------------------------

    def _host_mean_occupation(self, model, n_gauss_prim, **occ_kwargs):
        """Generic models: the callbacks are Python, so the Gauss-Legendre
        average of ``tabcorr/tabcorr.py:537-578`` is formed on the host and
        only the contraction runs on the device."""
        gal_type = self.gal_type
        nodes, weights = np.polynomial.legendre.leggauss(n_gauss_prim)
        nodes = 0.5 * (nodes + 1.0)
        log_lo = gal_type['log_prim_haloprop_min']
        width = gal_type['log_prim_haloprop_max'] - log_lo
        mass = 10.0**(log_lo[:, np.newaxis] + width[:, np.newaxis] * nodes)
        percentile = np.repeat(
            gal_type['sec_haloprop_percentile'][:, np.newaxis], n_gauss_prim,
            axis=1)
        central = gal_type.is_centrals()
        occupation = np.zeros(mass.shape)
        occupation[central] = np.reshape(model.mean_occupation_centrals(
            prim_haloprop=mass[central].ravel(),
            sec_haloprop_percentile=percentile[central].ravel(),
            **occ_kwargs), (-1, n_gauss_prim))
        occupation[~central] = np.reshape(model.mean_occupation_satellites(
            prim_haloprop=mass[~central].ravel(),
            sec_haloprop_percentile=percentile[~central].ravel(),
            **occ_kwargs), (-1, n_gauss_prim))
        if 'prim_haloprop_dist_index' in gal_type.colnames:
            power = mass**(gal_type['prim_haloprop_dist_index'][
                :, np.newaxis] + 1.0)
        else:
            power = np.ones_like(mass)
        return (np.sum(weights * occupation * power, axis=-1) /
                np.sum(weights * power, axis=-1))

    def mean_occupation(self, model, n_gauss_prim=10, check_consistency=True,
                        **occ_kwargs):
        """Mean occupation of every halo/galaxy bin
        (``tabcorr/tabcorr.py:465-578``).

        Returns
        -------
        n : numpy.ndarray
            Same length as ``self.gal_type``.

        Raises
        ------
        ValueError
            If the model and the table are inconsistent.
        """
        if check_consistency:
            # (the checks read these attributes and nothing else: a model that
            # passed them is not checked again while they are the same
            # objects / values -- one predict() per MCMC step)
            self._check_consistency_cached(model)
        spec = None if occ_kwargs else device_spec(model)
        if spec is None:
            return self._host_mean_occupation(model, n_gauss_prim,
                                              **occ_kwargs)
        return self.mean_occupation_batch(
            spec.theta[np.newaxis], n_gauss_prim=n_gauss_prim,
            modulate_with_cenocc=spec.modulate_with_cenocc,
            assembias=spec.assembias, family=spec.family)[0]

    def mean_occupation_batch(self, theta, n_gauss_prim=10,
                              modulate_with_cenocc=False, assembias=False,
                              family='zheng07'):
        """`mean_occupation` for a ``(n_draws, 5 | 7)`` array of Zheng07
        parameters (columns: logMmin, sigma_logM, logM0, logM1, alpha
        [, A_cen, A_sat]) -- or, with ``family='leauthaud11'``, the 14 columns
        of `models.Leauthaud11Model.device_theta`.  Returns
        ``(n_draws, n_bins)``."""
        device = self.to_device()
        theta = _lib.contiguous(np.atleast_2d(theta))
        flags = _flags(False, modulate_with_cenocc, assembias, family)
        occupation = np.empty((len(theta), device.n_bins))
        with device.lock:
            _lib.check(device.lib.tc_mean_occupation_zheng07_batch(
                device.handle, _lib.as_double_p(theta), theta.shape[1],
                len(theta), n_gauss_prim, flags,
                _lib.as_double_p(occupation)))
        return occupation

    # -- predict ---------------------------------------------------------------------

    def predict(self, model, separate_gal_type=False, n_gauss_prim=10,
                check_consistency=True, **occ_kwargs):
        """Number density and correlation function of a model
        (``tabcorr/tabcorr.py:580-683``).

        Parameters
        ----------
        model : HodModelFactory-like or numpy.ndarray
            A model object (see `tabcorr_amd.models`) or the mean occupation
            of every bin.  A 2-D array is treated as a batch.

        Returns
        -------
        ngal : float or dict
        xi : numpy.ndarray or dict
        """
        if isinstance(model, np.ndarray):
            batched = model.ndim == 2
            ngal, xi = self._predict_occupation(
                np.atleast_2d(model), separate_gal_type)
            return _unbatch(ngal, xi) if not batched else (ngal, xi)

        if check_consistency:
            # (the checks read these attributes and nothing else: a model that
            # passed them is not checked again while they are the same
            # objects / values -- one predict() per MCMC step)
            self._check_consistency_cached(model)
        spec = None if occ_kwargs else device_spec(model)
        if spec is None:
            occupation = self._host_mean_occupation(
                model, n_gauss_prim, **occ_kwargs)
            return _unbatch(*self._predict_occupation(
                occupation[np.newaxis], separate_gal_type))
        if not separate_gal_type:
            ngal, xi = self.to_device().predict_one(
                spec.values, n_gauss_prim,
                _flags(False, spec.modulate_with_cenocc, spec.assembias,
                       spec.family))
            return ngal, xi.reshape(self.tpcf_shape)
        return _unbatch(*self.predict_batch(
            spec.theta[np.newaxis], separate_gal_type=separate_gal_type,
            n_gauss_prim=n_gauss_prim,
            modulate_with_cenocc=spec.modulate_with_cenocc,
            assembias=spec.assembias, family=spec.family))

    @staticmethod
    def predict_joint(halotabs, model, n_gauss_prim=10, check_consistency=True):
        """``[halotab.predict(model) for halotab in halotabs]`` in ONE call.

        The reference's documented likelihood step evaluates two tables per
        model (``docs/guides/overview.rst:86-92``: ``halotab_wp.predict(model)``,
        then ``halotab_ds.predict(model)``).  Here every table's call is
        posted before the first answer is waited for, so the tables' round
        trips to the device overlap (``tc_predict_zheng07_joint``): two tables
        take about as long as one and a half.  Total correlation functions
        only; each table's result has the bits of its own ``predict(model)``.

        Returns
        -------
        results : list of ``(ngal, xi)``, one per table, as `predict` returns
            them.
        """
        halotabs = list(halotabs)
        if check_consistency:
            for halotab in halotabs:
                halotab._check_consistency_cached(model)
        spec = device_spec(model)
        if spec is None or len(halotabs) < 2 or len(halotabs) > 16:
            return [halotab.predict(model, n_gauss_prim=n_gauss_prim,
                                    check_consistency=False)
                    for halotab in halotabs]
        devices = [halotab.to_device() for halotab in halotabs]
        first = devices[0]
        cache = getattr(first, '_joint', None)
        if cache is None or cache[0] != [id(d) for d in devices]:
            handles = (ctypes.c_void_p * len(devices))(
                *[d.handle.value for d in devices])
            theta = np.zeros(16)
            ngal = np.zeros(len(devices))
            xi = [np.zeros(d.n_r) for d in devices]
            xi_p = (_lib.c_double_p * len(devices))(
                *[a.ctypes.data_as(_lib.c_double_p) for a in xi])
            # (every table's lock, in one fixed order whatever the order of
            # the tables in the call)
            locks = [d.lock for d in sorted(devices, key=id)]
            cache = ([id(d) for d in devices], handles, theta,
                     _lib.as_double_p(theta), ngal, _lib.as_double_p(ngal), xi,
                     xi_p, locks, devices)
            first._joint = cache
        _, handles, theta, theta_p, ngal, ngal_p, xi, xi_p, locks, _ = cache
        values = spec.values
        flags = _flags(False, spec.modulate_with_cenocc, spec.assembias,
                       spec.family)
        for lock in locks:
            lock.acquire()
        try:
            theta[:len(values)] = values
            status = first.lib.tc_predict_zheng07_joint(
                handles, len(devices), theta_p, len(values), n_gauss_prim,
                flags, ngal_p, xi_p)
            if status:
                _lib.check(status)
            return [(ngal[k], xi[k].reshape(halotab.tpcf_shape).copy())
                    for k, halotab in enumerate(halotabs)]
        finally:
            for lock in reversed(locks):
                lock.release()

    def predict_batch(self, theta, separate_gal_type=False, n_gauss_prim=10,
                      modulate_with_cenocc=False, assembias=False,
                      family='zheng07', out=None):
        """`predict` for a ``(n_draws, 5 | 7)`` array of Zheng07 parameters
        (``family='leauthaud11'``: 14 columns, see `mean_occupation_batch`).

        ``out=(ngal, xi)``: C-contiguous float64 arrays of ``n_draws [* 2]``
        and ``n_draws * n_components * n_r`` elements that receive the results;
        the returned arrays are views of them.  Page-locked ones
        (`tabcorr_amd.pinned_empty`) are written without an intermediate copy.
        Ordinary ones kept from call to call save the page faults of a fresh
        result array: 10^4 draws of a (19, 40) table are 61 MB, 3.2 ms per
        call into a kept array against 7.3 ms into a new one.

        Returns
        -------
        ngal : numpy.ndarray ``(n_draws, )`` or dict of such
        xi : numpy.ndarray ``(n_draws, ) + tpcf_shape`` or dict of such
        """
        if out is not None and all(pinned.is_pinned(a) for a in out):
            return self.predict_batch_async(
                theta, separate_gal_type=separate_gal_type,
                n_gauss_prim=n_gauss_prim,
                modulate_with_cenocc=modulate_with_cenocc,
                assembias=assembias, family=family, out=out).wait()
        device = self.to_device()
        theta = _lib.contiguous(np.atleast_2d(theta))
        flags = _flags(separate_gal_type, modulate_with_cenocc, assembias,
                       family)
        n_draws = len(theta)
        n_comp = device.n_components if separate_gal_type else 1
        shapes = [(n_draws, 2 if separate_gal_type else 1),
                  (n_draws, n_comp, device.n_r)]
        if out is not None:
            ngal, xi = pinned.caller_outputs(shapes, out)
        else:
            ngal, xi = np.empty(shapes[0]), np.empty(shapes[1])
        with device.lock:
            _lib.check(device.lib.tc_predict_zheng07_batch(
                device.handle, _lib.as_double_p(theta), theta.shape[1],
                n_draws, n_gauss_prim, flags, _lib.as_double_p(ngal),
                _lib.as_double_p(xi)))
        return self._package(ngal, xi, separate_gal_type)

    def predict_batch_async(self, theta, separate_gal_type=False,
                            n_gauss_prim=10, modulate_with_cenocc=False,
                            assembias=False, family='zheng07', out=None):
        """`predict_batch` without waiting for the device: the upload of the
        draws, the kernels and the download of the results are queued on one
        of the handle's lanes and a `tabcorr_amd.pinned.PendingPrediction` is
        returned; ``.wait()`` gives what `predict_batch` returns.  Calls made
        before the previous ones were waited for overlap on the device
        (transfers of one call with kernels of the next).

        For the full rate keep ``theta`` and ``out`` in page-locked arrays
        (`tabcorr_amd.pinned_empty`): pageable ``theta`` is copied to a pooled
        pinned block first, and without ``out`` the results are copied out of
        one at ``wait()``.  Do not modify a pinned ``theta`` or read ``out``
        before ``wait()`` returned.
        """
        device = self.to_device()
        theta = _lib.contiguous(np.atleast_2d(theta))
        flags = _flags(separate_gal_type, modulate_with_cenocc, assembias,
                       family)
        n_draws = len(theta)
        n_comp = device.n_components if separate_gal_type else 1
        shapes = [(n_draws, 2 if separate_gal_type else 1),
                  (n_draws, n_comp, device.n_r)]
        (ngal, xi), pooled_out = pinned.stage_outputs(shapes, out)
        (theta_pinned, ), pooled_in = pinned.stage_inputs([theta])
        ticket = ctypes.c_int64(-1)
        with device.lock:
            _lib.check(device.lib.tc_predict_zheng07_batch_async(
                device.handle, _lib.as_double_p(theta_pinned), theta.shape[1],
                n_draws, n_gauss_prim, flags, _lib.as_double_p(ngal),
                _lib.as_double_p(xi), ctypes.byref(ticket)))
        return pinned.PendingPrediction(
            device, device.lib.tc_table_wait, device.lib.tc_table_query,
            ticket.value, [theta_pinned], [ngal, xi], pooled_in, pooled_out,
            lambda n, x: self._package(n, x, separate_gal_type))

    def chi2_batch(self, theta, data, precision, n_gauss_prim=10,
                   modulate_with_cenocc=False, assembias=False,
                   family='zheng07'):
        """Gaussian ``chi^2 = (xi - data)^T precision (xi - data)`` of every draw,
        evaluated on the device right after the prediction (extension: the
        reference leaves this to the user's likelihood, ``README.md:7``).

        Parameters
        ----------
        theta : ``(n_draws, 5 | 7)`` Zheng07 parameters
        data : array of shape ``tpcf_shape``
        precision : inverse covariance, ``(n_r, n_r)``

        Returns
        -------
        ngal, chi2 : numpy.ndarray ``(n_draws, )``
        """
        device = self.to_device()
        theta = _lib.contiguous(np.atleast_2d(theta))
        data = _lib.contiguous(np.ravel(data))
        precision = _lib.contiguous(precision)
        if data.shape != (device.n_r, ) or precision.shape != (device.n_r,
                                                               device.n_r):
            raise ValueError('data must have {0} entries and precision shape '
                             '({0}, {0}).'.format(device.n_r))
        ngal = np.empty(len(theta))
        chi2 = np.empty(len(theta))
        with device.lock:
            _lib.check(device.lib.tc_chi2_zheng07_batch(
                device.handle, _lib.as_double_p(theta), theta.shape[1],
                len(theta), n_gauss_prim,
                _flags(False, modulate_with_cenocc, assembias, family),
                _lib.as_double_p(data), _lib.as_double_p(precision),
                _lib.as_double_p(ngal), _lib.as_double_p(chi2)))
        return ngal, chi2

    def chi2_batch_async(self, theta, data, precision, n_gauss_prim=10,
                         modulate_with_cenocc=False, assembias=False,
                         family='zheng07', out=None):
        """`chi2_batch` without waiting (see `predict_batch_async`): 8 bytes
        per parameter go to the device and 16 bytes per draw come back.
        ``out=(ngal, chi2)``: page-locked arrays of ``n_draws`` elements."""
        device = self.to_device()
        theta = _lib.contiguous(np.atleast_2d(theta))
        data = _lib.contiguous(np.ravel(data))
        precision = _lib.contiguous(precision)
        if data.shape != (device.n_r, ) or precision.shape != (device.n_r,
                                                               device.n_r):
            raise ValueError('data must have {0} entries and precision shape '
                             '({0}, {0}).'.format(device.n_r))
        n_draws = len(theta)
        (ngal, chi2), pooled_out = pinned.stage_outputs(
            [(n_draws, ), (n_draws, )], out)
        (theta_pinned, ), pooled_in = pinned.stage_inputs([theta])
        ticket = ctypes.c_int64(-1)
        with device.lock:
            _lib.check(device.lib.tc_chi2_zheng07_batch_async(
                device.handle, _lib.as_double_p(theta_pinned), theta.shape[1],
                n_draws, n_gauss_prim,
                _flags(False, modulate_with_cenocc, assembias, family),
                _lib.as_double_p(data), _lib.as_double_p(precision),
                _lib.as_double_p(ngal), _lib.as_double_p(chi2),
                ctypes.byref(ticket)))
        return pinned.PendingPrediction(
            device, device.lib.tc_table_wait, device.lib.tc_table_query,
            ticket.value, [theta_pinned], [ngal, chi2], pooled_in, pooled_out,
            lambda n, c: (n, c))

    def _predict_occupation(self, occupation, separate_gal_type):
        device = self.to_device()
        occupation = _lib.contiguous(occupation)
        if occupation.shape[1] != device.n_bins:
            raise ValueError(
                'The mean occupation array has {} entries but the table has '
                '{} halo/galaxy bins.'.format(occupation.shape[1],
                                              device.n_bins))
        n_draws = len(occupation)
        n_comp = device.n_components if separate_gal_type else 1
        ngal = np.empty((n_draws, 2 if separate_gal_type else 1))
        xi = np.empty((n_draws, n_comp, device.n_r))
        with device.lock:
            _lib.check(device.lib.tc_predict_occupation_batch(
                device.handle, _lib.as_double_p(occupation), n_draws,
                _flags(separate_gal_type), _lib.as_double_p(ngal),
                _lib.as_double_p(xi)))
        return self._package(ngal, xi, separate_gal_type)

    def _package(self, ngal, xi, separate_gal_type):
        shape = (len(ngal), ) + tuple(self.tpcf_shape)
        if not separate_gal_type:
            return ngal[:, 0], xi[:, 0].reshape(shape)
        # dict order: np.unique order of the gal_type column and
        # combinations_with_replacement of it (tabcorr/tabcorr.py:660-675).
        ngal_dict = {key: ngal[:, i] for i, key in enumerate(NGAL_KEYS)}
        xi_dict = {key: xi[:, i].reshape(shape)
                   for i, key in enumerate(XI_KEYS[self.attrs['mode']])}
        return ngal_dict, xi_dict


def _flags(separate_gal_type=False, modulate_with_cenocc=False,
           assembias=False, family='zheng07'):
    if family not in ('zheng07', 'leauthaud11'):
        raise ValueError("family must be 'zheng07' or 'leauthaud11'.")
    return ((_lib.FLAG_SEPARATE_GAL_TYPE if separate_gal_type else 0) |
            (_lib.FLAG_MODULATE_WITH_CENOCC if modulate_with_cenocc else 0) |
            (_lib.FLAG_ASSEMBIAS if assembias else 0) |
            (_lib.FLAG_LEAUTHAUD11 if family == 'leauthaud11' else 0))


def _unbatch(ngal, xi):
    """First element of a batch of one, as the scalar API returns it."""
    if isinstance(ngal, dict):
        return ({key: value[0] for key, value in ngal.items()},
                {key: value[0] for key, value in xi.items()})
    return ngal[0], xi[0]


def symmetric_matrix_to_array(matrix, check_symmetry=True):
    """Packed lower triangle of a symmetric matrix, ``p = i (i + 1) / 2 + j``
    for ``j <= i`` -- the column order of ``tpcf_matrix`` in mode ``'auto'``
    (``tabcorr/tabcorr.py:770-806``).

    Raises
    ------
    ValueError
        If ``check_symmetry`` and the matrix is not symmetric.
    """
    matrix = np.asarray(matrix)
    if check_symmetry and (matrix.ndim != 2 or
                           matrix.shape[0] != matrix.shape[1] or
                           not np.array_equal(matrix, matrix.T)):
        raise ValueError('The matrix you provided is not symmetric.')
    rows, cols = np.tril_indices(matrix.shape[0])
    return matrix[rows, cols]
