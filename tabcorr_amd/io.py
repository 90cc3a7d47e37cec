"""Table (de)serialisation.

Two containers hold the same content (``tabcorr/tabcorr.py:443-463``: the
seven root attributes, ``tpcf_matrix``, ``tpcf_shape``, ``tpcf_args/arg_%d``,
``tpcf_kwargs/<key>`` and the compound dataset ``gal_type``):

* HDF5 in the reference's exact layout (needs ``h5py``; files written by the
  reference are read as they are, and files written here are readable by the
  reference), and
* ``.npz`` with the same keys flattened, for environments without HDF5.
"""

import json
import os

import numpy as np

from .galtable import GalTypeTable

ATTR_KEYS = ['tpcf', 'mode', 'simname', 'redshift', 'Num_ptcl_requirement',
             'prim_haloprop_key', 'sec_haloprop_key']


def _h5py():
    try:
        import h5py
    except ImportError as error:
        raise ImportError(
            'Reading or writing HDF5 tables needs h5py; use the .npz '
            'container otherwise.') from error
    return h5py


def _plain(value):
    if isinstance(value, bytes):
        return value.decode()
    if isinstance(value, np.generic):
        return value.item()
    return value


def _is_group(obj):
    return hasattr(obj, 'attrs') and hasattr(obj, 'keys')


def read_tabcorr(cls, fname):
    if not _is_group(fname) and str(fname).endswith('.npz'):
        return _read_npz(cls, fname)
    h5py = _h5py()
    stream = fname if _is_group(fname) else h5py.File(fname, 'r')
    try:
        attrs = {key: _plain(stream.attrs[key]) for key in stream.attrs.keys()}
        matrix = stream['tpcf_matrix'][()].astype(np.float64)
        args = tuple(stream['tpcf_args'][key][()]
                     for key in stream['tpcf_args'].keys())
        kwargs = {}
        if 'tpcf_kwargs' in stream:
            kwargs = {key: stream['tpcf_kwargs'][key][()]
                      for key in stream['tpcf_kwargs'].keys()}
        shape = tuple(int(s) for s in stream['tpcf_shape'][()])
        gal_type = GalTypeTable(stream['gal_type'][()])
    finally:
        if not _is_group(fname):
            stream.close()
    return cls.from_arrays(gal_type, matrix, shape, attrs, args, kwargs)


def write_tabcorr(halotab, fname, overwrite=False, max_args_size=1000000,
                  matrix_dtype=np.float32):
    if not _is_group(fname) and str(fname).endswith('.npz'):
        return _write_npz(halotab, fname, overwrite, matrix_dtype)
    h5py = _h5py()
    stream = fname if _is_group(fname) else h5py.File(
        fname, 'w' if overwrite else 'w-')
    try:
        for key in ATTR_KEYS:
            stream.attrs[key] = halotab.attrs[key]
        stream['tpcf_matrix'] = np.asarray(halotab.tpcf_matrix).astype(
            matrix_dtype)
        for i, arg in enumerate(halotab.tpcf_args):
            if type(arg) is not np.ndarray or arg.size < max_args_size:
                stream['tpcf_args/arg_%d' % i] = arg
        if not halotab.tpcf_args:
            stream.require_group('tpcf_args')
        for key, value in halotab.tpcf_kwargs.items():
            if type(value) is not np.ndarray or value.size < max_args_size:
                stream['tpcf_kwargs/' + key] = value
        stream['tpcf_shape'] = halotab.tpcf_shape
        stream['gal_type'] = halotab.gal_type.as_array()
    finally:
        if not _is_group(fname):
            stream.close()


def _read_npz(cls, fname):
    with np.load(fname) as data:
        attrs = json.loads(str(data['attrs']))
        columns = {key[3:]: data[key] for key in data.files
                   if key.startswith('gt_')}
        args = tuple(data[key] for key in sorted(
            k for k in data.files if k.startswith('tpcf_args/')))
        kwargs = {key[len('tpcf_kwargs/'):]: data[key] for key in data.files
                  if key.startswith('tpcf_kwargs/')}
        order = [c for c in GalTypeTable.__init__.__globals__['COLUMNS']
                 if c in columns]
        gal_type = GalTypeTable({c: columns[c] for c in order})
        return cls.from_arrays(
            gal_type, data['tpcf_matrix'].astype(np.float64),
            tuple(int(s) for s in data['tpcf_shape']), attrs, args, kwargs)


def _write_npz(halotab, fname, overwrite, matrix_dtype):
    if os.path.exists(fname) and not overwrite:
        raise OSError("Unable to create file (file exists): '%s'" % fname)
    arrays = {'attrs': np.array(json.dumps(
        {key: _plain(halotab.attrs[key]) for key in ATTR_KEYS})),
        'tpcf_matrix': np.asarray(halotab.tpcf_matrix).astype(matrix_dtype),
        'tpcf_shape': np.array(halotab.tpcf_shape)}
    raw = halotab.gal_type.as_array()
    for name in raw.dtype.names:
        arrays['gt_' + name] = raw[name]
    for i, arg in enumerate(halotab.tpcf_args):
        arrays['tpcf_args/arg_%d' % i] = np.asarray(arg)
    for key, value in halotab.tpcf_kwargs.items():
        arrays['tpcf_kwargs/' + key] = np.asarray(value)
    np.savez(fname, **arrays)
