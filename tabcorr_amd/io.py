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
        return _tabcorr_from_flat(cls, {key: data[key] for key in data.files})


def _write_npz(halotab, fname, overwrite, matrix_dtype):
    if os.path.exists(fname) and not overwrite:
        raise OSError("Unable to create file (file exists): '%s'" % fname)
    np.savez(fname, **_tabcorr_to_flat(halotab, matrix_dtype))


# -- Interpolator ------------------------------------------------------------------
#
# HDF5 layout of the reference (tabcorr/interpolator.py:88-96, 118-122): a
# compound dataset ``param_dict_table`` (one column per key plus
# ``tabcorr_index``) and one group ``tabcorr_{i}`` per instance.

def read_interpolator(cls, fname):
    from .tabcorr import TabCorr
    if str(fname).endswith('.npz'):
        with np.load(fname) as data:
            keys = [str(k) for k in data['interp_keys']]
            points = data['interp_points']
            tables = []
            for i in range(len(points)):
                prefix = 'tabcorr_%d/' % i
                sub = {key[len(prefix):]: data[key] for key in data.files
                       if key.startswith(prefix)}
                tables.append(_tabcorr_from_flat(TabCorr, sub))
        return cls(tables, {key: points[:, d] for d, key in enumerate(keys)})
    h5py = _h5py()
    with h5py.File(fname, 'r') as stream:
        table = stream['param_dict_table'][()]
        order = np.argsort(table['tabcorr_index'])
        keys = [name for name in table.dtype.names if name != 'tabcorr_index']
        columns = {key: np.asarray(table[key], dtype=np.float64)[order]
                   for key in keys}
        tables = [read_tabcorr(TabCorr, stream['tabcorr_{}'.format(i)])
                  for i in range(len(order))]
    return cls(tables, columns)


def write_interpolator(interp, fname, overwrite=False, max_args_size=1000000,
                       matrix_dtype=np.float32):
    if str(fname).endswith('.npz'):
        if os.path.exists(fname) and not overwrite:
            raise OSError("Unable to create file (file exists): '%s'" % fname)
        arrays = {'interp_keys': np.array(interp.keys),
                  'interp_points': interp.points}
        for i, halotab in enumerate(interp.tabcorr_list):
            for key, value in _tabcorr_to_flat(halotab, matrix_dtype).items():
                arrays['tabcorr_%d/%s' % (i, key)] = value
        np.savez(fname, **arrays)
        return
    h5py = _h5py()
    with h5py.File(fname, 'w' if overwrite else 'w-') as stream:
        dtype = [(key, '<f8') for key in interp.keys] + [
            ('tabcorr_index', '<i8')]
        table = np.zeros(len(interp.points), dtype=dtype)
        for key in interp.keys:
            table[key] = interp.param_dict_table[key]
        table['tabcorr_index'] = interp.param_dict_table['tabcorr_index']
        stream['param_dict_table'] = table
        for i, halotab in enumerate(interp.tabcorr_list):
            write_tabcorr(halotab, stream.create_group('tabcorr_{}'.format(i)),
                          max_args_size=max_args_size,
                          matrix_dtype=matrix_dtype)


def _tabcorr_to_flat(halotab, matrix_dtype):
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
    return arrays


def _tabcorr_from_flat(cls, data):
    from .galtable import COLUMNS
    attrs = json.loads(str(data['attrs']))
    columns = {key[3:]: data[key] for key in data if key.startswith('gt_')}
    gal_type = GalTypeTable({c: columns[c] for c in COLUMNS if c in columns})
    args = tuple(data[key] for key in sorted(
        k for k in data if k.startswith('tpcf_args/')))
    kwargs = {key[len('tpcf_kwargs/'):]: data[key] for key in data
              if key.startswith('tpcf_kwargs/')}
    return cls.from_arrays(
        gal_type, np.asarray(data['tpcf_matrix']).astype(np.float64),
        tuple(int(s) for s in data['tpcf_shape']), attrs, args, kwargs)
