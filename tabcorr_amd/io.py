"""Table (de)serialisation.

Two containers hold the same content (``tabcorr/tabcorr.py:443-463``: the
seven root attributes, ``tpcf_matrix``, ``tpcf_shape``, ``tpcf_args/arg_%d``,
``tpcf_kwargs/<key>`` and the compound dataset ``gal_type``):

* HDF5 in the reference's exact layout, through the HDF5 C library
  (`tabcorr_amd.hdf5`, no h5py / astropy needed) or h5py when only that is
  installed; files written by the reference are read as they are, and files
  written here are readable by the reference, and
* ``.npz`` with the same keys flattened, for environments without HDF5.
"""

import json
import os

import numpy as np

from .galtable import GalTypeTable

ATTR_KEYS = ['tpcf', 'mode', 'simname', 'redshift', 'Num_ptcl_requirement',
             'prim_haloprop_key', 'sec_haloprop_key']


class _H5pyNode:
    """Adapter giving an h5py File / Group the small interface of
    `tabcorr_amd.hdf5.Group`."""

    def __init__(self, node):
        self.node = node

    def attrs(self):
        return {key: self.node.attrs[key] for key in self.node.attrs.keys()}

    def set_attr(self, name, value):
        self.node.attrs[name] = value

    def keys(self):
        return sorted(self.node.keys())

    def __contains__(self, name):
        return name in self.node

    def read(self, name):
        return self.node[name][()]

    def write(self, name, array):
        self.node[name] = array

    def group(self, name):
        return _H5pyNode(self.node[name])

    def create_group(self, name):
        return _H5pyNode(self.node.create_group(name))

    def require_group(self, name):
        return _H5pyNode(self.node.require_group(name))

    def close(self):
        pass


def _open(fname, mode):
    """Return (node, must_close) for a file name, an h5py Group or a
    `tabcorr_amd.hdf5.Group`."""
    from . import hdf5
    if isinstance(fname, (hdf5.Group, _H5pyNode)):
        return fname, False
    if _is_group(fname):
        return _H5pyNode(fname), False
    if hdf5.available():
        return hdf5.File(str(fname), mode), True
    try:
        import h5py
    except ImportError as error:
        raise ImportError(
            'Reading or writing HDF5 tables needs the HDF5 C library '
            '(libhdf5, see tabcorr_amd/hdf5.py) or h5py; use the .npz '
            'container otherwise.') from error
    return _H5pyNode(h5py.File(fname, mode)), True


def _plain(value):
    if isinstance(value, bytes):
        return value.decode()
    if isinstance(value, np.generic):
        return value.item()
    if isinstance(value, np.ndarray) and value.shape == ():
        return _plain(value[()])
    return value


def _is_group(obj):
    return hasattr(obj, 'attrs') and hasattr(obj, 'keys') and not isinstance(
        obj, dict)


def read_tabcorr(cls, fname):
    if isinstance(fname, (str, os.PathLike)) and str(fname).endswith('.npz'):
        return _read_npz(cls, fname)
    node, must_close = _open(fname, 'r')
    try:
        attrs = {key: _plain(value) for key, value in node.attrs().items()}
        matrix = np.asarray(node.read('tpcf_matrix')).astype(np.float64)
        # (a table written without positional arguments has no such group; the reference's
        # reader raises KeyError there -- tabcorr.py:401-404 -- this one returns ())
        args = ()
        if 'tpcf_args' in node:
            args_group = node.group('tpcf_args')
            args = tuple(args_group.read(key) for key in args_group.keys())
            args_group.close()
        kwargs = {}
        if 'tpcf_kwargs' in node:
            kwargs_group = node.group('tpcf_kwargs')
            kwargs = {key: kwargs_group.read(key)
                      for key in kwargs_group.keys()}
            kwargs_group.close()
        shape = tuple(int(s) for s in np.atleast_1d(node.read('tpcf_shape')))
        gal_type = GalTypeTable(node.read('gal_type'))
    finally:
        if must_close:
            node.close()
    return cls.from_arrays(gal_type, matrix, shape, attrs, args, kwargs)


def write_tabcorr(halotab, fname, overwrite=False, max_args_size=1000000,
                  matrix_dtype=np.float32):
    if isinstance(fname, (str, os.PathLike)) and str(fname).endswith('.npz'):
        return _write_npz(halotab, fname, overwrite, matrix_dtype)
    node, must_close = _open(fname, 'w' if overwrite else 'w-')
    try:
        for key in ATTR_KEYS:
            node.set_attr(key, halotab.attrs[key])
        node.write('tpcf_matrix', np.asarray(halotab.tpcf_matrix).astype(
            matrix_dtype))
        args_group = node.require_group('tpcf_args')
        for i, arg in enumerate(halotab.tpcf_args):
            if type(arg) is not np.ndarray or arg.size < max_args_size:
                args_group.write('arg_%d' % i, np.asarray(arg))
        args_group.close()
        if halotab.tpcf_kwargs:
            kwargs_group = node.require_group('tpcf_kwargs')
            for key, value in halotab.tpcf_kwargs.items():
                if (type(value) is not np.ndarray or
                        value.size < max_args_size):
                    kwargs_group.write(key, np.asarray(value))
            kwargs_group.close()
        node.write('tpcf_shape', np.asarray(halotab.tpcf_shape,
                                            dtype=np.int64))
        node.write('gal_type', halotab.gal_type.as_array())
    finally:
        if must_close:
            node.close()


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
    node, must_close = _open(fname, 'r')
    try:
        table = node.read('param_dict_table')
        order = np.argsort(table['tabcorr_index'])
        keys = [name for name in table.dtype.names if name != 'tabcorr_index']
        columns = {key: np.asarray(table[key], dtype=np.float64)[order]
                   for key in keys}
        tables = []
        for i in range(len(order)):
            group = node.group('tabcorr_{}'.format(i))
            tables.append(read_tabcorr(TabCorr, group))
            group.close()
    finally:
        if must_close:
            node.close()
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
    node, must_close = _open(fname, 'w' if overwrite else 'w-')
    try:
        dtype = [(key, '<f8') for key in interp.keys] + [
            ('tabcorr_index', '<i8')]
        table = np.zeros(len(interp.points), dtype=dtype)
        for key in interp.keys:
            table[key] = interp.param_dict_table[key]
        table['tabcorr_index'] = interp.param_dict_table['tabcorr_index']
        node.write('param_dict_table', table)
        for i, halotab in enumerate(interp.tabcorr_list):
            group = node.create_group('tabcorr_{}'.format(i))
            write_tabcorr(halotab, group, max_args_size=max_args_size,
                          matrix_dtype=matrix_dtype)
            group.close()
    finally:
        if must_close:
            node.close()


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
