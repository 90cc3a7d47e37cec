"""A small column table standing in for ``astropy.table.Table``.

The reference keeps the halo/galaxy bins in an astropy ``Table``
(``tabcorr/tabcorr.py:229-234``, read back at ``:414``).  astropy is not a
dependency of this package; `GalTypeTable` offers the handful of operations
user code performs on ``halotab.gal_type``: column access by name,
``colnames``, ``len``, row selection by mask / slice, ``as_array`` and
``copy``.  An astropy ``Table`` (or a NumPy structured array) is accepted
wherever a `GalTypeTable` is expected.
"""

import numpy as np

COLUMNS = ('n_h', 'log_prim_haloprop_min', 'log_prim_haloprop_max',
           'sec_haloprop_percentile_min', 'sec_haloprop_percentile_max',
           'prim_haloprop', 'sec_haloprop_percentile',
           'prim_haloprop_dist_index', 'gal_type')


class GalTypeTable:

    def __init__(self, data):
        if isinstance(data, GalTypeTable):
            data = data._data.copy()
        elif hasattr(data, 'as_array') and not isinstance(data, np.ndarray):
            data = np.array(data.as_array())          # astropy Table
        elif isinstance(data, dict):
            names = list(data.keys())
            n = len(data[names[0]])
            dtype = []
            for name in names:
                column = np.asarray(data[name])
                if name == 'gal_type':
                    dtype.append((name, 'S10'))
                else:
                    dtype.append((name, column.dtype))
            array = np.zeros(n, dtype=dtype)
            for name in names:
                array[name] = data[name]
            data = array
        else:
            data = np.array(data)
        if data.dtype.names is None:
            raise TypeError('gal_type must be a table with named columns.')
        self._data = data

    @property
    def colnames(self):
        return list(self._data.dtype.names)

    def __len__(self):
        return len(self._data)

    def __getitem__(self, item):
        if isinstance(item, str):
            column = self._data[item]
            if column.dtype.kind == 'S':
                # astropy compares bytes columns with str transparently
                # (relied upon at tabcorr/tabcorr.py:555); decode instead.
                return np.char.decode(column, 'utf-8')
            return column
        return GalTypeTable(self._data[item])

    def __setitem__(self, name, values):
        if name in self._data.dtype.names:
            self._data[name] = values
            return
        values = np.asarray(values)
        dtype = self._data.dtype.descr + [(name, values.dtype.str)]
        data = np.zeros(len(self._data), dtype=dtype)
        for key in self._data.dtype.names:
            data[key] = self._data[key]
        data[name] = values
        self._data = data

    def remove_column(self, name):
        names = [n for n in self._data.dtype.names if n != name]
        data = np.zeros(len(self._data), dtype=[
            (n, self._data.dtype[n]) for n in names])
        for n in names:
            data[n] = self._data[n]
        self._data = data

    def as_array(self):
        return self._data

    def copy(self):
        return GalTypeTable(self._data.copy())

    def is_centrals(self):
        column = self._data['gal_type']
        if column.dtype.kind == 'S':
            return column == b'centrals'
        return column == 'centrals'

    def __repr__(self):
        return 'GalTypeTable(%d rows: %s)' % (len(self), ', '.join(
            self.colnames))
