#!/opt/conda/bin/python3.9
"""HDF5 cross-read check against the REFERENCE implementation.

Runs only in the build container (needs ``/root/reference`` and the interpreter that has
h5py + astropy, like ``make_golden.py`` whose shims it reuses)::

    /opt/conda/bin/python3.9 tests/golden/cross_read_check.py

Both directions, for a ``TabCorr`` table (``tabcorr/tabcorr.py:374-463``) and for an
``Interpolator`` (``tabcorr/interpolator.py:72-122``):

* files written by ``tabcorr_amd`` (its own HDF5 writer over libhdf5, no h5py / astropy)
  are read by the unmodified reference, and
* files written by the reference are read by ``tabcorr_amd``,

and in both cases matrix, ``gal_type`` columns, attributes, ``tpcf_shape``, ``tpcf_args`` /
``tpcf_kwargs`` and the parameter table come out equal; the reference's ``predict`` on a
file written by ``tabcorr_amd`` equals its ``predict`` on the object it was written from.
No GPU is involved (I/O only).  Prints one line per check and exits non-zero on a mismatch.
"""

import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import make_golden as ref            # noqa: E402  (shims + the imported reference)
from tabcorr_amd import TabCorr, Interpolator, synthetic   # noqa: E402

tabcorr = ref.tabcorr
failures = []


def check(name, condition):
    print('%-78s %s' % (name, 'ok' if condition else 'MISMATCH'))
    if not condition:
        failures.append(name)


def text(value):
    return value.decode() if isinstance(value, bytes) else str(value)


def same_table(name, theirs, ours, matrix_dtype=np.float32):
    """`theirs`: a reference TabCorr; `ours`: a tabcorr_amd TabCorr."""
    expect = np.asarray(ours.tpcf_matrix).astype(matrix_dtype).astype(np.float64)
    check(name + ': tpcf_matrix', np.array_equal(np.asarray(theirs.tpcf_matrix), expect))
    check(name + ': tpcf_shape', tuple(int(s) for s in theirs.tpcf_shape) ==
          tuple(int(s) for s in ours.tpcf_shape))
    mine = ours.gal_type.as_array()
    for column in mine.dtype.names:
        a, b = np.asarray(theirs.gal_type[column]), mine[column]
        if b.dtype.kind == 'S':
            equal = [text(x) for x in a] == [text(x) for x in b]
        else:
            equal = np.array_equal(a, b)
        check(name + ': gal_type[%s]' % column, equal)
    check(name + ': gal_type columns', list(theirs.gal_type.colnames) == list(mine.dtype.names))
    for key in ('tpcf', 'mode', 'simname', 'redshift', 'Num_ptcl_requirement',
                'prim_haloprop_key', 'sec_haloprop_key'):
        a, b = theirs.attrs[key], ours.attrs[key]
        equal = text(a) == text(b) if isinstance(b, (str, bytes)) else a == b
        check(name + ': attrs[%s]' % key, equal)
    check(name + ': tpcf_args', len(theirs.tpcf_args) == len(ours.tpcf_args) and all(
        np.array_equal(np.asarray(a), np.asarray(b))
        for a, b in zip(theirs.tpcf_args, ours.tpcf_args)))
    check(name + ': tpcf_kwargs', sorted(theirs.tpcf_kwargs) == sorted(ours.tpcf_kwargs) and all(
        np.array_equal(np.asarray(theirs.tpcf_kwargs[k]), np.asarray(ours.tpcf_kwargs[k]))
        for k in ours.tpcf_kwargs))


def main():
    table = synthetic.synthetic_table(12, 2, (7, 3), 'auto', seed=11)
    rp_bins = np.logspace(-1, 1, 8)
    ours = TabCorr.from_arrays(table['gal_type'], table['tpcf_matrix'], table['tpcf_shape'],
                               table['attrs'], (rp_bins, 40.0), {'period': np.array([250.0] * 3)})
    theta = synthetic.zheng07_draws(5, seed=12)
    with tempfile.TemporaryDirectory() as tmp:
        # ---- TabCorr: tabcorr_amd writes, the reference reads --------------------------------
        for dtype in (np.float32, np.float64):
            path = os.path.join(tmp, 'ours_%s.hdf5' % np.dtype(dtype).name)
            ours.write(path, matrix_dtype=dtype)
            theirs = tabcorr.TabCorr.read(path)
            same_table('tabcorr_amd -> reference (%s)' % np.dtype(dtype).name, theirs, ours, dtype)
        # the reference predicts the same from the file as from the arrays it was written from
        direct = ref.make_reference_tabcorr(table)
        for t in theta:
            model = ref.DuckZheng07(t)
            a = theirs.predict(model)
            b = direct.predict(model)
            check('reference.predict on a tabcorr_amd file', a[0] == b[0] and np.array_equal(a[1], b[1]))
        # ---- TabCorr: the reference writes, tabcorr_amd reads ----------------------------------
        direct.tpcf_args = (rp_bins, 40.0)
        direct.tpcf_kwargs = {'period': np.array([250.0] * 3)}
        path = os.path.join(tmp, 'theirs.hdf5')
        direct.write(path)
        same_table('reference -> tabcorr_amd', direct, TabCorr.read(path))
        # ---- Interpolator, both directions -------------------------------------------------------
        tables, keys, points = synthetic.synthetic_interpolator((4, 4), 6, 1, (5, ), 'auto', seed=13)
        shuffle = np.random.default_rng(14).permutation(len(tables))
        tables = [tables[i] for i in shuffle]
        points = points[shuffle]
        mine = Interpolator([TabCorr.from_arrays(t['gal_type'], t['tpcf_matrix'], t['tpcf_shape'],
                                                 t['attrs']) for t in tables],
                            {k: points[:, d] for d, k in enumerate(keys)})
        path = os.path.join(tmp, 'interp_ours.hdf5')
        mine.write(path)
        theirs = tabcorr.Interpolator.read(path)
        check('Interpolator tabcorr_amd -> reference: keys',
              [k for k in theirs.param_dict_table.colnames if k != 'tabcorr_index'] == list(keys))
        for key in list(keys) + ['tabcorr_index']:
            check('Interpolator tabcorr_amd -> reference: param_dict_table[%s]' % key,
                  np.array_equal(np.asarray(theirs.param_dict_table[key]),
                                 np.asarray(mine.param_dict_table[key])))
        for i in (0, 7, 15):
            same_table('Interpolator tabcorr_amd -> reference: tabcorr_%d' % i,
                       theirs.tabcorr_list[i], mine.tabcorr_list[i])
        reference_interp = ref.make_reference_interpolator(tables, keys, points)
        for halotab in reference_interp.tabcorr_list:
            # (a table without positional arguments cannot be read back by the reference
            # itself: tabcorr.py:401-404; tabulate always records some)
            halotab.tpcf_args = (rp_bins, 40.0)
            halotab.tpcf_kwargs = {}
        path = os.path.join(tmp, 'interp_theirs.hdf5')
        reference_interp.write(path)
        back = Interpolator.read(path)
        for key in list(keys) + ['tabcorr_index']:
            check('Interpolator reference -> tabcorr_amd: param_dict_table[%s]' % key,
                  np.array_equal(np.asarray(back.param_dict_table[key]),
                                 np.asarray(reference_interp.param_dict_table[key])))
        for i in (0, 9):
            same_table('Interpolator reference -> tabcorr_amd: tabcorr_%d' % i,
                       reference_interp.tabcorr_list[i], back.tabcorr_list[i])
        # the reference interpolates the same from either file
        for t in theta:
            extra = {k: float(np.mean(points[:, d])) + 0.01 for d, k in enumerate(keys)}
            model = ref.DuckZheng07(t, **extra)
            a = theirs.predict(model)
            b = reference_interp.predict(model)
            check('reference Interpolator.predict on a tabcorr_amd file',
                  a[0] == b[0] and np.array_equal(a[1], b[1]))
    if failures:
        print('%d mismatch(es)' % len(failures))
        return 1
    print('cross-read check passed')
    return 0


if __name__ == '__main__':
    sys.exit(main())
