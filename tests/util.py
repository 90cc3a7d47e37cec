"""Helpers shared by the tests: golden fixtures -> plain table dicts."""

import json
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')

XI_KEYS_AUTO = ['centrals-centrals', 'centrals-satellites',
                'satellites-satellites']
XI_KEYS_CROSS = ['centrals', 'satellites']


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def table_from_golden(data, prefix=''):
    columns = [key[len(prefix) + 3:] for key in data.files
               if key.startswith(prefix + 'gt_')]
    dtype = np.dtype([(c, data[prefix + 'gt_' + c].dtype) for c in columns])
    gal_type = np.zeros(len(data[prefix + 'gt_n_h']), dtype=dtype)
    for c in columns:
        gal_type[c] = data[prefix + 'gt_' + c]
    return {'gal_type': gal_type,
            'tpcf_matrix': data[prefix + 'tpcf_matrix'].astype(np.float64),
            'tpcf_shape': tuple(int(s) for s in data[prefix + 'tpcf_shape']),
            'attrs': json.loads(str(data[prefix + 'attrs']))}


def interpolator_tables_from_golden(data):
    """Tables of the synthetic interpolator fixtures (shared gal_type except
    for n_h, one matrix per table)."""
    base = table_from_golden(data, 'table0_')
    tables = []
    for k in range(len(data['points'])):
        table = dict(base)
        table['gal_type'] = base['gal_type'].copy()
        table['gal_type']['n_h'] = data['n_h'][k]
        table['tpcf_matrix'] = data['tpcf_matrices'][k].astype(np.float64)
        tables.append(table)
    return tables


def xi_keys(table):
    return XI_KEYS_AUTO if table['attrs']['mode'] == 'auto' else XI_KEYS_CROSS


def assert_rel(actual, desired, rtol, what='', floor=1e-14):
    """Elementwise relative agreement, with an absolute floor of ``floor`` x
    the largest |desired| (a few ulp of the scale of the array: elements more
    than ~1e4 below the largest one are sums whose rounding is set by their
    larger neighbours' scale)."""
    actual = np.asarray(actual, dtype=np.float64)
    desired = np.asarray(desired, dtype=np.float64)
    assert actual.shape == desired.shape, (what, actual.shape, desired.shape)
    scale = np.max(np.abs(desired)) if desired.size else 0.0
    np.testing.assert_allclose(actual, desired, rtol=rtol,
                               atol=floor * scale, err_msg=what)
