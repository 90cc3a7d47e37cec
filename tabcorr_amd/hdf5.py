"""Minimal HDF5 access through the HDF5 C library (ctypes), no h5py / astropy.

The reference stores tables with h5py + ``astropy.table`` (layout:
``tabcorr/tabcorr.py:438-463``, ``tabcorr/interpolator.py:118-122``): scalar
attributes (variable- or fixed-length strings, int64, float64), contiguous
numeric datasets, one compound dataset ``gal_type`` (eight ``<f8`` columns and
an ``S10`` column) and plain groups.  This module reads and writes exactly
that subset so that ``TabCorr.read`` / ``write`` work wherever ``libhdf5`` is
installed, with files interchangeable with the reference's.

The shared library is looked up in ``$TABCORR_AMD_HDF5_LIB``, through
``ctypes.util.find_library('hdf5')`` and in a few conventional places.
"""

import ctypes
import ctypes.util
import glob
import os

import numpy as np

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int

H5F_ACC_RDONLY = 0
H5F_ACC_TRUNC = 2
H5F_ACC_EXCL = 4
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND, H5T_ENUM = 0, 1, 3, 6, 8
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5T_SGN_NONE = 0
H5_INDEX_NAME = 0
H5_ITER_NATIVE = 2
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1

_lib = None


class HDF5Error(OSError):
    pass


def _candidates():
    paths = []
    if os.environ.get('TABCORR_AMD_HDF5_LIB'):
        paths.append(os.environ['TABCORR_AMD_HDF5_LIB'])
    found = ctypes.util.find_library('hdf5')
    if found:
        paths.append(found)
    for pattern in ['/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*',
                    '/usr/lib/x86_64-linux-gnu/libhdf5*.so*',
                    '/usr/lib64/libhdf5.so*', '/opt/conda/lib/libhdf5.so',
                    '/usr/local/lib/libhdf5.so*']:
        paths.extend(sorted(glob.glob(pattern)))
    return paths


def library():
    """Load libhdf5 (once).  Raises ImportError if it cannot be found."""
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
        except OSError as error:
            last = error
            continue
        if lib.H5open() < 0:
            continue
        _declare(lib)
        _lib = lib
        return lib
    raise ImportError('The HDF5 C library (libhdf5) was not found; set '
                      'TABCORR_AMD_HDF5_LIB or use the .npz container. (%s)'
                      % last)


def available():
    try:
        library()
        return True
    except ImportError:
        return False


def _declare(lib):
    def sig(name, restype, *argtypes):
        function = getattr(lib, name)
        function.restype = restype
        function.argtypes = list(argtypes)
    c_char_p, c_void_p, c_size_t = (ctypes.c_char_p, ctypes.c_void_p,
                                    ctypes.c_size_t)
    sig('H5Eset_auto2', herr_t, hid_t, c_void_p, c_void_p)
    sig('H5Fopen', hid_t, c_char_p, ctypes.c_uint, hid_t)
    sig('H5Fcreate', hid_t, c_char_p, ctypes.c_uint, hid_t, hid_t)
    sig('H5Fclose', herr_t, hid_t)
    sig('H5Gopen2', hid_t, hid_t, c_char_p, hid_t)
    sig('H5Gcreate2', hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t)
    sig('H5Gclose', herr_t, hid_t)
    sig('H5Lexists', ctypes.c_int, hid_t, c_char_p, hid_t)
    sig('H5Literate', herr_t, hid_t, ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(hsize_t), c_void_p, c_void_p)
    sig('H5Oopen', hid_t, hid_t, c_char_p, hid_t)
    sig('H5Oclose', herr_t, hid_t)
    sig('H5Iget_type', ctypes.c_int, hid_t)
    sig('H5Dopen2', hid_t, hid_t, c_char_p, hid_t)
    sig('H5Dcreate2', hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t,
        hid_t)
    sig('H5Dclose', herr_t, hid_t)
    sig('H5Dget_type', hid_t, hid_t)
    sig('H5Dget_space', hid_t, hid_t)
    sig('H5Dread', herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p)
    sig('H5Dwrite', herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p)
    sig('H5Dvlen_reclaim', herr_t, hid_t, hid_t, hid_t, c_void_p)
    sig('H5Screate', hid_t, ctypes.c_int)
    sig('H5Screate_simple', hid_t, ctypes.c_int, ctypes.POINTER(hsize_t),
        ctypes.POINTER(hsize_t))
    sig('H5Sclose', herr_t, hid_t)
    sig('H5Sget_simple_extent_ndims', ctypes.c_int, hid_t)
    sig('H5Sget_simple_extent_dims', ctypes.c_int, hid_t,
        ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t))
    sig('H5Tclose', herr_t, hid_t)
    sig('H5Tcopy', hid_t, hid_t)
    sig('H5Tcreate', hid_t, ctypes.c_int, c_size_t)
    sig('H5Tinsert', herr_t, hid_t, c_char_p, c_size_t, hid_t)
    sig('H5Tget_class', ctypes.c_int, hid_t)
    sig('H5Tget_size', c_size_t, hid_t)
    sig('H5Tget_sign', ctypes.c_int, hid_t)
    sig('H5Tset_size', herr_t, hid_t, c_size_t)
    sig('H5Tis_variable_str', ctypes.c_int, hid_t)
    sig('H5Tset_strpad', herr_t, hid_t, ctypes.c_int)
    sig('H5Tget_nmembers', ctypes.c_int, hid_t)
    sig('H5Tget_member_name', c_void_p, hid_t, ctypes.c_uint)
    sig('H5Tget_member_type', hid_t, hid_t, ctypes.c_uint)
    sig('H5Tget_member_offset', c_size_t, hid_t, ctypes.c_uint)
    sig('H5Tget_super', hid_t, hid_t)
    sig('H5free_memory', herr_t, c_void_p)
    sig('H5Aget_num_attrs', ctypes.c_int, hid_t)
    sig('H5Tset_cset', herr_t, hid_t, ctypes.c_int)
    sig('H5Aopen_by_idx', hid_t, hid_t, c_char_p, ctypes.c_int, ctypes.c_int,
        hsize_t, hid_t, hid_t)
    sig('H5Aget_name', ctypes.c_ssize_t, hid_t, c_size_t, c_char_p)
    sig('H5Aget_type', hid_t, hid_t)
    sig('H5Aget_space', hid_t, hid_t)
    sig('H5Aread', herr_t, hid_t, hid_t, c_void_p)
    sig('H5Acreate2', hid_t, hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t)
    sig('H5Awrite', herr_t, hid_t, hid_t, c_void_p)
    sig('H5Aclose', herr_t, hid_t)
    lib.H5Eset_auto2(0, None, None)      # errors are reported through return codes


def _native(name):
    return hid_t.in_dll(library(), name).value


def _check(value, what):
    if value < 0:
        raise HDF5Error('HDF5: %s failed' % what)
    return value


# -- HDF5 type <-> NumPy dtype -------------------------------------------------------

def _dtype_of(type_id):
    """NumPy dtype for an HDF5 file type (None for variable-length strings)."""
    lib = library()
    cls = lib.H5Tget_class(type_id)
    size = lib.H5Tget_size(type_id)
    if cls == H5T_INTEGER:
        signed = lib.H5Tget_sign(type_id) != H5T_SGN_NONE
        return np.dtype('%s%d' % ('i' if signed else 'u', size))
    if cls == H5T_FLOAT:
        return np.dtype('f%d' % size)
    if cls == H5T_STRING:
        if lib.H5Tis_variable_str(type_id) > 0:
            return None
        return np.dtype('S%d' % size)
    if cls == H5T_ENUM:
        parent = lib.H5Tget_super(type_id)
        dtype = _dtype_of(parent)
        lib.H5Tclose(parent)
        return dtype
    if cls == H5T_COMPOUND:
        names, formats, offsets = [], [], []
        for i in range(lib.H5Tget_nmembers(type_id)):
            pointer = lib.H5Tget_member_name(type_id, i)
            names.append(ctypes.string_at(pointer).decode())
            lib.H5free_memory(pointer)
            member = lib.H5Tget_member_type(type_id, i)
            formats.append(_dtype_of(member))
            lib.H5Tclose(member)
            offsets.append(lib.H5Tget_member_offset(type_id, i))
        if any(f is None for f in formats):
            raise HDF5Error('variable-length strings inside compound types '
                            'are not supported')
        return np.dtype({'names': names, 'formats': formats,
                         'offsets': offsets, 'itemsize': size})
    raise HDF5Error('unsupported HDF5 type class %d' % cls)


_NATIVE = {'i1': 'H5T_NATIVE_SCHAR_g', 'u1': 'H5T_NATIVE_UCHAR_g',
           'i2': 'H5T_NATIVE_SHORT_g', 'u2': 'H5T_NATIVE_USHORT_g',
           'i4': 'H5T_NATIVE_INT_g', 'u4': 'H5T_NATIVE_UINT_g',
           'i8': 'H5T_NATIVE_LONG_g', 'u8': 'H5T_NATIVE_ULONG_g',
           'f4': 'H5T_NATIVE_FLOAT_g', 'f8': 'H5T_NATIVE_DOUBLE_g'}


def _type_of(dtype):
    """New HDF5 type id describing a NumPy dtype as laid out in memory."""
    lib = library()
    dtype = np.dtype(dtype)
    if dtype.names is not None:
        type_id = _check(lib.H5Tcreate(H5T_COMPOUND, dtype.itemsize),
                         'H5Tcreate')
        for name in dtype.names:
            member = _type_of(dtype.fields[name][0])
            _check(lib.H5Tinsert(type_id, name.encode(),
                                 dtype.fields[name][1], member), 'H5Tinsert')
            lib.H5Tclose(member)
        return type_id
    if dtype.kind == 'S':
        type_id = _check(lib.H5Tcopy(_native('H5T_C_S1_g')), 'H5Tcopy')
        lib.H5Tset_size(type_id, max(dtype.itemsize, 1))
        # NumPy 'S' fields are null-PADDED: all itemsize bytes may be characters
        # ('satellites' fills an S10 completely)
        lib.H5Tset_strpad(type_id, 1)          # H5T_STR_NULLPAD
        return type_id
    if dtype.kind == 'b':
        return _check(lib.H5Tcopy(_native('H5T_NATIVE_SCHAR_g')), 'H5Tcopy')
    key = '%s%d' % (dtype.kind, dtype.itemsize)
    if key not in _NATIVE:
        raise HDF5Error('unsupported dtype %s' % dtype)
    return _check(lib.H5Tcopy(_native(_NATIVE[key])), 'H5Tcopy')


def _shape_of(space_id):
    lib = library()
    ndims = lib.H5Sget_simple_extent_ndims(space_id)
    if ndims <= 0:
        return ()
    dims = (hsize_t * ndims)()
    lib.H5Sget_simple_extent_dims(space_id, dims, None)
    return tuple(int(d) for d in dims)


class Group:
    """A group (or the file root) of an open HDF5 file."""

    def __init__(self, file, loc_id, owned=True):
        self.file = file
        self.id = loc_id
        self._owned = owned

    # context management mirrors h5py closely enough for io.py
    def close(self):
        if self.id and self._owned:
            library().H5Gclose(self.id)
        self.id = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __contains__(self, name):
        return library().H5Lexists(self.id, name.encode(), H5P_DEFAULT) > 0

    def keys(self):
        names = []
        callback_type = ctypes.CFUNCTYPE(herr_t, hid_t, ctypes.c_char_p,
                                         ctypes.c_void_p, ctypes.c_void_p)

        def visit(group, name, info, data):
            names.append(name.decode())
            return 0
        callback = callback_type(visit)
        index = hsize_t(0)
        _check(library().H5Literate(
            self.id, H5_INDEX_NAME, H5_ITER_NATIVE, ctypes.byref(index),
            ctypes.cast(callback, ctypes.c_void_p), None), 'H5Literate')
        return sorted(names)

    def group(self, name):
        return Group(self.file, _check(library().H5Gopen2(
            self.id, name.encode(), H5P_DEFAULT), 'H5Gopen2 ' + name))

    def create_group(self, name):
        return Group(self.file, _check(library().H5Gcreate2(
            self.id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT),
            'H5Gcreate2 ' + name))

    def require_group(self, name):
        return self.group(name) if name in self else self.create_group(name)

    def is_group(self, name):
        lib = library()
        obj = _check(lib.H5Oopen(self.id, name.encode(), H5P_DEFAULT),
                     'H5Oopen ' + name)
        kind = lib.H5Iget_type(obj)
        lib.H5Oclose(obj)
        return kind == 2        # H5I_GROUP

    # -- datasets -----------------------------------------------------------

    def read(self, name):
        lib = library()
        dataset = _check(lib.H5Dopen2(self.id, name.encode(), H5P_DEFAULT),
                         'H5Dopen2 ' + name)
        try:
            file_type = lib.H5Dget_type(dataset)
            space = lib.H5Dget_space(dataset)
            shape = _shape_of(space)
            dtype = _dtype_of(file_type)
            if dtype is None:
                raise HDF5Error('variable-length string datasets are not '
                                'supported (%s)' % name)
            # packed in-memory layout (drops file padding of compounds)
            if dtype.names is not None:
                dtype = np.dtype([(n, dtype.fields[n][0])
                                  for n in dtype.names])
            out = np.empty(shape, dtype=dtype)
            memory_type = _type_of(dtype)
            status = lib.H5Dread(dataset, memory_type, H5S_ALL, H5S_ALL,
                                 H5P_DEFAULT,
                                 out.ctypes.data_as(ctypes.c_void_p))
            lib.H5Tclose(memory_type)
            lib.H5Tclose(file_type)
            lib.H5Sclose(space)
            _check(status, 'H5Dread ' + name)
        finally:
            lib.H5Dclose(dataset)
        return out if out.shape else out[()]

    def write(self, name, array):
        lib = library()
        parts = name.split('/')
        parent = self
        opened = []
        for part in parts[:-1]:
            parent = parent.require_group(part)
            opened.append(parent)
        array = np.asarray(array)
        if array.dtype.kind == 'U':
            array = np.char.encode(array, 'utf-8')
        if array.dtype.kind == 'O':
            raise HDF5Error('object arrays cannot be written (%s)' % name)
        if not array.flags.c_contiguous:      # (ascontiguousarray would make 0-d 1-d)
            array = array.copy(order='C')
        if array.shape:
            dims = (hsize_t * array.ndim)(*array.shape)
            space = _check(lib.H5Screate_simple(array.ndim, dims, None),
                           'H5Screate_simple')
        else:
            space = _check(lib.H5Screate(H5S_SCALAR), 'H5Screate')
        type_id = _type_of(array.dtype)
        dataset = _check(lib.H5Dcreate2(
            parent.id, parts[-1].encode(), type_id, space, H5P_DEFAULT,
            H5P_DEFAULT, H5P_DEFAULT), 'H5Dcreate2 ' + name)
        status = lib.H5Dwrite(dataset, type_id, H5S_ALL, H5S_ALL, H5P_DEFAULT,
                              array.ctypes.data_as(ctypes.c_void_p))
        lib.H5Dclose(dataset)
        lib.H5Tclose(type_id)
        lib.H5Sclose(space)
        for group in reversed(opened):
            group.close()
        _check(status, 'H5Dwrite ' + name)

    # -- attributes ------------------------------------------------------------

    def attrs(self):
        lib = library()
        result = {}
        for index in range(max(lib.H5Aget_num_attrs(self.id), 0)):
            attr = _check(lib.H5Aopen_by_idx(
                self.id, b'.', H5_INDEX_NAME, H5_ITER_NATIVE, index,
                H5P_DEFAULT, H5P_DEFAULT), 'H5Aopen_by_idx')
            size = lib.H5Aget_name(attr, 0, None)
            buffer = ctypes.create_string_buffer(size + 1)
            lib.H5Aget_name(attr, size + 1, buffer)
            name = buffer.value.decode()
            file_type = lib.H5Aget_type(attr)
            space = lib.H5Aget_space(attr)
            shape = _shape_of(space)
            dtype = _dtype_of(file_type)
            if dtype is None:       # variable-length string(s)
                count = int(np.prod(shape)) if shape else 1
                pointers = (ctypes.c_char_p * count)()
                # the file's own type: same character set, char* in memory
                memory_type = lib.H5Tcopy(file_type)
                _check(lib.H5Aread(attr, memory_type, pointers), 'H5Aread')
                values = [(p or b'').decode('utf-8', 'replace')
                          for p in pointers]
                lib.H5Dvlen_reclaim(memory_type, space, H5P_DEFAULT, pointers)
                lib.H5Tclose(memory_type)
                value = np.array(values).reshape(shape) if shape else values[0]
            else:
                out = np.empty(shape, dtype=dtype)
                memory_type = _type_of(dtype)
                _check(lib.H5Aread(attr, memory_type,
                                   out.ctypes.data_as(ctypes.c_void_p)),
                       'H5Aread')
                lib.H5Tclose(memory_type)
                value = out if shape else out[()]
                if lib.H5Tget_class(file_type) == H5T_ENUM and not shape:
                    value = bool(value)
            lib.H5Tclose(file_type)
            lib.H5Sclose(space)
            lib.H5Aclose(attr)
            result[name] = value
        return result

    def set_attr(self, name, value):
        lib = library()
        space = _check(lib.H5Screate(H5S_SCALAR), 'H5Screate')
        if isinstance(value, str):
            # variable-length UTF-8 string, as h5py writes a Python str
            type_id = lib.H5Tcopy(_native('H5T_C_S1_g'))
            lib.H5Tset_size(type_id, H5T_VARIABLE)
            lib.H5Tset_cset(type_id, 1)            # H5T_CSET_UTF8
            pointer = ctypes.c_char_p(value.encode('utf-8'))
            data = ctypes.byref(pointer)
        else:
            array = np.asarray(value)
            if array.dtype.kind == 'U':
                array = np.char.encode(array, 'utf-8')
            array = array.copy(order='C')
            type_id = _type_of(array.dtype)
            data = array.ctypes.data_as(ctypes.c_void_p)
        attr = _check(lib.H5Acreate2(self.id, name.encode(), type_id, space,
                                     H5P_DEFAULT, H5P_DEFAULT),
                      'H5Acreate2 ' + name)
        status = lib.H5Awrite(attr, type_id, data)
        lib.H5Aclose(attr)
        lib.H5Tclose(type_id)
        lib.H5Sclose(space)
        _check(status, 'H5Awrite ' + name)


class File(Group):
    """``File(path, 'r' | 'w' | 'w-')``."""

    def __init__(self, path, mode='r'):
        lib = library()
        encoded = os.fsencode(path)
        if mode == 'r':
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            file_id = lib.H5Fopen(encoded, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == 'w':
            file_id = lib.H5Fcreate(encoded, H5F_ACC_TRUNC, H5P_DEFAULT,
                                    H5P_DEFAULT)
        elif mode == 'w-':
            if os.path.exists(path):
                raise OSError("Unable to create file (file exists): '%s'"
                              % path)
            file_id = lib.H5Fcreate(encoded, H5F_ACC_EXCL, H5P_DEFAULT,
                                    H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r', 'w' or 'w-'")
        if file_id < 0:
            raise HDF5Error("cannot open '%s' (mode %s)" % (path, mode))
        self.file_id = file_id
        root = _check(lib.H5Gopen2(file_id, b'/', H5P_DEFAULT), 'H5Gopen2 /')
        Group.__init__(self, self, root)

    def close(self):
        if self.id:
            library().H5Gclose(self.id)
            library().H5Fclose(self.file_id)
        self.id = 0
