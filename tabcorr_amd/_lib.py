"""ctypes binding of libtabcorr_hip.so (C ABI: include/tabcorr_amd.h).

The library is the only compute backend.  If it is missing, or if no gfx950
device is usable, the product path raises -- there is deliberately no CPU
fallback.
"""

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY = os.path.join(HERE, 'libtabcorr_hip.so')

TC_OK = 0
TC_ERR_INVALID = 1
TC_ERR_HIP = 2
TC_ERR_UNSUPPORTED = 3
TC_ERR_RCCL = 4

MODE = {'auto': 0, 'cross': 1}
DTYPE_F64 = 0
DTYPE_F32 = 1

FLAG_SEPARATE_GAL_TYPE = 1
FLAG_MODULATE_WITH_CENOCC = 2
FLAG_ASSEMBIAS = 4
FLAG_LEAUTHAUD11 = 16

UNIQUE_ID_BYTES = 128

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
c_int64_p = ctypes.POINTER(ctypes.c_int64)
c_uint8_p = ctypes.POINTER(ctypes.c_uint8)
c_float_p = ctypes.POINTER(ctypes.c_float)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (argtypes); every function returns int except tc_last_error.
SIGNATURES = {
    'tc_device_count': [c_int_p],
    'tc_set_device': [ctypes.c_int],
    'tc_get_device': [c_int_p],
    'tc_runtime_version': [c_int_p],
    'tc_device_name': [ctypes.c_char_p, ctypes.c_size_t],
    'tc_device_synchronize': [],
    'tc_device_malloc': [c_void_pp, ctypes.c_size_t],
    'tc_device_free': [ctypes.c_void_p],
    'tc_memcpy_h2d': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t],
    'tc_memcpy_d2h': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t],
    'tc_host_alloc': [c_void_pp, ctypes.c_size_t],
    'tc_host_free': [ctypes.c_void_p],
    'tc_host_register': [ctypes.c_void_p, ctypes.c_size_t],
    'tc_host_unregister': [ctypes.c_void_p],
    'tc_host_is_pinned': [ctypes.c_void_p, ctypes.c_size_t, c_int_p],
    'tc_gauss_legendre': [ctypes.c_int, c_double_p, c_double_p],
    'tc_debug_fastmath': [ctypes.c_int, ctypes.c_int64, c_double_p,
                          c_double_p],
    'tc_pair_indices': [ctypes.c_int, c_int32_p, c_int32_p, c_int32_p],
    'tc_spline_interpolation_matrix': [ctypes.c_int, c_double_p, c_double_p],
    'tc_plan_debug': [ctypes.c_int, ctypes.c_int, c_uint8_p, ctypes.c_int,
                      c_int64_p, c_int32_p, c_int32_p, c_int32_p],
    'tc_debug_triangle_parts': [ctypes.c_int, ctypes.c_int, c_int_p, c_int_p, c_int_p],
    'tc_debug_central_series': [ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                ctypes.c_int64, c_double_p, c_double_p, c_double_p, c_double_p,
                                c_int32_p],
    'tc_debug_satellite_series': [ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                  ctypes.c_double, ctypes.c_int64, c_double_p, c_double_p,
                                  c_double_p, c_double_p, c_double_p, c_int32_p],
    'tc_debug_node_groups': [ctypes.c_int, ctypes.c_int, c_double_p, c_double_p,
                             c_int32_p, c_int32_p, c_int_p, c_int_p],
    'tc_debug_quad_schedule': [ctypes.c_int] * 10 + [c_int_p, c_int_p, c_int_p,
                                                     c_int64_p, c_int64_p],
    'tc_debug_quad_emulate': [ctypes.c_int, ctypes.c_int, c_double_p, c_uint8_p,
                              ctypes.c_int, ctypes.c_int, c_double_p,
                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int,
                              ctypes.c_int, ctypes.c_int, c_double_p],
    'tc_table_create': [
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
        ctypes.c_void_p, ctypes.c_int, c_double_p, c_double_p, c_double_p,
        c_double_p, c_double_p, c_uint8_p, ctypes.c_int, c_void_pp],
    'tc_table_destroy': [ctypes.c_void_p],
    'tc_table_autotune_result': [ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, c_int_p,
                                 c_int64_p, c_int_p, c_float_p],
    'tc_table_batch_invariant': [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, c_int_p],
    'tc_set_copy_threads': [ctypes.c_int, ctypes.c_int],
    'tc_table_resident_stats': [ctypes.c_void_p, c_int64_p, c_int64_p, c_int64_p, c_int_p],
    'tc_table_synchronize': [ctypes.c_void_p],
    'tc_table_info': [ctypes.c_void_p, c_int_p, c_int_p, c_int_p, c_int64_p,
                      c_int_p, c_int64_p],
    'tc_mean_occupation_zheng07_batch': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, c_double_p],
    'tc_predict_zheng07_batch': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p],
    'tc_predict_zheng07_batch_device': [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p],
    'tc_chi2_zheng07_batch': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p, c_double_p,
        c_double_p],
    'tc_chi2_zheng07_batch_device': [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p, ctypes.c_void_p,
        ctypes.c_void_p],
    'tc_predict_zheng07_many': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int,
        ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p],
    'tc_predict_zheng07_joint': [
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, c_double_p, ctypes.c_int,
        ctypes.c_int, ctypes.c_uint, c_double_p, ctypes.POINTER(c_double_p)],
    'tc_predict_zheng07_batch_async': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p, c_int64_p],
    'tc_chi2_zheng07_batch_async': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, ctypes.c_int64,
        ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p, c_double_p,
        c_double_p, c_int64_p],
    'tc_table_wait': [ctypes.c_void_p, ctypes.c_int64],
    'tc_table_query': [ctypes.c_void_p, ctypes.c_int64, c_int_p],
    'tc_interp_predict_zheng07_batch_async': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p,
        c_int64_p],
    'tc_interp_chi2_zheng07_batch_async': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p,
        c_double_p, c_double_p, c_int64_p],
    'tc_interp_wait': [ctypes.c_void_p, ctypes.c_int64],
    'tc_interp_query': [ctypes.c_void_p, ctypes.c_int64, c_int_p],
    'tc_predict_occupation_batch': [
        ctypes.c_void_p, c_double_p, ctypes.c_int64, ctypes.c_uint,
        c_double_p, c_double_p],
    'tc_interp_create': [c_void_pp, ctypes.c_int, ctypes.c_int, c_double_p,
                         c_void_pp],
    'tc_interp_destroy': [ctypes.c_void_p],
    'tc_interp_synchronize': [ctypes.c_void_p],
    'tc_interp_axis': [ctypes.c_void_p, ctypes.c_int, c_int_p, c_double_p,
                       ctypes.c_int],
    'tc_interp_predict_zheng07_batch': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p],
    'tc_interp_predict_zheng07_batch_device': [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p,
        ctypes.c_void_p],
    'tc_interp_chi2_zheng07_batch': [
        ctypes.c_void_p, c_double_p, ctypes.c_int, c_double_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p,
        c_double_p, c_double_p],
    'tc_interp_chi2_zheng07_batch_device': [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_int, ctypes.c_uint, c_double_p, c_double_p,
        ctypes.c_void_p, ctypes.c_void_p],
    'tc_table_set_option': [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int],
    'tc_table_timer_begin': [ctypes.c_void_p, ctypes.c_int],
    'tc_table_timer_end': [ctypes.c_void_p, c_float_p],
    'tc_table_kernel_time': [ctypes.c_void_p, c_int_p, c_float_p],
    'tc_debug_trace': [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                       ctypes.c_int64, c_int64_p],
    'tc_debug_wave_trace': [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                            ctypes.c_int64, c_int64_p],
    'tc_debug_resident_ticks': [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64),
                                ctypes.c_int64, c_int64_p],
    'tc_debug_ensemble_stamps': [ctypes.c_void_p,
                                 ctypes.POINTER(ctypes.c_uint64)],
    'tc_table_last_launch': [ctypes.c_void_p, c_int_p, c_int_p, c_int_p,
                             c_int_p],
    'tc_pair_count_rppi': [c_double_p, ctypes.c_int64, c_double_p,
                           ctypes.c_int64, c_double_p, c_double_p, ctypes.c_int,
                           ctypes.c_double, ctypes.c_int,
                           ctypes.POINTER(ctypes.c_uint64)],
    'tc_pair_count_smu': [c_double_p, ctypes.c_int64, c_double_p,
                          ctypes.c_int64, c_double_p, c_double_p, ctypes.c_int,
                          ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)],
    'tc_pair_count_rppi_labelled': [c_double_p, c_int32_p, ctypes.c_int64,
                                    c_double_p, c_int32_p, ctypes.c_int64,
                                    ctypes.c_int, c_double_p, c_double_p,
                                    ctypes.c_int, ctypes.c_double,
                                    ctypes.POINTER(ctypes.c_uint64)],
    'tc_pair_count_smu_labelled': [c_double_p, c_int32_p, ctypes.c_int64,
                                   c_double_p, c_int32_p, ctypes.c_int64,
                                   ctypes.c_int, c_double_p, c_double_p,
                                   ctypes.c_int, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_uint64)],
    'tc_mass_in_cylinders': [c_double_p, ctypes.c_int64, c_double_p,
                             ctypes.c_int64, c_double_p, c_double_p, c_double_p,
                             ctypes.c_int, c_double_p],
    'tc_comm_unique_id': [ctypes.c_void_p],
    'tc_comm_create': [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                       c_void_pp],
    'tc_comm_destroy': [ctypes.c_void_p],
    'tc_comm_gather': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                       ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                       ctypes.c_int],
    'tc_comm_release': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int],
    'tc_comm_gather_interp': [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                              ctypes.c_int],
    'tc_comm_release_interp': [ctypes.c_void_p, ctypes.c_void_p,
                               ctypes.c_int],
    'tc_comm_barrier': [ctypes.c_void_p],
    'tc_comm_synchronize': [ctypes.c_void_p],
}

_lib = None


class TabCorrHipError(RuntimeError):
    """The HIP backend failed (no device, launch error, RCCL error)."""


def load():
    """Load the shared library (once) and declare every signature."""
    global _lib
    if _lib is not None:
        return _lib
    # developer A/B runs: TABCORR_AMD_LIBRARY points at another build of the library
    library = os.environ.get('TABCORR_AMD_LIBRARY') or LIBRARY
    if not os.path.exists(library):
        raise TabCorrHipError(
            'libtabcorr_hip.so is missing: build it with '
            '`python -m tabcorr_amd.build` (needs hipcc). There is no CPU '
            'fallback.')
    # RTLD_DEEPBIND + -z now: the library keeps the ROCm runtime it was linked
    # against even when another copy (e.g. the one bundled with PyTorch) is
    # already in the process.
    mode = os.RTLD_NOW | os.RTLD_LOCAL | getattr(os, 'RTLD_DEEPBIND', 0)
    lib = ctypes.CDLL(library, mode=mode)
    lib.tc_last_error.restype = ctypes.c_char_p
    lib.tc_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        function = getattr(lib, name)
        function.restype = ctypes.c_int
        function.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    """Map a status code to the exception type the reference would raise."""
    if status == TC_OK:
        return
    message = load().tc_last_error().decode(errors='replace')
    if status == TC_ERR_INVALID:
        raise ValueError(message)
    if status == TC_ERR_UNSUPPORTED:
        raise NotImplementedError(message)
    raise TabCorrHipError(message)


def as_double_p(array):
    """Pointer to the first element of a C-contiguous float64 array for a
    ``c_double_p`` argument.  ``from_buffer`` + ``byref`` costs a third of
    ``array.ctypes.data_as`` (0.5 against 1.5 us, three of them per un-batched
    call); read-only and empty arrays take the slow way."""
    try:
        return ctypes.byref(ctypes.c_double.from_buffer(array))
    except (TypeError, ValueError, BufferError):
        return array.ctypes.data_as(c_double_p)


def device_count():
    lib = load()
    count = ctypes.c_int(0)
    status = lib.tc_device_count(ctypes.byref(count))
    if status != TC_OK:
        return 0
    return count.value


def require_device():
    """Raise unless a HIP device is usable."""
    lib = load()
    count = ctypes.c_int(0)
    check(lib.tc_device_count(ctypes.byref(count)))
    if count.value < 1:
        raise TabCorrHipError('no HIP device found; tabcorr_amd has no CPU '
                              'fallback')
    return count.value


def runtime_version():
    version = ctypes.c_int(0)
    check(load().tc_runtime_version(ctypes.byref(version)))
    return version.value


def device_name():
    buffer = ctypes.create_string_buffer(256)
    check(load().tc_device_name(buffer, 256))
    return buffer.value.decode()


def contiguous(array, dtype=np.float64):
    return np.ascontiguousarray(array, dtype=dtype)
