"""
ctypes binding of libtdship.so (C ABI: include/tdship.h).  The library is built in-tree by
``torchdrivesim_amd/csrc/Makefile`` (``python -c "import __graft_entry__ as g; g.build()"``).

There is deliberately NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import contextlib
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libtdship.so')
#: the same sources built with -DTDS_TESTING: the product plus the hooks of the header's "testing hooks" section (tools/, tests/ only)
TESTING_LIB_PATH = os.path.join(_HERE, 'lib', 'libtdship_testing.so')
CSRC = os.path.join(_HERE, 'csrc')

METRIC_IOU, METRIC_DISCS = 0, 1
OUT_F32, OUT_U8 = 0, 1
RASTER_NO_TRIM = 1

_lib = None


class RasterAux(ctypes.Structure):
    """tds_raster_aux_t (include/tdship.h): optional outputs of tds_raster_scene for a later backward pass"""
    _fields_ = [('index_slices', ctypes.c_void_p), ('index_slices_bytes', ctypes.c_int64), ('keys', ctypes.c_uint32 * 16),
                ('n_keys', ctypes.c_int32), ('index_bits', ctypes.c_int32), ('flags', ctypes.c_int32)]


_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> argtypes  (mirrors include/tdship.h; tests/test_abi.py checks that every declared symbol is exported)
_SIGNATURES = {
    'tds_version': [],
    'tds_last_error': [ctypes.c_char_p, ctypes.c_size_t],
    'tds_bicycle_step_f32': [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _i32, _vp],
    'tds_bicycle_step_bwd_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _i32, _i32, _vp],
    'tds_simple_step_f32': [_vp, _vp, _vp, _i64, _f32, ctypes.POINTER(_f32), _i32, _vp],
    'tds_simple_step_bwd_f32': [_vp, _vp, _vp, _vp, _vp, _i64, _f32, ctypes.POINTER(_f32), _i32, _vp],
    'tds_unicycle_step_f32': [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp],
    'tds_unicycle_step_bwd_f32': [_vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp],
    'tds_collision_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    'tds_collision_bwd_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    'tds_overlap_count_f32': [_vp, _vp, _vp, _vp, _i64, _i64, _vp],
    'tds_pairwise_overlap_f32': [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    'tds_pairwise_discs_f32': [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    'tds_box2corners_f32': [_vp, _vp, _vp, _i64, _vp],
    'tds_occlusion_mask_f32': [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp],
    'tds_map_create': [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i32, _f32, ctypes.POINTER(_vp)],
    'tds_map_destroy': [_vp],
    'tds_map_info': [_vp, ctypes.POINTER(_i64)],
    'tds_map_info_ex': [_vp, ctypes.POINTER(_i64), _i32],
    'tds_rows_hash_u64': [_vp, _i64, _i64, _i64, ctypes.c_uint64, _vp, _vp],
    'tds_rows_equal_u8': [_vp, _i64, _i64, _i64, _vp, _vp, _vp],
    'tds_mapset_create': [ctypes.POINTER(_vp), _i32, ctypes.POINTER(_vp)],
    'tds_mapset_destroy': [_vp],
    'tds_offroad_multi_f32': [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    'tds_offroad_multi_bwd_f32': [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    'tds_raster_scene_multi': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp],
    'tds_offroad_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    'tds_offroad_bwd_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    'tds_raster_scene': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _i32, _vp, _vp, _i64, _vp, _i32, _i32, _vp, _vp, _i64, _vp, _vp],
    'tds_raster_index_slices_bytes': [_i64, _i32, ctypes.POINTER(_i64)],
    'tds_raster_scene_bwd_idx_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _i64, _i64, _i64, _f32, _i32, _vp, _vp, _vp, _vp, _vp],
    'tds_raster_scene_bwd_f32': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f32, _i32, _vp, _vp, _vp, _vp],
    'tds_raster_scene_workspace_bytes': [_i64, _i32, ctypes.POINTER(_i64)],
    'tds_raster_scene_workspace_bytes_for': [_i64, _i32, _i32, _i32, ctypes.POINTER(_i64)],
    'tds_map_keys': [_vp, _vp, _i32, ctypes.POINTER(_i32)],
    'tds_mapset_keys': [_vp, _vp, _i32, ctypes.POINTER(_i32)],
    'tds_raster_mesh': [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _i32, _f32, _i32, _i32, _vp, _i32, _vp],
    'tds_buffer_create': [_i64, _i32, _i32, ctypes.POINTER(_vp)],
    'tds_buffer_ptr': [_vp],
    'tds_buffer_info': [_vp, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i32)],
    'tds_buffer_destroy': [_vp],
    'tds_torch_alloc': [ctypes.c_size_t, _i32, _vp],
    'tds_torch_free': [_vp, ctypes.c_size_t, _i32, _vp],
    'tds_stream_create': [_i32, _vp, _i32, ctypes.POINTER(_vp)],
    'tds_stream_destroy': [_i32, _vp],
    'tds_device_cu_count': [_i32, ctypes.POINTER(_i32)],
    'tds_stream_places': [_vp, _vp, _i32],
    'tds_lanelet_centerline_f64': [_vp, _i32, _vp, _i32, _vp, ctypes.POINTER(_i32)],
    'tds_lanes_create': [_vp, _vp, _vp, _vp, _vp, _i32, _f32, _f32, ctypes.POINTER(_vp)],
    'tds_lanes_destroy': [_vp],
    'tds_lanes_info': [_vp, ctypes.POINTER(_i64)],
    'tds_laneset_create': [ctypes.POINTER(_vp), _i32, ctypes.POINTER(_vp)],
    'tds_laneset_destroy': [_vp],
    'tds_wrong_way_f32': [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp],
    'tds_lanelet_directions_f64': [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _f32, _vp],
}


#: entry points that do not return an error code
_RESTYPES = {'tds_buffer_ptr': _vp, 'tds_torch_alloc': _vp, 'tds_torch_free': None}

#: entry points that only libtdship_testing.so exports (include/tdship.h, "testing hooks")
_TESTING_SIGNATURES = {
    'tds_raster_set_strip_width': [_i32],
    'tds_raster_set_bits_waves': [_i32],
    'tds_raster_set_list_lds': [_i32],
    'tds_raster_set_list_waves': [_i32],
    'tds_raster_set_debug': [_i32],
    'tds_raster_get_stats': [ctypes.POINTER(ctypes.c_ulonglong)],
    'tds_testing_set_near_lists': [_i32],
}


def build(force=False):
    """Compile every HIP source for gfx950 into torchdrivesim_amd/lib/libtdship.so (hipcc cross-compiles without a GPU)."""
    cmd = ['make', '-C', CSRC, '-j', str(min(8, os.cpu_count() or 1))]
    if force:
        cmd.append('-B')
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


def _load(path, signatures):
    if not os.path.exists(path):
        raise RuntimeError(
            f'{path} is missing: the HIP extension has not been built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc). There is no CPU fallback.')
    try:
        L = ctypes.CDLL(path)
    except OSError as e:
        raise RuntimeError(f'cannot load {path}: {e}') from e
    for name, argtypes in signatures.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    return L


def lib():
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, _SIGNATURES)
    return _lib


_testing_lib = None


def testing_lib():
    """libtdship_testing.so (tools/ and tests/ only; the product never loads it)"""
    global _testing_lib
    if _testing_lib is None:
        _testing_lib = _load(TESTING_LIB_PATH, {**_SIGNATURES, **_TESTING_SIGNATURES})
    return _testing_lib


@contextlib.contextmanager
def testing():
    """Route every call of this process through the testing build for the duration of the block (handles created inside must be
    used and dropped inside: the two libraries are separate images).  Yields the testing library for its hooks."""
    global _lib
    saved, _lib = _lib, testing_lib()
    try:
        yield _lib
    finally:
        _lib = saved


def last_error():
    buf = ctypes.create_string_buffer(512)
    lib().tds_last_error(buf, 512)
    return buf.value.decode(errors='replace')


E_INVAL, E_HIP, E_NOMEM, E_LIMIT = -1, -2, -3, -4          # TDS_EINVAL, TDS_EHIP, TDS_ENOMEM, TDS_ELIMIT of include/tdship.h
BUFFER_DENSE = 1


class TdsError(RuntimeError):
    """A C-ABI entry point returned a negative code (`.code`; the message is tds_last_error's).  A RuntimeError, the type the reference
    handles around rendering (rendering/base.py:190-201)."""

    def __init__(self, what, code, message):
        super().__init__(f'{what} failed (code {code}): {message}')
        self.code = code


def check(rc, what):
    if rc != 0:
        raise TdsError(what, rc, last_error())


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def dev_ptr(t, dtype, name):
    """Device pointer of a dense tensor; refuses anything the kernels cannot read as-is."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f'{name}: torchdrivesim_amd kernels run on an MI355X; got a {t.device} tensor (no CPU fallback)')
    if t.dtype != dtype:
        raise RuntimeError(f'{name}: expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise RuntimeError(f'{name}: tensor must be contiguous')
    return ctypes.c_void_p(t.data_ptr())


def call(name, device, *args):
    """Run a C-ABI entry point with `device` current (the stream argument must belong to it)."""
    with torch.cuda.device(device):
        rc = getattr(lib(), name)(*args)
    check(rc, name)
