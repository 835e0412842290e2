"""CPU-only checks of the drop-in boundary: libtdship.so loads and exports every symbol include/tdship.h declares
(no compute calls -- there is no GPU here), and argument validation fails loudly."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _header(testing):
    src = open(os.path.join(ROOT, 'include', 'tdship.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    hooks = re.search(r'#ifdef TDS_TESTING\n(.*?)#endif', src, flags=re.S)
    assert hooks, 'include/tdship.h has no "testing hooks" section'
    return hooks.group(1) if testing else src.replace(hooks.group(0), '')


def declared_symbols(testing=False):
    return sorted(set(re.findall(r'\b(?:int|void)\s*\*?\s*(tds_\w+)\s*\(', _header(testing))))


def exported_symbols(path):
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True, check=True).stdout
    return sorted(ln.split()[-1] for ln in out.splitlines() if ' T tds_' in ln)


def test_header_symbols_are_exported_and_bound():
    from torchdrivesim_amd import _native
    _native.build()
    L = _native.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/tdship.h but not exported by libtdship.so'
        assert n in _native._SIGNATURES, f'{n} has no ctypes signature in _native.py'
    assert set(_native._SIGNATURES) <= set(names)
    assert L.tds_version() == 1
    # the product exports exactly what the header declares outside its testing section
    assert exported_symbols(_native.LIB_PATH) == names


def test_testing_hooks_are_not_in_the_product():
    """ablation switches, work counters, tuning knobs and the switch that disables K2b's candidate lists live in
    libtdship_testing.so only; the product library neither exports them nor reads the environment."""
    from torchdrivesim_amd import _native
    _native.build()
    hooks = declared_symbols(testing=True)
    assert set(hooks) == set(_native._TESTING_SIGNATURES) and len(hooks) >= 5
    product = exported_symbols(_native.LIB_PATH)
    assert not set(hooks) & set(product)
    assert exported_symbols(_native.TESTING_LIB_PATH) == sorted(set(product) | set(hooks))
    T = _native.testing_lib()
    for n in hooks:
        assert hasattr(T, n)
    blob = open(_native.LIB_PATH, 'rb').read()
    assert b'TDS_NO_NEAR_LISTS' not in blob
    import subprocess
    undefined = subprocess.run(['nm', '-D', '--undefined-only', _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert ' getenv' not in undefined, 'the product library must not read the environment'


def test_errors_are_loud_without_touching_the_gpu():
    from torchdrivesim_amd import _native
    L = _native.lib()
    # negative size -> TDS_EINVAL with a message; no kernel is launched
    rc = L.tds_bicycle_step_f32(None, None, None, None, -1, 0.1, 5.0, 1.5, 0, 0, None)
    assert rc == -1 and 'out of range' in _native.last_error()
    with pytest.raises(RuntimeError):
        _native.check(rc, 'tds_bicycle_step_f32')
    T = _native.testing_lib()
    assert T.tds_raster_set_strip_width(5) == -1 and T.tds_raster_set_strip_width(0) == 0


def test_cpu_tensors_are_rejected():
    import torch
    from torchdrivesim_amd import _ops
    s = torch.zeros(2, 3, 4)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _ops.bicycle_step(s, torch.zeros(2, 3, 2), torch.ones(2, 3))


def _build_c_caller(out_dir):
    """tests/abi_smoke.c with plain gcc against include/tdship.h and libtdship.so: the header is valid C, every entry point links"""
    import subprocess
    from torchdrivesim_amd import _native
    _native.build()
    exe = os.path.join(str(out_dir), 'abi_smoke')
    lib_dir = os.path.dirname(_native.LIB_PATH)
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'abi_smoke.c'), '-o', exe, '-L' + lib_dir, '-ltdship', '-L/opt/rocm/lib', '-lamdhip64', '-lm',
                           '-Wl,-rpath,' + lib_dir, '-Wl,-rpath,/opt/rocm/lib'])
    return exe


def test_a_plain_c_program_builds_against_the_header(tmp_path):
    assert os.path.exists(_build_c_caller(tmp_path))


@pytest.mark.gpu
def test_a_plain_c_program_runs_through_the_abi(tmp_path):
    """no Python and no torch in the process: kinematics known answer, an IoU scene, and the error path"""
    import subprocess
    out = subprocess.run([_build_c_caller(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 'abi smoke ok' in out.stdout, out.stdout + out.stderr


def test_no_per_call_entry_point_allocates_or_synchronises():
    """SURVEY 8(b) "Ownership": the library never allocates per call -- device memory is owned by the explicit create / destroy handles
    (maps, lane tables, output buffers) -- and never synchronises, so every per-step call can be captured into a HIP graph
    (tests/test_gpu_graph.py).  The kernel sources must not even mention an allocation or a blocking call (the work
    queues of the persistent raster launch used to live in a pool the library allocated lazily inside tds_raster_scene)."""
    csrc = os.path.join(ROOT, 'torchdrivesim_amd', 'csrc')
    banned = ('hipMalloc', 'hipFree', 'hipMemset', 'hipMemcpy(', 'hipMemcpyAsync', 'hipStreamSynchronize', 'hipDeviceSynchronize', 'hipEventSynchronize', 'hipHostMalloc')
    for name in ('raster.hip', 'raster_bwd.hip', 'collision.hip', 'backward.hip', 'kinematic.hip'):
        src = re.sub(r'//.*', '', open(os.path.join(csrc, name)).read())
        for word in banned:
            assert word not in src, f'{name} mentions {word}'
    # where allocations do live: create / destroy functions only
    for name in ('map.hip', 'lanes.hip', 'alloc.hip'):
        src = open(os.path.join(csrc, name)).read()
        assert 'hipMalloc' in src or 'hipMemCreate' in src
