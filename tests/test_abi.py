"""CPU-only checks of the drop-in boundary: libtdship.so loads and exports every symbol include/tdship.h declares
(no compute calls -- there is no GPU here), and argument validation fails loudly."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'tdship.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(tds_\w+)\s*\(', src)))


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


def test_errors_are_loud_without_touching_the_gpu():
    from torchdrivesim_amd import _native
    L = _native.lib()
    # negative size -> TDS_EINVAL with a message; no kernel is launched
    rc = L.tds_bicycle_step_f32(None, None, None, None, -1, 0.1, 5.0, 1.5, 0, 0, None)
    assert rc == -1 and 'out of range' in _native.last_error()
    rc = L.tds_raster_set_strip_width(5)
    assert rc == -1
    with pytest.raises(RuntimeError):
        _native.check(rc, 'tds_raster_set_strip_width')
    assert L.tds_raster_set_strip_width(0) == 0


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
