"""
The register / scratch budgets the BENCH numbers rest on, asserted on the BUILT product (tools/kernel_resources.py reads the AMDGPU
metadata of the code objects inside torchdrivesim_amd/lib/libtdship.so).  VERDICT r5, weak 7: the K3s speed-up rests on how one line of
scan_init is written (raster.hip, "through the launch's (always zero) debug word": 112 instead of 368 bytes of scratch per lane, 0.73 instead of
0.98 ms) and the headline kernel lives at 168 VGPRs / three waves per SIMD -- a ROCm point release or an innocent edit must not cost a mode
25 % silently.  Runs in the CPU suite: hipcc cross-compiles, llvm-readelf reads the notes, no GPU involved.
"""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


@pytest.fixture(scope='module')
def table():
    import kernel_resources
    from torchdrivesim_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build()
    t = kernel_resources.kernel_table(_native.LIB_PATH)
    assert len(t) > 100, 'the code objects of the library were not found'
    return t


def matching(table, pattern):
    out = {k: v for k, v in table.items() if re.fullmatch(pattern, k)}
    assert out, f'no kernel matches {pattern}'
    return out


def test_headline_rasteriser_keeps_three_waves_per_simd(table):
    """raster_scene_bits_kernel<4 waves, index bits, out type, args, EMIT, MINWG = 3>: the persistent instantiations (float32 256 x 256 headline:
    <4, 3, float, SceneArgs, false, 3>) -- at most 168 VGPRs (three waves per SIMD) and 80 bytes of scratch per lane"""
    head = table['raster_scene_bits_kernel<4, 3, float, SceneArgs, false, 3>']
    assert head['waves_per_simd'] >= 3 and head['private_segment_fixed_size'] <= 80 and head['vgpr_count'] <= 168, head
    for name, e in matching(table, r'raster_scene_bits_kernel<4, \d, (float|unsigned char), SceneArgs(Ex)?, (false|true), 3>').items():
        assert e['waves_per_simd'] >= 3 and e['private_segment_fixed_size'] <= 80, (name, e)
    # the 8-wave instantiations (six or seven keys at 256 x 256: two workgroups of eight waves per CU = four waves per SIMD)
    for name, e in matching(table, r'raster_scene_bits_kernel<8, \d, (float|unsigned char), SceneArgsEx, false, 3>').items():
        assert e['waves_per_simd'] >= 4 and e['private_segment_fixed_size'] <= 176, (name, e)
    # the instantiations for four workgroups per CU (LDS allows it): 128 VGPRs
    for name, e in matching(table, r'raster_scene_bits_kernel<4, \d, (float|unsigned char), SceneArgs(Ex)?, false, 4>').items():
        assert e['waves_per_simd'] >= 4 and e['private_segment_fixed_size'] <= 192, (name, e)


def test_scan_kernel_of_the_split_form_stays_at_its_scratch(table):
    """K3s: both instantiations (without / with per-camera triangles) at 112 bytes of scratch and five waves per SIMD"""
    for name, e in matching(table, r'scan_faces_kernel<SceneArgs(Ex)?>').items():
        assert e['private_segment_fixed_size'] <= 112 and e['waves_per_simd'] >= 5, (name, e)


def test_kernels_that_must_not_spill(table):
    for pattern, waves in ((r'raster_list_bits_kernel<\d, (float|unsigned char), \d>', 5), (r'raster_scene_bwd_idx_kernel<\d>', 5), (r'offroad_kernel', 6),
                           (r'offroad_bwd_kernel', 4), (r'collision_scene_bwd_kernel<1>', 8), (r'bicycle_step_kernel', 8), (r'wrong_way_kernel', 4)):
        for name, e in matching(table, pattern).items():
            assert e['private_segment_fixed_size'] == 0 and e['vgpr_spill_count'] == 0 and e['waves_per_simd'] >= waves, (name, e)


def test_waves_per_simd_rule():
    import kernel_resources as kr
    assert [kr.waves_per_simd(v) for v in (168, 169, 128, 129, 96, 97, 80, 81, 64, 65, 24)] == [3, 2, 4, 3, 5, 4, 6, 5, 8, 7, 8]
