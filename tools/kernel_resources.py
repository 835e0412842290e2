#!/usr/bin/env python3
"""
Register / scratch / LDS budget of every kernel of the BUILT product, read from the code objects inside libtdship.so (not from compiler
remarks of some other compile): the numbers of BENCH depend on a few kernels staying at their occupancy -- the headline rasteriser at three
waves per SIMD with 80 bytes of scratch, K3s at 112 bytes because of how one line of scan_init is written -- and a ROCm point release or an
innocent edit can change that silently.  tests/test_kernel_resources.py asserts the budgets in the CPU suite.

    python tools/kernel_resources.py [--lib PATH] [--out resource_usage.json]

How: the `.hip_fatbin` section of the shared library holds one clang offload bundle per translation unit; each bundle's gfx950 entry is an
ELF code object whose AMDGPU metadata note lists, per kernel, .vgpr_count, .agpr_count, .sgpr_count, .private_segment_fixed_size (scratch bytes
per lane), .vgpr_spill_count, .sgpr_spill_count and .group_segment_fixed_size (static LDS).  Waves per SIMD follow from the 512-entry unified
register file of a gfx950 SIMD lane (allocation granule 8, at most 8 waves).
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'private_segment_fixed_size', 'vgpr_spill_count', 'sgpr_spill_count', 'group_segment_fixed_size',
          'max_flat_workgroup_size', 'wavefront_size')


def code_objects(lib_path, arch='gfx950'):
    """the `arch` code objects (bytes) of a shared library built by hipcc"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', lib_path, os.path.join(tmp, 'copy.so')])
        data = open(fat, 'rb').read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        o = i + len(MAGIC)
        count, = struct.unpack_from('<Q', data, o)
        o += 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from('<QQQ', data, o)
            o += 24
            triple = data[o:o + tlen].decode()
            o += tlen
            if triple.endswith(arch) and size:
                out.append(data[i + off:i + off + size])
        pos = i + 1
    return out


def waves_per_simd(vgprs, agprs=0):
    """gfx950: 512 unified registers per SIMD lane, allocated in granules of 8, at most 8 waves"""
    total = max(vgprs, 1)
    if agprs:
        total = ((vgprs + 3) // 4) * 4 + agprs          # the accumulation registers start at a multiple of 4
    alloc = ((total + 7) // 8) * 8
    return min(8, 512 // alloc)


def kernel_table(lib_path):
    """{demangled kernel name: {field: int, ..., 'waves_per_simd': int}} over every gfx950 kernel of the library"""
    import yaml
    raw = {}
    for blob in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix='.o') as f:
            f.write(blob)
            f.flush()
            notes = subprocess.check_output([os.path.join(LLVM, 'llvm-readelf'), '--notes', f.name], text=True)
        # the note is a YAML document between `---` and `...`
        for doc in re.findall(r'^\s*---\s*$(.*?)^\s*\.\.\.\s*$', notes, flags=re.S | re.M):
            meta = yaml.safe_load(doc)
            for k in (meta or {}).get('amdhsa.kernels', []):
                raw[k['.name']] = {f: int(k.get('.' + f, 0)) for f in FIELDS}
    names = list(raw)
    demangled = subprocess.check_output(['c++filt'], input='\n'.join(names), text=True).splitlines()
    out = {}
    for sym, nice in zip(names, demangled):
        e = raw[sym]
        e['waves_per_simd'] = waves_per_simd(e['vgpr_count'], e['agpr_count'])
        nice = re.sub(r'\(anonymous namespace\)::', '', nice)
        nice = re.sub(r'^void ', '', nice)
        # `name<template args>(parameters)` -> `name<template args>`; plain `name(parameters)` -> `name`
        depth, cut = 0, len(nice)
        for i, ch in enumerate(nice):
            if ch == '<':
                depth += 1
            elif ch == '>':
                depth -= 1
            elif ch == '(' and depth == 0:
                cut = i
                break
        out[nice[:cut]] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--lib', default=os.path.join(ROOT, 'torchdrivesim_amd', 'lib', 'libtdship.so'))
    ap.add_argument('--out', default=None)
    ap.add_argument('--grep', default=None, help='only kernels whose name contains this')
    args = ap.parse_args()
    table = kernel_table(args.lib)
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(table, f, indent=1, sort_keys=True)
    for name in sorted(table):
        if args.grep and args.grep not in name:
            continue
        e = table[name]
        print(f"{e.get('vgpr_count', 0):4d} VGPR {e.get('agpr_count', 0):3d} AGPR {e.get('vgpr_spill_count', 0):4d} spilled {e.get('private_segment_fixed_size', 0):5d} B scratch "
              f"{e.get('group_segment_fixed_size', 0):6d} B LDS {e['waves_per_simd']} waves  {name}")


if __name__ == '__main__':
    sys.exit(main())
