#!/usr/bin/env python3
"""
Opcode histogram of the LOOPS of a compiled kernel: which instructions do the hot loops of the instruction-bound raster
modes consist of, and which of them cost more than one issue slot?

    python tools/opcode_histogram.py --kernel 'raster_scene_bits_kernel<4, 3, unsigned char, SceneArgs, false, 3>' [--lib PATH] [--min-insts 20]

Reads the code object out of the built library (tools/kernel_resources.py), disassembles the kernel with llvm-objdump, finds the loops from the
backward branches (a branch whose target address is not above its own: the body is [target, branch]) and prints, per INNERMOST loop in address
order: instruction count by class, the multi-cycle instructions by opcode, and a few landmark counts (ds_or / LDS atomics, readlane, DPP,
bpermute, stores) by which a loop can be matched to its source (process_batch_bits' row items, the edge walks, the grid scan, write_out_bits).
Trip counts are not in the binary: tools/raster_stats.py (work counters of the testing build) gives rounds per image for the loops it counts;
profiles/r06_opcode_histogram_u8.txt combines the two.

Issue cost classes (MI355X_MICROARCH.md / CDNA3 ISA guide; a wave64 VALU instruction issues over 4 cycles on a 16-lane SIMD):
  full   32-bit VALU: add, sub, and, or, xor, shifts, min / max, cndmask, mov, perm, cmp, 24-bit mul / mad, fma / mul / add f32, packed f32
  quarter  v_mul_lo_u32 / v_mul_hi_u32 / v_mul_lo_i32 / v_mad_u64_u32 (4 x), 64-bit shifts and adds (2 x: v_lshlrev_b64, v_ashrrev_i64, v_add_co + addc pairs),
           transcendental f32 (v_rcp / v_rsq / v_sqrt / v_exp / v_log / v_sin / v_cos: 4 x), f64 arithmetic (2 - 4 x), v_div_scale / fmas / fixup (part of a division)
  cross    v_readlane / v_writelane / v_readfirstlane, DPP moves, ds_bpermute / ds_permute / ds_swizzle (LDS crossbar)
  lds      ds_read* / ds_write* / ds_or / ds_max / ds_add ... (LDS instructions; those with RTN return a value)
  mem      global_* / buffer_* / scratch_* / flat_*
  scalar   s_* (another issue port)
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import kernel_resources as kr                                      # noqa: E402

QUARTER = re.compile(r'^(v_mul_lo_u32|v_mul_hi_u32|v_mul_lo_i32|v_mul_hi_i32|v_mad_u64_u32|v_mad_i64_i32|v_rcp_|v_rsq_|v_sqrt_|v_exp_|v_log_|v_sin_|v_cos_|'
                     r'v_rcp_iflag|v_div_scale|v_div_fmas|v_div_fixup|v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64|v_.*_f64|v_cvt_f64|v_cvt_.*_f64|v_trig_preop)')
CROSS = re.compile(r'^(v_readlane|v_writelane|v_readfirstlane|ds_bpermute|ds_permute|ds_swizzle|v_permlane)')


def classify(op, text):
    if op.startswith('s_'):
        return 'scalar'
    if CROSS.match(op) or 'dpp' in op or ' row_' in text or 'quad_perm' in text or 'wave_shr' in text or 'row_bcast' in text:
        return 'cross'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')):
        return 'mem'
    if QUARTER.match(op):
        return 'quarter'
    if op.startswith('v_'):
        return 'full'
    return 'other'


def disassemble(lib, kernel):
    table = kr.kernel_table(lib)
    if kernel not in table:
        raise SystemExit(f'no kernel {kernel!r}; e.g. ' + ', '.join(sorted(k for k in table if 'raster' in k)[:6]))
    for blob in kr.code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix='.o') as f:
            f.write(blob)
            f.flush()
            syms = subprocess.check_output([os.path.join(kr.LLVM, 'llvm-objdump'), '-t', f.name], text=True)
            names = [ln.split()[-1] for ln in syms.splitlines() if ' F .text' in ln or ' .text' in ln]
            dem = subprocess.check_output(['c++filt'], input='\n'.join(names), text=True).splitlines()
            for sym, nice in zip(names, dem):
                nice = re.sub(r'\(anonymous namespace\)::', '', re.sub(r'^void ', '', nice))
                if nice.startswith(kernel + '('):
                    return subprocess.check_output([os.path.join(kr.LLVM, 'llvm-objdump'), '-d', f'--disassemble-symbols={sym}', f.name], text=True)
    raise SystemExit(f'{kernel}: symbol not found in any code object')


def parse(asm):
    """[(address, opcode, full text)]"""
    out = []
    for ln in asm.splitlines():
        m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):', ln)
        if m:
            out.append((int(m.group(3), 16), m.group(1), (m.group(1) + ' ' + m.group(2)).strip()))
    return out


def loops(insts):
    """innermost loops as (start index, end index) from backward branches"""
    addr_to_i = {a: i for i, (a, _, _) in enumerate(insts)}
    found = []
    for i, (a, op, text) in enumerate(insts):
        if op.startswith(('s_cbranch', 's_branch')):
            m = re.match(r'^\S+\s+(-?\d+)', text)
            if not m:
                continue
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            target = a + 4 + 4 * off
            if target <= a and target in addr_to_i:
                found.append((addr_to_i[target], i))
    inner = [lp for lp in found if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in found)]
    return sorted(set(inner)), sorted(set(found))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='raster_scene_bits_kernel<4, 3, unsigned char, SceneArgs, false, 3>')
    ap.add_argument('--lib', default=os.path.join(ROOT, 'torchdrivesim_amd', 'lib', 'libtdship.so'))
    ap.add_argument('--min-insts', type=int, default=24, help='loops whose own body is shorter than this are listed in one line')
    args = ap.parse_args()
    insts = parse(disassemble(args.lib, args.kernel))
    inner, every = loops(insts)
    total = collections.Counter(classify(op, t) for _, op, t in insts)
    print(f'# {args.kernel}: {len(insts)} instructions, {len(every)} loops ({len(inner)} innermost); whole kernel by class: {dict(total)}')
    print('# per loop, in address order: its OWN body (instructions inside it but inside none of the loops nested in it), by issue class; landmarks; the')
    print('# multi-cycle and cross-lane opcodes; the most frequent opcodes.  depth = loops around it.')
    base = insts[0][0]
    for n, (s, e) in enumerate(every):
        nested = [o for o in every if o != (s, e) and s <= o[0] and o[1] <= e]
        own = [insts[i] for i in range(s, e + 1) if not any(o[0] <= i <= o[1] for o in nested)]
        cls = collections.Counter(classify(op, t) for _, op, t in own)
        ops = collections.Counter(op for _, op, _ in own)
        marks = dict(ds_or=sum(v for k, v in ops.items() if k.startswith('ds_or')), lds_rtn=sum(v for k, v in ops.items() if k.startswith('ds_') and 'rtn' in k),
                     lds_read=sum(v for k, v in ops.items() if k.startswith('ds_read')), lds_write=sum(v for k, v in ops.items() if k.startswith('ds_write')),
                     readlane=ops.get('v_readlane_b32', 0), bpermute=ops.get('ds_bpermute_b32', 0), dpp=sum(1 for _, op, t in own if 'dpp' in op or 'row_' in t),
                     perm=ops.get('v_perm_b32', 0), stores=sum(v for k, v in ops.items() if 'store' in k and not k.startswith('scratch')),
                     loads=sum(v for k, v in ops.items() if k.startswith(('global_load', 'buffer_load'))), scratch=sum(v for k, v in ops.items() if k.startswith('scratch_')),
                     barrier=ops.get('s_barrier', 0))
        depth = sum(1 for o in every if o != (s, e) and o[0] <= s and e <= o[1])
        head = (f'loop {n:3d}  +0x{insts[s][0] - base:05x}..+0x{insts[e][0] - base:05x}  own {len(own):4d} of {e - s + 1:5d} insts  depth {depth:2d}  nested {len(nested):2d}  ' +
                '  '.join(f'{k} {cls.get(k, 0)}' for k in ('full', 'quarter', 'cross', 'lds', 'mem', 'scalar')))
        print(head)
        if len(own) < args.min_insts:
            continue
        print('          landmarks: ' + '  '.join(f'{k} {v}' for k, v in marks.items() if v))
        slow = {k: v for k, v in ops.items() if classify(k, k) in ('quarter', 'cross') or (k.startswith('ds_') and 'rtn' in k)}
        if slow:
            print('          multi-cycle / cross-lane: ' + '  '.join(f'{k} x{v}' for k, v in sorted(slow.items(), key=lambda kv: -kv[1])))
        print('          top opcodes: ' + '  '.join(f'{k} x{v}' for k, v in ops.most_common(14)))


if __name__ == '__main__':
    sys.exit(main())
