#!/usr/bin/env python3
"""Runs only the K3 raster kernel on the bench workload (for rocprofv3 and for ablations).
   python tools/profile_raster.py [--batch 256] [--iters 5] [--tw 0] [--debug 0] [--u8]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--agents', type=int, default=64)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--tw', type=int, nargs='*', default=[0])
    ap.add_argument('--debug', type=int, nargs='*', default=[0])
    ap.add_argument('--u8', action='store_true')
    ap.add_argument('--res', type=int, default=0, help='resolution (default: the bench resolution)')
    ap.add_argument('--cell', type=float, default=0.0, help='grid cell size of the rendering map in metres (0: the library default)')
    ap.add_argument('--six-keys', action='store_true', help='two agent types: one more distinct key than the bench scene (6 bit planes)')
    ap.add_argument('--agent-types', type=int, default=1, help='1 .. 6 agent types (vehicle, pedestrian, bicycle, ego, ground_truth, prediction): 5 .. 10 distinct keys')
    ap.add_argument('--bits-waves', type=int, nargs='*', default=[4])
    ap.add_argument('--list-waves', type=int, nargs='*', default=[], help='testing build: waves per workgroup of the list rasteriser to sweep (1, 2, 4; 0 = automatic)')
    ap.add_argument('--lib', default='', help='another build of libtdship_testing.so (path) to load instead')
    ap.add_argument('--list-lds', type=int, nargs='*', default=[], help='testing build: LDS budgets (KiB per workgroup) of the list rasteriser of the split path to sweep')
    ap.add_argument('--no-bits', action='store_true', help='packed-key kernels instead of the bit-plane kernel')
    ap.add_argument('--no-ws', action='store_true', help='fused per-strip kernel instead of the binned persistent kernel')
    ap.add_argument('--steps-before', type=int, default=5, help='simulation steps before rendering (spreads the agents)')
    args = ap.parse_args()
    from torchdrivesim_amd import _native, _ops
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device('cuda', 0)
    # ablation switches / tuning knobs exist only in the testing build; a plain timing or counter run measures the PRODUCT library
    plain = args.tw == [0] and args.debug == [0] and args.bits_waves == [4] and not args.list_lds and not args.lib and not args.list_waves
    if args.lib:
        # another build (e.g. an older round's, for a same-box comparison): entry points it does not have yet are simply not bound
        _native.TESTING_LIB_PATH = os.path.abspath(args.lib)
        probe = ctypes.CDLL(_native.TESTING_LIB_PATH)
        for table in (_native._SIGNATURES, _native._TESTING_SIGNATURES):
            for name in [n for n in table if not hasattr(probe, n)]:
                del table[name]
    L = None
    if not plain:
        L = _native.testing_lib()
        _native._lib = L                 # every call of this process goes through it
    print('library:', 'libtdship.so (product)' if plain else 'libtdship_testing.so')
    _ops.use_workspace = not args.no_ws
    _ops.use_bitplanes = not args.no_bits
    if args.cell > 0:
        _orig = _ops.StaticMap.__init__
        _ops.StaticMap.__init__ = lambda self, *a, **k: _orig(self, *a, **{**k, 'cell_size': args.cell})
    sim, actions, _ = bench.build_simulator(args.batch, args.agents, dev, seed=1234)
    if args.six_keys:
        args.agent_types = 2
    if args.agent_types > 1:
        sim._agent_types = ['vehicle', 'pedestrian', 'bicycle', 'ego', 'ground_truth', 'prediction'][:args.agent_types]
        sim.agent_type = sim.agent_type.clone()
        sim.agent_type[:, ::2] = 1
        if args.agent_types > 2:
            sim.agent_type[:, 1::4] = 2
        for extra in range(3, args.agent_types):
            sim.agent_type[:, extra::8] = extra
        sim._scene_cache = None
    if args.u8:
        sim.renderer.cfg.out_dtype = 'uint8'
    for i in range(args.steps_before):
        sim.step(actions[i % 8])
    res = Resolution(args.res or bench.RES, args.res or bench.RES)
    img = sim.render_egocentric(res=res, fov=bench.FOV)
    torch.cuda.synchronize()
    nbytes = img.numel() * img.element_size()
    print(f'images {img.shape[0] * img.shape[1]}, output {nbytes / 1e9:.2f} GB, nonzero fraction {(img[:8] > 0).float().mean().item():.3f}')
    del img
    for tw, bw, ll, lwv in [(t, b, l, w) for t in args.tw for b in args.bits_waves for l in (args.list_lds or [0]) for w in (args.list_waves or [0])]:
        hooks = L is not None and hasattr(L, 'tds_raster_set_debug')          # (a product build loaded with --lib has none)
        if hooks:
            L.tds_raster_set_bits_waves(bw)
            if ll:
                L.tds_raster_set_list_lds(ll)
            L.tds_raster_set_list_waves(lwv)
        for dbg in args.debug:
            if hooks:
                L.tds_raster_set_strip_width(tw)
                L.tds_raster_set_debug(dbg)
            _ops.raster_events = []
            for _ in range(args.iters):
                sim.render_egocentric(res=res, fov=bench.FOV)
            torch.cuda.synchronize()
            ms = np.array([a.elapsed_time(b) for a, b in _ops.raster_events])
            _ops.raster_events = None
            print(f'tw={tw:3d} waves={bw} list-lds={ll:3d} list-waves={lwv} debug={dbg:5d}: {ms.min():8.3f} ms min, {np.median(ms):8.3f} ms median -> {nbytes / np.median(ms) / 1e6:8.1f} GB/s '
                  f'({nbytes / np.median(ms) / 1e6 / 80:.1f}% of 8 TB/s)')
    if L is not None and hasattr(L, 'tds_raster_set_debug'):
        L.tds_raster_set_strip_width(0)
        L.tds_raster_set_debug(0)


if __name__ == '__main__':
    main()
