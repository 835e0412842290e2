#!/usr/bin/env python3
"""When does each XCD start and finish its eighth of the headline raster launch, per output allocation?  (testing build, debug flag 4096:
wall clock of the first workgroup's start and the last wave's end per XCD.)  A slow allocation could be one XCD lagging or all eight being slower.
   python tools/xcd_finish_times.py [--buffers 5]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--buffers', type=int, default=5)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--debug', type=int, default=0, help='further ablation flags of the testing build (1|2: no rasterisation, 4: no store)')
    ap.add_argument('--u8', action='store_true')
    args = ap.parse_args()
    from torchdrivesim_amd import _native, _ops
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device('cuda', 0)
    L = _native.testing_lib()
    _native._lib = L
    B, A = args.batch, 64
    sim, actions, _ = bench.build_simulator(B, A, dev, seed=1234)
    for i in range(5):
        sim.step(actions[i % 8])
    res = Resolution(bench.RES, bench.RES)
    if args.u8:
        sim.renderer.cfg.out_dtype = 'uint8'
    bufs = [torch.empty((B, A, 3, bench.RES, bench.RES), dtype=torch.uint8 if args.u8 else torch.float32, device=dev) for _ in range(args.buffers)]
    st = (ctypes.c_ulonglong * 16)()
    for bi, buf in enumerate(bufs):
        for _ in range(2):
            sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
        torch.cuda.synchronize()
        L.tds_raster_get_stats(st)
        L.tds_raster_set_debug(4096 | args.debug)
        _ops.raster_events = []
        sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
        torch.cuda.synchronize()
        ms = _ops.raster_events[0][0].elapsed_time(_ops.raster_events[0][1])
        _ops.raster_events = None
        L.tds_raster_set_debug(0)
        L.tds_raster_get_stats(st)
        end = np.array([st[i] for i in range(8)], dtype=np.float64)
        start = np.array([(~st[8 + i]) & 0xffffffffffffffff for i in range(8)], dtype=np.float64)
        t0 = start.min()
        print(f'buffer {bi} ptr {buf.data_ptr():x}: launch {ms:.2f} ms; per XCD start (us after the first) ' + ' '.join(f'{(s - t0) / 100:.0f}' for s in start) +
              ' | finish (ms after the first start) ' + ' '.join(f'{(e - t0) / 1e5:.2f}' for e in end), flush=True)


if __name__ == '__main__':
    main()
