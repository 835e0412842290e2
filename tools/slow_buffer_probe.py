#!/usr/bin/env python3
"""Where inside a 51.5 GB output allocation does the headline raster kernel lose its time?  (DESIGN_HISTORY.md section 4: an allocation is
persistently "fast", 7.2 ms, or "slow", 8.4 ms, for this kernel while fill_ takes 7.45 ms on both.)

For each of N float32 output tensors of the bench shape, allocated one after the other in a fresh process (optionally behind a filler
allocation of --filler-gb that shifts where they land):
  * the whole launch (render_egocentric(out=buffer)), and torch's fill_ over it;
  * the same render in CHUNKS of --chunk scenes into consecutive slices of the buffer, each chunk timed by itself: shows whether
    the loss is spread over the allocation or sits in one address range of it;
  * optionally (--stream-only) the kernel with rasterisation switched off (testing build: clear + stream-out only).
   python tools/slow_buffer_probe.py [--buffers 5] [--chunk 64] [--filler-gb 0] [--stream-only]"""
import argparse
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
    ap.add_argument('--chunk', type=int, default=64, help='scenes per chunk of the chunked pass')
    ap.add_argument('--filler-gb', type=float, default=0.0, help='an allocation of this size made BEFORE the output tensors')
    ap.add_argument('--stream-only', action='store_true', help='testing build, no rasterisation: the store pattern alone')
    ap.add_argument('--reps', type=int, default=3)
    ap.add_argument('--no-chunks', action='store_true')
    args = ap.parse_args()
    from torchdrivesim_amd import _native, _ops
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device('cuda', 0)
    L = None
    if args.stream_only:
        L = _native.testing_lib()
        _native._lib = L
        L.tds_raster_set_debug(1 | 2)
    B, A = 1024, 64
    sim, actions, _ = bench.build_simulator(B, A, dev, seed=1234)
    for i in range(5):
        sim.step(actions[i % 8])
    res = Resolution(bench.RES, bench.RES)
    filler = torch.empty(int(args.filler_gb * 2 ** 30), dtype=torch.uint8, device=dev) if args.filler_gb > 0 else None
    if filler is not None:
        filler.fill_(1)
    shape = (B, A, 3, bench.RES, bench.RES)
    bufs = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(args.buffers)]
    free, total = torch.cuda.mem_get_info(dev)
    print(f'filler {args.filler_gb:g} GB, {args.buffers} buffers of {bufs[0].numel() * 4 / 1e9:.2f} GB, device memory free {free / 1e9:.1f} of {total / 1e9:.1f} GB', flush=True)
    subs = [sim.select_batch_elements(torch.arange(lo, min(lo + args.chunk, B)), in_place=False) for lo in range(0, B, args.chunk)]

    def timed(fn):
        ms = []
        for _ in range(args.reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        return min(ms)

    for bi, buf in enumerate(bufs):
        sim.render_egocentric(res=res, fov=bench.FOV, out=buf)            # first touch
        torch.cuda.synchronize()
        _ops.raster_events = []
        for _ in range(args.reps):
            sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
        torch.cuda.synchronize()
        whole = min(a.elapsed_time(b) for a, b in _ops.raster_events)
        _ops.raster_events = None
        fill = timed(lambda: buf.fill_(2.0))
        chunks = []
        for ci, sub in enumerate(subs if not args.no_chunks else []):
            lo = ci * args.chunk
            view = buf[lo:lo + sub.batch_size]
            sub.render_egocentric(res=res, fov=bench.FOV, out=view)
            _ops.raster_events = []
            for _ in range(args.reps):
                sub.render_egocentric(res=res, fov=bench.FOV, out=view)
            torch.cuda.synchronize()
            chunks.append(min(a.elapsed_time(b) for a, b in _ops.raster_events))
            _ops.raster_events = None
        if args.no_chunks:
            continue
        chunks = np.array(chunks)
        gbs = (args.chunk * A * 3 * bench.RES * bench.RES * 4) / (chunks * 1e-3) / 1e9
        print(f'buffer {bi} ptr {buf.data_ptr():x}: whole launch {whole:.2f} ms, fill_ {fill:.2f} ms, sum of chunks {chunks.sum():.2f} ms; '
              f'GB/s per chunk of {args.chunk} scenes: ' + ' '.join(f'{g:.0f}' for g in gbs), flush=True)


if __name__ == '__main__':
    main()
