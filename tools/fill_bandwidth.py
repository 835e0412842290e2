#!/usr/bin/env python3
"""What the part's HBM takes when a kernel only WRITES: torch's fill_ / zero_ over the 51.5 GB the headline raster launch writes
(B=1024 x A=64 x 3 x 256 x 256 float32).  DESIGN_HISTORY.md section 4 quotes it beside the raster kernel's launch time on the same box.
   python tools/fill_bandwidth.py"""
import torch

x = torch.empty(1024 * 64 * 3 * 256 * 256, dtype=torch.float32, device='cuda')
for fn, name in ((lambda: x.zero_(), 'zero_'), (lambda: x.fill_(1.5), 'fill_')):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    print(f'{name}: min {min(ts):.3f} ms -> {x.numel() * 4 / min(ts) / 1e6:.0f} GB/s over {x.numel() * 4 / 1e9:.1f} GB')
