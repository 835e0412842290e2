#!/usr/bin/env python3
"""Launch time of the headline raster kernel into each of several 51.5 GB output tensors of one process, and torch's fill_ over the same tensors
(DESIGN_HISTORY.md section 4: an allocation is persistently fast, 7.2 ms, or slow, 8.4 ms, for this kernel; fill_ takes 7.45 ms on both).
   python tools/output_buffer_speed.py [library.so under tools/scratch | -]"""
import sys, os, torch, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from torchdrivesim_amd import _native
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != '-' else None
if lib: _native.LIB_PATH = os.path.join(R, 'tools', 'scratch', lib)
import bench
from torchdrivesim_amd import _ops
from torchdrivesim_amd.utils import Resolution
dev = torch.device('cuda', 0)
sim, actions, _ = bench.build_simulator(1024, 64, dev, seed=1234)
for i in range(5): sim.step(actions[i % 8])
res = Resolution(256, 256)
shape = (1024, 64, 3, 256, 256)
bufs = [torch.empty(shape, dtype=torch.float32, device=dev) for _ in range(5)]
pads = []
cur = {'i': 0}
real_empty = torch.empty
def fake_empty(*a, **k):
    sh = a[0] if a and isinstance(a[0], (tuple, list, torch.Size)) else a
    if tuple(sh) == shape and k.get('dtype', torch.float32) == torch.float32:
        return bufs[cur['i']]
    return real_empty(*a, **k)
_ops.torch.empty = fake_empty
try:
    for rnd in range(3):
        for bi in range(len(bufs)):
            cur['i'] = bi
            _ops.raster_events = []
            for _ in range(4): sim.render_egocentric(res=res, fov=35.0)
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in _ops.raster_events]
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            bufs[bi].fill_(1.0); a.record(); bufs[bi].fill_(2.0); b.record(); torch.cuda.synchronize()
            fill_ms = a.elapsed_time(b)
            print(lib, 'fill_ %.2f ms |' % fill_ms, 'round', rnd, 'buffer', bi, 'ptr %x' % bufs[bi].data_ptr(), 'offset in 2 MiB units mod 8: %d' % ((bufs[bi].data_ptr() >> 21) & 7), ['%.2f' % m for m in ms], flush=True)
finally:
    _ops.torch.empty = real_empty
