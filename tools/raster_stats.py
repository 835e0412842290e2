#!/usr/bin/env python3
"""Work counters of the K3 bit-plane kernel on the bench workload (debug flag 128): per wave and image."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from torchdrivesim_amd import _native
from torchdrivesim_amd.utils import Resolution
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
RES = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device('cuda', 0)
L = _native.testing_lib()           # the work counters exist only in the testing build
_native._lib = L
sim, actions, _ = bench.build_simulator(B, 64, dev, seed=1234)
for i in range(5):
    sim.step(actions[i % 8])
buf = (ctypes.c_ulonglong * 16)()
sim.render_egocentric(res=Resolution(RES, RES), fov=35.0); torch.cuda.synchronize()
L.tds_raster_get_stats(buf)
SPLIT = len(sys.argv) > 3 and sys.argv[3] == 'split'   # 'split': whatever form the library chooses at this resolution (K3s + K3r below 144 / 208 pixels)
L.tds_raster_set_debug(128 if SPLIT else 128 | 8192)          # default: the fused kernel at every resolution
sim.render_egocentric(res=Resolution(RES, RES), fov=35.0); torch.cuda.synchronize()
L.tds_raster_set_debug(0)
L.tds_raster_get_stats(buf)
names = ['batches', 'faces', 'fill chunks', 'fill windows', 'fill rows', 'edges', 'edge rounds', 'V chunks', 'V windows', 'V rows', 'H chunks',
         'H windows', 'H rows', 'V edges', 'clipped edges', 'faces with rows']
nimg = B * 64
for n, v in zip(names, buf):
    print(f'{n:14s} {v / nimg:10.1f} per image {v / nimg / 4:10.2f} per wave')
