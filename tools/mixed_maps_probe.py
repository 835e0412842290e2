#!/usr/bin/env python3
"""Where does a mixed-map launch lose time against the single-map one?  Raster launches only, B = 1024 x 64, 256 x 256 float32:
  town01        the headline (one map, MapView in the kernel arguments)
  town02        one map, the other town
  mixed         collated Town01 / Town02, a map set (MapView read per camera through scene_map)
  town01_as_set Town01 only, but through a set of two identical maps: the cost of the set alone
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from torchdrivesim_amd import _ops                                  # noqa: E402
from torchdrivesim_amd.utils import Resolution                      # noqa: E402


def time_render(sim, buf, reps=8):
    res = Resolution(bench.RES, bench.RES)
    for _ in range(2):
        sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
    torch.cuda.synchronize()
    _ops.raster_events = []
    for _ in range(reps):
        sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in _ops.raster_events]
    _ops.raster_events = None
    return float(np.mean(ms)), float(np.min(ms))


def main():
    dev = torch.device('cuda', 0)
    B, A = int(os.environ.get('B', 1024)), 64
    buf = _ops.owned_image((B, A, 3, bench.RES, bench.RES), torch.float32, dev)
    out = {}
    sim, _, _ = bench.build_simulator(B, A, dev, seed=1234)
    out['town01'] = time_render(sim, buf)
    # the same scenes through a set of two maps with identical content
    scene = sim._scene()
    one = scene['maps'][0][0]
    gen = sim.birdview_mesh_generator
    lv = one.levels
    twin = sim.renderer.make_static_map(gen.background_mesh[0:1], lv, device=dev)
    scene['maps'][0] = (_ops.StaticMapSet([one, twin], torch.arange(B, dtype=torch.int32) % 2), None)
    out['town01_as_set'] = time_render(sim, buf)
    del sim, scene
    sim, _, _ = bench.build_simulator(B, A, dev, seed=4321, mixed=True)
    out['mixed'] = time_render(sim, buf)
    odd = sim.select_batch_elements(list(range(1, B, 2)) * 2, in_place=False)
    out['town02'] = time_render(odd, buf)
    for k, v in out.items():
        print(f'{k:16s} mean {v[0]:.3f} ms  min {v[1]:.3f} ms')


if __name__ == '__main__':
    main()
