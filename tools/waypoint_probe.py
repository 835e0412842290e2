#!/usr/bin/env python3
"""render_egocentric with waypoint goals at the bench shape (B x 64 cameras, 256 x 256): every camera draws the discs of its own goal
(per-camera triangles: `extra_tri` of tds_raster_scene, ten triangles per waypoint, one more key) -- ms per render call against the scene without goals,
float32 and uint8, and the host-side cost of building the triangles."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from torchdrivesim_amd import _ops                                  # noqa: E402
from torchdrivesim_amd.goals import WaypointGoal                    # noqa: E402
from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config   # noqa: E402
from torchdrivesim_amd.utils import Resolution                      # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    B, A = int(os.environ.get('B', 1024)), 64
    res = Resolution(bench.RES, bench.RES)
    for dtype in ('float32', 'uint8'):
        sim, actions, _ = bench.build_simulator(B, A, dev, seed=1234)
        sim.renderer = renderer_from_config(HipRendererConfig(out_dtype=dtype), res=res, fov=bench.FOV)
        sim._scene_cache = None
        for i in range(3):
            sim.step(actions[i])
        s = sim.get_state()
        g = torch.Generator(device=dev).manual_seed(1)
        for M in (0, 1, 3):
            if M:
                wp = s[..., None, None, :2] + (torch.rand(B, A, 2, M, 2, device=dev, generator=g) - 0.5) * 30.0        # N = 2 collections of M waypoints near the agent
                sim.waypoint_goals = WaypointGoal(wp)
            else:
                sim.waypoint_goals = None
            for _ in range(2):
                img = sim.render_egocentric(res=res, fov=bench.FOV)
            torch.cuda.synchronize()
            _ops.raster_events = []
            t0 = time.perf_counter()
            for _ in range(8):
                img = sim.render_egocentric(res=res, fov=bench.FOV)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 8 * 1e3
            ms = float(np.median([a.elapsed_time(b) for a, b in _ops.raster_events]))
            _ops.raster_events = None
            print(f'B={B} {dtype}: {M} waypoints per camera ({10 * M} triangles): raster launch {ms:.3f} ms, render call {wall:.3f} ms wall', flush=True)
        del sim, img
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
