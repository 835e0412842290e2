#!/usr/bin/env python3
"""The GymEnv.step body with the metrics beside the rasteriser (Simulator.overlap_infractions) or behind it, on the SAME two output buffers,
for a given build of the library: ms per step and the raster launch as HIP events see it.
   python tools/overlap_experiment.py [path of a libtdship.so | -] [--steps 20]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('lib', nargs='?', default='-')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--side-priority', type=int, default=-1)
    args = ap.parse_args()
    from torchdrivesim_amd import _native
    if args.lib != '-':
        _native.LIB_PATH = os.path.abspath(args.lib)
    import bench
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.rendering import allocate_image_ring
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device('cuda', 0)
    from torchdrivesim_amd.simulator import Simulator
    Simulator._side_streams[0] = torch.cuda.Stream(device=dev, priority=args.side_priority)
    B, A = args.batch, 64
    sim, actions, _ = bench.build_simulator(B, A, dev, seed=1234)
    res = Resolution(bench.RES, bench.RES)
    bufs, probe = allocate_image_ring(lambda out: sim.render_egocentric(res=res, fov=bench.FOV, out=out), (B, A, 3, bench.RES, bench.RES), torch.float32, dev)
    print('library', _native.LIB_PATH, '| ring probe', [round(x, 2) for x in probe['launch_ms']], 'kept', probe['kept'], flush=True)
    sink = {}
    state0 = sim.get_state().clone()

    def step(i):
        sim.step(actions[i % actions.shape[0]])
        sink['img'] = sim.render_egocentric(res=res, fov=bench.FOV, out=bufs[i % 2])
        sink['col'] = sim.compute_collision()
        sink['off'] = sim.compute_offroad()

    for rnd in range(3):
        for overlap in (True, False):
            sim.overlap_infractions = overlap
            sim.kinematic_model.set_state(state0.clone())          # every block simulates the same steps (agents that have left the map cost the off-road query more)
            for i in range(3):
                step(i)
            torch.cuda.synchronize()
            _ops.raster_events = []
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(3 + i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps * 1e3
            raster = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events]))
            _ops.raster_events = None
            print(f'round {rnd} overlap {overlap!s:5}: {dt:.3f} ms per step, raster launch {raster:.3f} ms, step - raster {dt - raster:.3f}', flush=True)


if __name__ == '__main__':
    main()
