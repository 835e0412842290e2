#!/usr/bin/env python3
"""
The single-GPU configurations of BASELINE.json other than the headline one (which is bench.py), one JSON line each:
  config 2   B=256 x A=64: KinematicBicycle.step + render_egocentric 256x256 + compute_collision(iou)
  config 3   + compute_offroad + compute_wrong_way (Town01 lane map)
  config 5   config 3's forward with state and action requiring grad, then backward through kinematics, IoU, off-road and the rasteriser
  headline+  B=1024 x A=64 with the wrong-way query added, and the time of every part measured on its own
    python tools/bench_configs.py [--steps 20] [--warmup 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(fn, steps, warmup):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(warmup + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--only', default=None, help='run one configuration (config2, config3, config5, headline+wrong_way)')
    args = ap.parse_args()
    from torchdrivesim_amd import lanelet2
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device('cuda', 0)
    lanes = lanelet2.load_lanelet_map(os.path.join(ROOT, 'tests', 'golden', 'carla_Town01.osm.gz'), origin=(0.0, 0.0))
    res = Resolution(bench.RES, bench.RES)
    sink = {}

    def report(name, B, A, dt, **extra):
        print(json.dumps(dict(config=name, batch=B, agents=A, ms_per_step=1e3 * dt, agent_steps_per_s=B * A / dt, **extra)), flush=True)

    for name, B in (('config2', 256), ('config3', 256), ('config5', 256), ('headline+wrong_way', 1024)):
        A = 64
        if args.only and name != args.only:
            continue
        sim, actions, _ = bench.build_simulator(B, A, dev, seed=1234, lanelet_map=lanes)
        state0 = sim.get_state().clone()

        def fwd(i, name=name, sim=sim, actions=actions):
            sim.step(actions[i % actions.shape[0]])
            sink['img'] = sim.render_egocentric(res=res, fov=bench.FOV)
            sink['col'] = sim.compute_collision()
            if name != 'config2':
                sink['off'] = sim.compute_offroad()
                sink['ww'] = sim.compute_wrong_way()

        def fwd_bwd(i, sim=sim, actions=actions, state0=state0):
            s0 = state0.clone().requires_grad_(True)
            act = actions[i % actions.shape[0]].clone().requires_grad_(True)
            sim.kinematic_model.set_state(s0)          # not Simulator.set_state: its `where(mask, new, current)` keeps the previous graph alive
            sim.step(act)
            img = sim.render_egocentric(res=res, fov=bench.FOV)
            loss = img.mean() + sim.compute_collision().sum() + sim.compute_offroad().sum()
            loss.backward()
            sink['g'] = (s0.grad, act.grad)

        if name == 'config5':
            dt = timed(fwd_bwd, args.steps, args.warmup)
            g_state, g_act = sink['g']
            report(name, B, A, dt, grad_state_abs_mean=float(g_state.abs().mean()), grad_action_abs_mean=float(g_act.abs().mean()),
                   note='forward + backward (loss = mean(image) + sum(collision) + sum(offroad)); state reset every step')
        else:
            dt = timed(fwd, args.steps, args.warmup)
            extra = {}
            if name.startswith('headline'):
                parts = dict(step=lambda i: sim.step(actions[i % 8]), render=lambda i: sink.__setitem__('img', sim.render_egocentric(res=res, fov=bench.FOV)),
                             collision=lambda i: sim.compute_collision(), offroad=lambda i: sim.compute_offroad(), wrong_way=lambda i: sim.compute_wrong_way())
                extra['parts_ms'] = {k: 1e3 * timed(f, args.steps, args.warmup) for k, f in parts.items()}
                extra['wrong_way_nonzero_fraction'] = float((sink['ww'] > 0).float().mean())
            report(name, B, A, dt, **extra)
        del sim
        sink.clear()
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
