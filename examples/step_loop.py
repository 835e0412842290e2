#!/usr/bin/env python3
"""
The loop of the reference's examples/gym_env.py:83-126 (GymEnv.step) on this framework: act -> Simulator.step -> bird's-eye observation ->
infractions as reward terms.  Runs on one MI355X with the Town01 package shipped under tests/golden/ (mesh, stop lines, light programmes,
lane map); the "policy" is random.

    python examples/step_loop.py [--batch 64] [--agents 16] [--steps 50] [--res 128]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (scene construction shared with the benchmark)
from torchdrivesim_amd import lanelet2  # noqa: E402
from torchdrivesim_amd.map import load_map_config, traffic_controls_from_map_config  # noqa: E402
from torchdrivesim_amd.traffic_lights import current_light_state_tensor_from_controller  # noqa: E402
from torchdrivesim_amd.utils import Resolution  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--agents', type=int, default=16)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--res', type=int, default=128)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    gold = os.path.join(ROOT, 'tests', 'golden')
    lanes = lanelet2.load_lanelet_map(os.path.join(gold, 'carla_Town01.osm.gz'), origin=(0.0, 0.0))
    cfg = load_map_config(os.path.join(gold, 'maps', 'carla_Town01', 'metadata.json'))
    sim, _, _ = bench.build_simulator(args.batch, args.agents, dev, seed=0, lanelet_map=lanes)
    # traffic lights: the programmes run on the host, the device sees one index per light and step
    controls = {k: v.extend(args.batch).to(dev) for k, v in traffic_controls_from_map_config(cfg).items()}
    sim.traffic_controls = controls
    programme = cfg.traffic_light_controller
    light_ids = [s.actor_id for s in cfg.stoplines if s.agent_type == 'traffic_light']
    res = Resolution(args.res, args.res)
    g = torch.Generator(device=dev).manual_seed(0)
    totals = {k: torch.zeros((), device=dev) for k in ('collision', 'offroad', 'wrong_way', 'red_light')}     # summed on the device: no sync per step
    warmup = 3                                   # the first steps build the device maps, lane tables and workspaces (once per simulator): not timed
    for it in range(warmup + args.steps):
        if it == warmup:
            for v in totals.values():
                v.zero_()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        action = torch.rand((args.batch, args.agents, 2), device=dev, generator=g) * 2 - 1
        programme.tick(0.1)
        controls['traffic_light'].set_state(current_light_state_tensor_from_controller(programme, light_ids).unsqueeze(0).expand(args.batch, -1).to(dev))
        sim.step(action)
        obs = sim.render_egocentric(res=res, fov=35.0)                       # (B, A, 3, H, W): what a policy would consume
        totals['collision'] += (sim.compute_collision() > 0).float().mean()
        totals['offroad'] += (sim.compute_offroad() > 0).float().mean()
        totals['wrong_way'] += (sim.compute_wrong_way() > 0).float().mean()
        totals['red_light'] += sim.compute_traffic_lights_violations().float().mean()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{args.steps} steps of {args.batch} x {args.agents} agents, observations {tuple(obs.shape)}: {1e3 * dt / args.steps:.2f} ms per step, '
          f'{args.batch * args.agents * args.steps / dt / 1e6:.2f} M agent-steps/s')
    print('fraction of agents per step: ' + ', '.join(f'{k} {float(v) / args.steps:.3f}' for k, v in totals.items()))


if __name__ == '__main__':
    main()
