#!/usr/bin/env python3
"""
Headline benchmark (BASELINE.json): agent-steps/s of the GymEnv.step body (examples/gym_env.py:83-126 of the reference)
    Simulator.step(action) -> render_egocentric(256x256, fov 35 m) -> compute_collision() -> compute_offroad()
at B = 1024 scenes x A = 64 agents per GPU on the Town01 map (tests/golden/town01_mesh.npz), synthetic agents.

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ...`: one process per GPU, the scene batch is
sharded (B per GPU is fixed: weak scaling), there is no data-path collective; RCCL is only used for the barrier and
the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RES, FOV = 256, 35.0
ALGO_BYTES_PER_IMAGE = 3 * RES * RES * 4          # fp32 CHW raster output, the algorithmic bytes of K3
HBM_PEAK_GBS = 8000.0                             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0                             # what a float4 copy reaches on this part (same guide): the practical roof, reported beside the spec


def load_town01():
    t = np.load(os.path.join(ROOT, 'tests', 'golden', 'town01_mesh.npz'))
    return t['verts'], t['faces'], t['vert_category'], [str(c) for c in t['categories']]


def synth_agents(B, A, road_verts, seed):
    """positions = random road vertices + N(0,1 m); psi ~ U(-pi,pi); v ~ U(0,10); size (4.5,2.0)*U(0.9,1.1);
    lr = 1.5; present ~ Bernoulli(0.9), agent 0 always present."""
    g = np.random.default_rng(seed)
    xy = road_verts[g.integers(0, len(road_verts), (B, A))] + g.normal(0.0, 1.0, (B, A, 2))
    state = np.concatenate([xy, g.uniform(-np.pi, np.pi, (B, A, 1)), g.uniform(0, 10, (B, A, 1))], -1).astype(np.float32)
    size = (np.array([4.5, 2.0]) * g.uniform(0.9, 1.1, (B, A, 2))).astype(np.float32)
    present = g.uniform(size=(B, A)) < 0.9
    present[:, 0] = True
    actions = g.uniform(-1, 1, (8, B, A, 2)).astype(np.float32)
    return state, size, present, actions


def build_simulator(B, A, device, seed, metric='iou', lanelet_map=None):
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig, CollisionMetric
    from torchdrivesim_amd.utils import Resolution
    verts, faces, vcat, cats = load_town01()
    road = BirdviewMesh(verts=torch.from_numpy(verts)[None], faces=torch.from_numpy(faces.astype(np.int64))[None], categories=cats,
                        colors={}, zs={}, vert_category=torch.from_numpy(vcat.astype(np.int64))[None]).expand(B).to(device)
    state, size, present, actions = synth_agents(B, A, verts[vcat == cats.index('road')], seed)
    km = KinematicBicycle()
    km.set_params(lr=torch.full((B, A), 1.5, device=device))
    km.set_state(torch.from_numpy(state).to(device))
    cfg = TorchDriveConfig(collision_metric=CollisionMetric(metric), renderer=HipRendererConfig())
    renderer = renderer_from_config(cfg.renderer, res=Resolution(RES, RES), fov=FOV)
    sim = Simulator(road, km, torch.from_numpy(size).to(device), torch.from_numpy(present).to(device), cfg, renderer=renderer,
                    lanelet_map=None if lanelet_map is None else [lanelet_map] * B)
    return sim, torch.from_numpy(actions).to(device), (state, size, present, actions, verts, faces, vcat, cats)


def cpu_baseline(host, n_scenes, A):
    """The oracle (a C port of the reference's CPU path, oracle/tds_oracle.c) timed on this box's host cores on a bounded
    sample of the same workload: the first `n_scenes` scenes, one step."""
    from oracle import oracle as orc
    orc.build()
    state, size, present, actions, verts, faces, vcat, cats = host
    sv, sa, sf = orc.static_mesh_arrays(verts, faces, vcat, cats)

    def one_step(lo, hi):
        st, sz, pr, act = state[lo:hi], size[lo:hi], present[lo:hi], actions[0, lo:hi]
        n = hi - lo
        s1 = orc.bicycle_step(st, act, np.full((n, A), 1.5, np.float32))
        sc = np.stack([np.sin(s1[..., 2]), np.cos(s1[..., 2])], -1).astype(np.float32)
        mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (n, A, A)))
        img = orc.render_scenes(s1, sz, mask, s1[..., :2].copy(), sc, sv, sa, sf, FOV, RES, agent_sc=sc)
        boxes = np.concatenate([s1[..., :2], sz, s1[..., 2:3]], -1)
        col = orc.collision(boxes, pr, metric='iou', sc=sc)
        off = orc.offroad(s1, sz, verts, faces, 0.5, present=pr, sc=sc)
        return img, col, off

    cores = os.cpu_count() or 1
    out = {}
    for label, threads, count, chunk in (('all', cores, n_scenes, 64), ('one', 1, max(1, min(16, n_scenes // 16)), 4)):
        orc.set_num_threads(threads)
        dt = 0.0
        for lo in range(0, count, chunk):                  # in chunks: the images of a chunk of 64 scenes are 3.2 GB of host memory
            t0 = time.perf_counter()
            one_step(lo, min(lo + chunk, count))
            dt += time.perf_counter() - t0
        out[label] = (count * A / dt, dt, count)
    orc.set_num_threads(cores)
    return dict(value=out['all'][0], unit='agent-steps/s', cores=cores, kind='port',
                sample=f"{out['all'][2]} scenes x {A} agents, 1 step of the same workload (oracle/tds_oracle.c, OpenMP over images) in "
                       f"{out['all'][1]:.2f} s; single thread: {out['one'][0]:.1f} agent-steps/s on {out['one'][2]} scenes ({out['one'][1]:.2f} s)",
                value_single_thread=out['one'][0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='scenes per GPU')
    ap.add_argument('--agents', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-scenes', type=int, default=320, help='scenes of the CPU-baseline sample (all cores, about 10 s); the single-thread run uses a sixteenth')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('--gpus N > 1 must be launched with python -m torch.distributed.run --nproc-per-node N')
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=device)

    B, A = args.batch, args.agents
    sim, actions, host = build_simulator(B, A, device, seed=1234 + rank)
    from torchdrivesim_amd.utils import Resolution
    res = Resolution(RES, RES)

    from torchdrivesim_amd import _ops
    sink = {}

    def step(i):
        sim.step(actions[i % actions.shape[0]])
        sink['img'] = sim.render_egocentric(res=res, fov=FOV)
        sink['col'] = sim.compute_collision()
        sink['off'] = sim.compute_offroad()

    from torchdrivesim_amd import parallel

    def barrier():
        parallel.barrier(device)            # dist.barrier() when there is a process group, then torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    _ops.raster_events = []             # HIP events around every raster launch of the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device)        # slowest rank
    raster_ms = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events])) if _ops.raster_events else float('nan')
    _ops.raster_events = None

    if rank == 0:
        value = world * B * A * args.steps / elapsed
        achieved = (B * A * ALGO_BYTES_PER_IMAGE) / (raster_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'raster_traffic.json')
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get('batch') == B and tj.get('agents') == A and tj.get('res') == RES:
                traffic = tj.get('hbm_bytes_per_launch')
        line = dict(
            metric='agent-steps/sec (whole node) at B=1024xA=64, 256x256 BEV', value=value, unit='agent-steps/s', n_gpus=world,
            steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / max(args.steps, 1), higher_is_better=True, scaling='weak',
            vs_baseline=None, dtype='f32', data='synthetic',
            config=dict(workload=f'B={B}xA={A} per GPU, carla_Town01 mesh (30750 faces): KinematicBicycle.step + render_egocentric {RES}x{RES} '
                                 f'fp32 fov {FOV:g} m + compute_collision(iou) + compute_offroad', global_batch=world * B, agents=A, res=RES,
                        parallelism=f'scene-batch sharding x{world}, no collectives'),
            roofline=dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS, traffic=traffic,
                          kernel='raster_scene_bits_kernel', avg_launch_ms=raster_ms, algorithmic_bytes_per_launch=B * A * ALGO_BYTES_PER_IMAGE,
                          measured_copy_peak=HBM_COPY_GBS, frac_of_measured_copy=achieved / HBM_COPY_GBS))
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(host, min(args.cpu_scenes, B), A)
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
