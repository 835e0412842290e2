#!/usr/bin/env python3
"""
Headline benchmark (BASELINE.json): agent-steps/s of the GymEnv.step body (examples/gym_env.py:83-126 of the reference)
    Simulator.step(action) -> render_egocentric(256x256, fov 35 m) -> compute_collision() -> compute_offroad()
at B = 1024 scenes x A = 64 agents per GPU on the Town01 map (tests/golden/town01_mesh.npz), synthetic agents.

    python bench.py --gpus N --steps K --warmup W
One process per GPU; the scene batch is sharded (B per GPU is fixed: weak scaling) and there is NO data-path collective.
N > 1 runs either way:
  * `python bench.py --gpus N ...` by itself: this process never touches a GPU; it starts N children (one per GPU, each with
    HIP_VISIBLE_DEVICES narrowed to its own device), which rendezvous on 127.0.0.1 over gloo -- used only for the barrier
    around the timed region and the max-over-ranks of the elapsed time;
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`: the same worker, barrier over RCCL.
Rank 0 prints ONE JSON line.  `--dry-run` runs the whole launch / barrier / reduction / reporting path without a GPU and
without kernels (CPU test of the launcher, tests/test_bench_launcher.py).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RES, FOV = 256, 35.0
ALGO_BYTES_PER_IMAGE = 3 * RES * RES * 4          # fp32 CHW raster output, the algorithmic bytes of K3
HBM_PEAK_GBS = 8000.0                             # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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


def load_town02():
    t = np.load(os.path.join(ROOT, 'tests', 'golden', 'town02_mesh.npz'))
    return t['verts'], t['faces'], t['vert_category'], [str(c) for c in t['categories']]


def build_simulator(B, A, device, seed, metric='iou', lanelet_map=None, mixed=False):
    """`mixed`: the batch is collated from Town01 and Town02 (mesh.py:232-245 of the reference: padded to the larger mesh), even scenes on
    Town01, odd scenes on Town02, every scene's agents on its own town's roads.  `mixed='town02'`: the same collated rows, every scene on Town02
    (one distinct mesh: a single map)."""
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig, CollisionMetric
    from torchdrivesim_amd.utils import Resolution
    verts, faces, vcat, cats = load_town01()
    town = lambda v, f, vc, c: BirdviewMesh(verts=torch.from_numpy(v)[None], faces=torch.from_numpy(f.astype(np.int64))[None], categories=c,
                                            colors={}, zs={}, vert_category=torch.from_numpy(vc.astype(np.int64))[None])
    if not mixed:
        road = town(verts, faces, vcat, cats).expand(B).to(device)
        state, size, present, actions = synth_agents(B, A, verts[vcat == cats.index('road')], seed)
    else:
        v2, f2, vc2, c2 = load_town02()
        which = np.arange(B) % 2 if mixed is True else np.full(B, int(mixed == 'town02'))
        road = BirdviewMesh.collate([town(verts, faces, vcat, cats), town(v2, f2, vc2, c2)]).to(device)[which.tolist()]       # B padded rows, a real (not expanded) batch
        state, size, present, actions = synth_agents(B, A, verts[vcat == cats.index('road')], seed)
        s2 = synth_agents(B, A, v2[vc2 == c2.index('road')], seed + 1)[0]
        state[which == 1] = s2[which == 1]
    km = KinematicBicycle()
    km.set_params(lr=torch.full((B, A), 1.5, device=device))
    km.set_state(torch.from_numpy(state).to(device))
    cfg = TorchDriveConfig(collision_metric=CollisionMetric(metric), renderer=HipRendererConfig())
    renderer = renderer_from_config(cfg.renderer, res=Resolution(RES, RES), fov=FOV)
    sim = Simulator(road, km, torch.from_numpy(size).to(device), torch.from_numpy(present).to(device), cfg, renderer=renderer,
                    lanelet_map=None if lanelet_map is None else [lanelet_map] * B)
    return sim, torch.from_numpy(actions).to(device), (state, size, present, actions, verts, faces, vcat, cats)


def cpu_quota():
    """CPU quota of this process's cgroup in CPUs (None: unlimited) -- reported beside the thread count of the CPU baseline"""
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else int(q) / int(p)
    except (OSError, ValueError):
        try:
            q, p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()), int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            return q / p if q > 0 and p > 0 else None
        except (OSError, ValueError):
            return None


def cpu_baseline(host, n_scenes, A):
    """The oracle (a C port of the reference's CPU path, oracle/tds_oracle.c) timed on this box's host cores on a bounded
    sample of the same workload: the first `n_scenes` scenes, one step."""
    from oracle import oracle as orc
    orc.build()
    state, size, present, actions, verts, faces, vcat, cats = host
    sv, sa, sf = orc.static_mesh_arrays(verts, faces, vcat, cats)

    buffers = {}                                 # image buffers by chunk size: allocated (and first touched) once, not per chunk

    def one_step(lo, hi):
        st, sz, pr, act = state[lo:hi], size[lo:hi], present[lo:hi], actions[0, lo:hi]
        n = hi - lo
        if n not in buffers:
            buffers[n] = np.empty((n, A, 3, RES, RES), np.float32)
            buffers[n][:] = 0.0
        s1 = orc.bicycle_step(st, act, np.full((n, A), 1.5, np.float32))
        sc = np.stack([np.sin(s1[..., 2]), np.cos(s1[..., 2])], -1).astype(np.float32)
        mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (n, A, A)))
        img = orc.render_scenes(s1, sz, mask, s1[..., :2].copy(), sc, sv, sa, sf, FOV, RES, agent_sc=sc, out=buffers[n])
        boxes = np.concatenate([s1[..., :2], sz, s1[..., 2:3]], -1)
        col = orc.collision(boxes, pr, metric='iou', sc=sc)
        off = orc.offroad(s1, sz, verts, faces, 0.5, present=pr, sc=sc)
        return img, col, off

    cores, quota = orc.usable_cpus(), cpu_quota()         # the GPU boxes of this pool show 256 logical CPUs and grant 16 CPUs' worth of time
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
                sample=f"{out['all'][2]} scenes x {A} agents, 1 step of the same workload (oracle/tds_oracle.c, OpenMP over images, {cores} threads: "
                       f"{os.cpu_count()} logical CPUs, CPU quota of the cgroup {'none' if quota is None else f'{quota:g} CPUs'}) in "
                       f"{out['all'][1]:.2f} s; single thread: {out['one'][0]:.1f} agent-steps/s on {out['one'][2]} scenes ({out['one'][1]:.2f} s)",
                value_single_thread=out['one'][0])


def kernel_source_stamp():
    """sha256 over everything the measured FETCH_SIZE / WRITE_SIZE of the raster kernel depend on: the kernel's sources, the layout of the
    grid entries it reads (map.hip), the build flags (Makefile) and the host function that chooses the grid cell size
    (rendering/hip.py: HipRenderer.make_static_map).
    PMC figures committed under profiles/ carry the stamp of the build they were measured on and are refused for any other (VERDICT r1:
    the static traffic figure must not go stale silently)."""
    h = hashlib.sha256()
    pkg = os.path.join(ROOT, 'torchdrivesim_amd')
    for path in (os.path.join(pkg, 'csrc', 'raster.hip'), os.path.join(pkg, 'csrc', 'tds_common.h'), os.path.join(pkg, 'csrc', 'map.hip'),
                 os.path.join(pkg, 'csrc', 'Makefile')):
        with open(path, 'rb') as f:
            h.update(f.read())
    # of the host side, the function that chooses the cell size of the rendering grid (the rest of rendering/hip.py does not touch the kernel's reads)
    import inspect
    from torchdrivesim_amd.rendering.hip import HipRenderer
    h.update(inspect.getsource(HipRenderer.make_static_map).encode())
    return h.hexdigest()[:16]


def stamped_traffic(B, A, mode='f32', res=None):
    """(HBM bytes per launch from the PMC passes committed in profiles/raster_traffic.json, note) -- null unless it was measured on
    exactly this kernel source and this workload."""
    tpath = os.path.join(ROOT, 'profiles', 'raster_traffic.json')
    if not os.path.exists(tpath):
        return None, 'no PMC figure committed'
    tj = json.load(open(tpath))
    ent = tj.get(mode) if isinstance(tj.get(mode), dict) else (tj if mode == 'f32' else None)
    if not ent:
        return None, f'no PMC figure for mode {mode}'
    if ent.get('kernel_source_sha') != kernel_source_stamp():
        return None, f"PMC figure refused: measured on kernel source {ent.get('kernel_source_sha')}, this build is {kernel_source_stamp()}"
    if (ent.get('batch'), ent.get('agents'), ent.get('res')) != (B, A, res if res is not None else RES):
        return None, 'PMC figure is for another workload'
    return ent.get('hbm_bytes_per_launch'), f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {ent.get('source')}"


def stamped_limiter(B, A, mode, res=None):
    """What bounds a mode that is NOT bound by HBM, from the PMC passes committed in profiles/raster_traffic.json (same stamp rule as the traffic):
    dict(bound='valu', valu_busy_simds_per_se_of_32 = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES (raw; a saturating add chain reads 47.5),
    valu_issue_fraction_of_add_chain = that / 47.5, cycles_per_valu_instruction = 4 / the fraction, valu_lane_occupancy, ...) -- or
    dict(bound='valu', note=why there is no figure)."""
    tpath = os.path.join(ROOT, 'profiles', 'raster_traffic.json')
    ent = json.load(open(tpath)).get(mode) if os.path.exists(tpath) else None
    if not isinstance(ent, dict) or not ent.get('valu'):
        return dict(bound='valu', note=f'no PMC figure committed for mode {mode}')
    if ent.get('kernel_source_sha') != kernel_source_stamp():
        return dict(bound='valu', note=f"PMC figure refused: measured on kernel source {ent.get('kernel_source_sha')}, this build is {kernel_source_stamp()}")
    if (ent.get('batch'), ent.get('agents'), ent.get('res')) != (B, A, res if res is not None else RES):
        return dict(bound='valu', note='PMC figure is for another workload')
    return dict(bound='valu', **ent['valu'], source=f"rocprofv3 --pmc passes, {ent.get('source')}")


class _ImageProbe(torch.autograd.Function):
    """The image term of config 5's loss: <image, w> for a fixed dense random field w -- what the first layer of a policy network would
    hand back.  Forward: dot products over slices of the batch (both tensors read once); backward: w itself goes to the rasteriser's backward (the
    probe is the last term of the loss, its upstream gradient is 1), so that no 12.9 GB temporary is produced by the LOSS."""

    events = []
    checked = False          # the shortcut of backward() has been checked against the upstream gradient (once, in the warm-up)

    @staticmethod
    def forward(ctx, img, w):
        ctx.save_for_backward(w)
        # dot products over slices of the batch (each below 2^31 elements), summed on the device
        a, b = img.flatten(), w.flatten()
        step = 1 << 30
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        out = torch.stack([torch.dot(a[i:i + step], b[i:i + step]) for i in range(0, a.numel(), step)]).sum()
        ev[1].record()
        _ImageProbe.events.append(ev)
        return out

    @staticmethod
    def backward(ctx, g):
        if not _ImageProbe.checked:
            # handing w itself back is right only while the probe enters the loss with coefficient exactly 1: checked, not assumed
            # (one host synchronisation, in the first warm-up step -- never inside a timed region)
            if float(g) != 1.0:
                raise RuntimeError(f'_ImageProbe: the upstream gradient is {float(g)}, not 1 -- the loss scales the image term; return g * w instead')
            _ImageProbe.checked = True
        return ctx.saved_tensors[0], None


def other_configs(device, steps, warmup, only=None, overlap='reserved', overlap_diff='off'):
    """BASELINE.json's other single-GPU configurations (2, 3, 5) and the uint8 output mode of the headline kernel, measured in
    this run after the timed region: ms/step, agent-steps/s, the dominant kernel and its fraction of the HBM roof."""
    from torchdrivesim_amd import _ops, lanelet2
    from torchdrivesim_amd.utils import Resolution
    res = Resolution(RES, RES)
    lanes = lanelet2.load_lanelet_map(os.path.join(ROOT, 'tests', 'golden', 'carla_Town01.osm.gz'), origin=(0.0, 0.0))
    out, sink = [], {}

    def timed(fn):
        for i in range(warmup):
            fn(i)
        torch.cuda.synchronize(device)
        _ops.raster_events, _ops.raster_bwd_events = [], []
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        fwd = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events])) if _ops.raster_events else None
        bwd = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_bwd_events])) if _ops.raster_bwd_events else None
        _ops.raster_events = _ops.raster_bwd_events = None
        return dt, fwd, bwd

    B, A = 256, 64
    for name in ('config2', 'config3', 'config5', 'config5_sum'):
        if only is not None and name != only:
            continue
        sim, actions, _ = build_simulator(B, A, device, seed=1234, lanelet_map=lanes)
        state0 = sim.get_state().clone()

        ring = None
        # as the headline: the loop's stream is kept off four CUs per XCD, the metrics run beside the raster launch on those.  The differentiable
        # configuration can fork too since round 6 (its metric nodes are autograd nodes of the side stream, so their backward runs beside the
        # rasteriser's; same gradients bit for bit: tests/test_gpu_simulator.py) but gains nothing from it at this size -- measured on one box, ms per step:
        # off 4.070 / 4.074, 'stream' 4.090 / 4.076, 'reserved' 4.115 (its backward kernel is not bound by the write stream and pays for the CUs
        # 'reserved' keeps it off; profiles/r06_config5_overlap.log) -- so `overlap_diff` defaults to off
        mode = overlap_diff if name.startswith('config5') else overlap
        if mode == 'reserved':
            sim.overlap_infractions = 'reserved'
            torch.cuda.synchronize(device)
            torch.cuda.set_stream(sim.raster_stream())
        elif mode == 'stream' and name.startswith('config5'):
            sim.overlap_infractions = True
        if not name.startswith('config5'):
            # the forward-only configurations render into a two-buffer ring like the headline (the differentiable one cannot: autograd owns its image)
            from torchdrivesim_amd.rendering import allocate_image_ring
            ring, ring_rep = allocate_image_ring(lambda out: sim.render_egocentric(res=res, fov=FOV, out=out), (B, A, 3, RES, RES), torch.float32, device, count=2)

        def fwd(i):
            sim.step(actions[i % actions.shape[0]])
            sink['img'] = sim.render_egocentric(res=res, fov=FOV, out=ring[i % 2])
            sink['col'] = sim.compute_collision()
            if name != 'config2':
                sink['off'] = sim.compute_offroad()
                sink['ww'] = sim.compute_wrong_way()

        def fwd_bwd(i):
            s0 = state0.clone().requires_grad_(True)
            act = actions[i % actions.shape[0]].clone().requires_grad_(True)
            sim.kinematic_model.set_state(s0)
            sim.step(act)
            img = sim.render_egocentric(res=res, fov=FOV)
            if name == 'config5':
                image_term = _ImageProbe.apply(img, sink['w'])
            else:
                # the benchmark's own reduction over 12.9 GB, timed apart like config 5's probe (HIP events on the loop's stream)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                image_term = img.sum() * (1.0 / 255.0)
                ev[1].record()
                _ImageProbe.events.append(ev)
            loss = image_term + sim.compute_collision().sum() + sim.compute_offroad().sum()
            loss.backward()
            sink['g'] = (s0.grad, act.grad)

        if name == 'config5':
            # the incoming image gradient is a fixed random field (what a policy network's first layer would hand back), not the
            # gradient of a mean: the loss costs one fused multiply-reduce over the image instead of round 1's mean + its materialised gradient
            sink['w'] = torch.rand(B, A, 3, RES, RES, device=device)
        dt, k_fwd, k_bwd = timed(fwd_bwd if name.startswith('config5') else fwd)
        torch.cuda.synchronize(device)
        torch.cuda.set_stream(torch.cuda.default_stream(device))
        what = {'config2': 'step + render_egocentric 256x256 + compute_collision(iou)',
                'config3': 'config2 + compute_offroad + compute_wrong_way (Town01 lane map)',
                'config5': 'step + render + collision + offroad, then backward through kinematics, IoU, off-road and the rasteriser; '
                           'loss = <image, fixed random weights> + sum(collision) + sum(offroad): a DENSE incoming image gradient',
                'config5_sum': 'config5 with the image term sum(image) / 255 (round 1 used mean(image)): the incoming image gradient is one '
                               'constant broadcast over all cameras, which the backward reads as such'}[name]
        ent = dict(config=name, what=what, batch=B, agents=A, ms_per_step=1e3 * dt, agent_steps_per_s=B * A / dt,
                   dominant_kernel='raster_scene_bits_kernel', dominant_kernel_ms=k_fwd,
                   dominant_kernel_frac_of_hbm_peak=None if not k_fwd else B * A * ALGO_BYTES_PER_IMAGE / (k_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS)
        if ring is not None:
            ent['ring_probe'] = dict(launch_ms=ring_rep['launch_ms'], fast=ring_rep['fast'], kept=ring_rep['kept'], fill_ms=ring_rep['fill_ms'], yardstick=ring_rep['yardstick'])
            ent['overlap'] = mode
            # what torch's fill_ takes over the very buffers of this loop: at B = 256 the raster launch beats it (a 12.9 GB write burst is short for
            # this memory system: profiles/r06_tail_attempts.log), so the launch's fraction of the 8 TB/s is read beside this figure
            ent['same_buffer_fill_ms'] = ring_rep['fill_ms']
            ent['raster_launch_over_fill'] = None if not k_fwd else k_fwd / ring_rep['fill_ms']
        else:
            # autograd owns the image of the differentiable step: a fresh tensor per step from the image pool (spread-out physical pages)
            ent['image_allocation'] = 'per step, torch memory pool over tds_torch_alloc (csrc/alloc.hip)'
        if k_bwd is not None:
            ent['raster_backward_kernel_ms'] = k_bwd
        if name.startswith('config5'):
            ent['overlap'] = mode
        if name.startswith('config5') and _ImageProbe.events:
            # the image term of the LOSS (config5: 25.8 GB read by torch.dot; config5_sum: 12.9 GB read by sum()) is the benchmark's, not the
            # library's: reported so that it can be told apart
            ent['loss_probe_ms'] = float(np.mean([a.elapsed_time(b) for a, b in _ImageProbe.events[-steps:]]))
            _ImageProbe.events.clear()
            # the entry LEADS with the library's figure (a reader of the first number saw the probe); the probe-inclusive
            # wall time stays beside it
            total = ent['ms_per_step']
            lib_ms = total - ent['loss_probe_ms']
            ent = dict(config=name, what=ent['what'], batch=B, agents=A, ms_per_step=lib_ms, agent_steps_per_s=B * A / (lib_ms * 1e-3),
                       ms_per_step_is='ms_per_step_library: the wall time per step minus the benchmark\'s own loss probe (loss_probe_ms, HIP events)',
                       ms_per_step_library=lib_ms, ms_per_step_with_loss_probe=total, agent_steps_per_s_with_loss_probe=B * A / (total * 1e-3),
                       **{k: v for k, v in ent.items() if k not in ('config', 'what', 'batch', 'agents', 'ms_per_step', 'agent_steps_per_s')})
            ent['ms_per_step_without_loss_probe'] = lib_ms          # (the name of rounds 3 and 4)
        out.append(ent)
        del sim, ring
        sink.clear()
        _ops.release_image_pool()         # the blocks the image pool has cached go back to the driver (torch.cuda.empty_cache() leaves a MemPool alone)
        torch.cuda.empty_cache()
    return out


def same_run_write_roofs(sim, buf, res, device, reps=5):
    """What this part, in this run, writes into the very output buffer of the timed region (the reference points the roofline entry carries
    beside the 8 TB/s of the data sheet):
      measured_fill_gbs    torch's fill_ over the buffer (a front-to-back stream of the same bytes);
      measured_stream_gbs  the raster kernel itself with nothing to rasterise -- the same launch geometry and store pattern, the cameras moved
                           off the map so that no face and no agent is in view: the bare write stream of this kernel.  NOT an upper bound of
                           the launch with work: stores that arrive paced by the rasterisation are served faster than stores issued
                           back to back (7.07 against 7.29 ms on the same buffer), so both figures are reported as rates, not as roofs."""
    from torchdrivesim_amd import _ops
    n = buf.numel() * buf.element_size()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    buf.fill_(0.0)
    for a, b in ev:
        a.record()
        buf.fill_(0.0)
        b.record()
    torch.cuda.synchronize(device)
    fill_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    state = sim.get_state()
    far = state[..., :2] * 0.0 + 1.0e6
    psi = state[..., 2:3]
    sim.render(far, psi, res=res, fov=FOV, out=buf)
    _ops.raster_events = []
    for _ in range(reps):
        sim.render(far, psi, res=res, fov=FOV, out=buf)
    torch.cuda.synchronize(device)
    stream_ms = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events]))
    _ops.raster_events = None
    assert not bool(buf[:2].any()), 'the cameras of the stream-only launch see something'
    return dict(measured_fill_gbs=n / (fill_ms * 1e-3) / 1e9, measured_stream_gbs=n / (stream_ms * 1e-3) / 1e9, measured_stream_ms=stream_ms)


def u8_mode(device, steps, warmup, B, A):
    """The uint8 output mode of the headline render (a quarter of the bytes, same pixels): a SEPARATE roofline entry, never `value`."""
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = build_simulator(B, A, device, seed=1234)
    sim.renderer = renderer_from_config(HipRendererConfig(out_dtype='uint8'), res=Resolution(RES, RES), fov=FOV)
    sim._scene_cache = None
    sim.step(actions[0])
    for _ in range(warmup):
        img = sim.render_egocentric(res=Resolution(RES, RES), fov=FOV)
    torch.cuda.synchronize(device)
    _ops.raster_events = []
    for _ in range(steps):
        img = sim.render_egocentric(res=Resolution(RES, RES), fov=FOV)
    torch.cuda.synchronize(device)
    ms = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events]))
    _ops.raster_events = None
    assert img.dtype == torch.uint8
    algo = B * A * 3 * RES * RES
    achieved = algo / (ms * 1e-3) / 1e9
    traffic, note = stamped_traffic(B, A, 'u8')
    del sim, img
    torch.cuda.empty_cache()
    lim = valu_roof(stamped_limiter(B, A, 'u8'), ms)
    return dict(mode='uint8 output (3*H*W bytes per camera)', **lim,
                bound_note='instruction issue (VALU), not HBM: achieved / peak / frac are the HBM figures of the same launch, for comparison with the float32 mode',
                achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS,
                traffic=traffic, traffic_source=note, kernel='raster_scene_bits_kernel<uint8> (persistent launch)', avg_launch_ms=ms, algorithmic_bytes_per_launch=algo)


def low_res_mode(device, steps, warmup, B, A, res=128):
    """The headline render at a lower resolution, float32 (128 x 128: a quarter of the pixels; 64 x 64: the reference's default resolution,
    rendering/base.py:144-149 -- the same ~1 200 faces per view either way): bound by instruction issue, not by the write stream, and served
    by the split form of the bit-plane path (K3s lists the faces, K3r rasterises the lists) -- SEPARATE roofline entries, never `value`.
    `avg_launch_ms` spans everything tds_raster_scene enqueues for the call (HIP events around it)."""
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = build_simulator(B, A, device, seed=1234)
    sim.step(actions[0])
    r = Resolution(res, res)
    for _ in range(warmup):
        img = sim.render_egocentric(res=r, fov=FOV)
    torch.cuda.synchronize(device)
    _ops.raster_events = []
    for _ in range(steps):
        img = sim.render_egocentric(res=r, fov=FOV)
    torch.cuda.synchronize(device)
    ms = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events]))
    _ops.raster_events = None
    assert img.shape[-2:] == (res, res) and img.dtype == torch.float32
    algo = B * A * 3 * res * res * 4
    achieved = algo / (ms * 1e-3) / 1e9
    del sim, img
    torch.cuda.empty_cache()
    traffic, note = stamped_traffic(B, A, f'f32_{res}', res=res)
    lim = valu_roof(stamped_limiter(B, A, f'f32_{res}', res=res), ms)
    return dict(mode=f'{res}x{res} float32 output', **lim,
                bound_note='instruction issue (VALU), not HBM: achieved / peak / frac are the HBM figures of the same call, for comparison with the 256 x 256 mode',
                achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=achieved / HBM_PEAK_GBS, traffic=traffic, traffic_source=note,
                kernel='scan_faces_kernel + raster_list_bits_kernel (the split form of the bit-plane path)', avg_launch_ms=ms, algorithmic_bytes_per_launch=algo)


def valu_roof(lim, launch_ms):
    """adds `frac_of_valu_issue_roof` to a stamped_limiter() entry: the time the launch's VALU instructions take at one per 4 cycles on every SIMD
    (1 024 SIMDs at 2.4 GHz: the issue rate of wave64 on 16-lane SIMDs; PMC instruction count of the same build) over the launch time measured in
    THIS run -- the compute-side counterpart of `frac` for a launch that is bound by instruction issue"""
    if lim.get('valu_issue_ms_at_4_cycles') and launch_ms:
        lim = dict(lim, frac_of_valu_issue_roof=lim['valu_issue_ms_at_4_cycles'] / launch_ms)
    return lim


def mixed_maps_mode(device, steps, warmup, B, A, overlap, headline_ms, ring):
    """The headline loop on a batch collated from TWO towns (even scenes Town01, odd scenes Town02: the reference's collated, padded mesh batch,
    mesh.py:232-245): one device map per DISTINCT mesh (two rendering maps, two off-road maps; `_ops.group_rows` finds them by content on the
    device), served by the same launches through a map set.  Reports the set-up (grouping + map builds, until the first image exists) and the
    step beside the single-map headline of this run.  `ring`: the headline's own two output buffers -- the same physical pages under both
    measurements (what a write stream reaches depends on them, DESIGN_HISTORY.md section 4), so that the difference is the maps'."""
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.utils import Resolution
    res = Resolution(RES, RES)
    _ops.map_cache.clear()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    sim, actions, _ = build_simulator(B, A, device, seed=4321, mixed=True)
    torch.cuda.synchronize(device)
    t_build = time.perf_counter() - t0
    n0, t0 = _ops.map_creations, time.perf_counter()
    sim._scene()
    torch.cuda.synchronize(device)
    t_render_maps = time.perf_counter() - t0
    n1, t0 = _ops.map_creations, time.perf_counter()
    sim.compute_offroad()
    torch.cuda.synchronize(device)
    t_off_maps = time.perf_counter() - t0
    n2 = _ops.map_creations
    # a batch operation afterwards: the handles come from the cache
    t0 = time.perf_counter()
    half = sim.select_batch_elements(list(range(0, B, 2)) + list(range(1, B, 4)), in_place=False)
    half._scene()
    torch.cuda.synchronize(device)
    t_select = time.perf_counter() - t0
    n3 = _ops.map_creations
    del half
    sink = {}
    if overlap == 'reserved':
        sim.overlap_infractions = 'reserved'
        torch.cuda.synchronize(device)
        torch.cuda.set_stream(sim.raster_stream())

    def step(i):
        sim.step(actions[i % actions.shape[0]])
        sink['img'] = sim.render_egocentric(res=res, fov=FOV, out=ring[i % 2])
        sink['col'] = sim.compute_collision()
        sink['off'] = sim.compute_offroad()

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(device)
    _ops.raster_events = []
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    k = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events]))
    _ops.raster_events = None
    frac_lit = float((sink['img'][:8] != 0).float().mean())
    # the same loop with EVERY scene on Town02 (one map): Town02's roads are denser than Town01's (0.42 against 0.23 faces per square metre, twice
    # the faces per view), so a mixed batch is to be held against the mean of the two single-map batches, not against Town01 alone
    del sim
    sim, actions, _ = build_simulator(B, A, device, seed=4321, mixed='town02')
    if overlap == 'reserved':
        sim.overlap_infractions = 'reserved'
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(device)
    _ops.raster_events = []
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize(device)
    dt2 = (time.perf_counter() - t0) / steps
    k2 = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events]))
    _ops.raster_events = None
    torch.cuda.set_stream(torch.cuda.default_stream(device))
    del sim
    sink.clear()
    return dict(what=f'the headline loop at B={B}xA={A}, {RES}x{RES} float32, on a batch collated from Town01 (even scenes) and Town02 (odd scenes), '
                     'rendered into the headline\'s own two output buffers',
                ms_per_step=1e3 * dt, agent_steps_per_s=B * A / dt, vs_single_map_headline=None if not headline_ms else 1e3 * dt / headline_ms,
                single_map_town02=dict(ms_per_step=1e3 * dt2, dominant_kernel_ms=k2),
                vs_mean_of_the_single_map_runs_of_both_towns=None if not headline_ms else 1e3 * dt / (0.5 * (headline_ms + 1e3 * dt2)),
                dominant_kernel='raster_scene_bits_kernel', dominant_kernel_ms=k,
                dominant_kernel_frac_of_hbm_peak=B * A * ALGO_BYTES_PER_IMAGE / (k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                map_creations=dict(rendering=n1 - n0, offroad=n2 - n1, after_select_batch_elements=n3 - n2),
                setup_s=dict(build_simulator_incl_collate_and_upload=t_build, group_scenes_and_build_rendering_maps=t_render_maps,
                             group_scenes_and_build_offroad_maps_incl_first_query=t_off_maps, select_batch_elements_and_regroup=t_select),
                nonzero_pixel_share_first_scenes=frac_lit,
                overlap=overlap)


def device_identity(device, reserved_usable=None):
    """Which physical GPU this rank ran on (a SCALE record must be able to show that N ranks sat on N distinct GPUs):
    uuid and PCI address from the device properties, architecture, CU count, and whether the 'reserved' CU layout was usable there."""
    if torch.device(device).type != 'cuda':
        return dict(device=str(device), uuid=None, pci=None, arch=None, compute_units=None, reserved_usable=None, visible=os.environ.get('HIP_VISIBLE_DEVICES'))
    p = torch.cuda.get_device_properties(device)
    pci = None
    if all(hasattr(p, k) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')):
        pci = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}'
    return dict(device=str(device), name=p.name, uuid=str(getattr(p, 'uuid', None)), pci=pci, arch=getattr(p, 'gcnArchName', None),
                compute_units=p.multi_processor_count, total_memory_gb=round(p.total_memory / 2 ** 30, 1), reserved_usable=reserved_usable,
                visible=os.environ.get('HIP_VISIBLE_DEVICES'))


def distinct_devices(reports):
    """number of distinct physical devices among the ranks' reports (by uuid, else PCI address; None when no rank could tell)"""
    ids = [(r.get('device_identity') or {}).get('uuid') or (r.get('device_identity') or {}).get('pci') for r in reports]
    ids = [i for i in ids if i and i != 'None']
    return len(set(ids)) if ids else None


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch(args, argv):
    """`python bench.py --gpus N` by itself: start one worker per GPU and wait.  This process makes NO GPU call (it would otherwise
    hold a context on device 0 beside rank 0's)."""
    n = args.gpus
    visible = os.environ.get('HIP_VISIBLE_DEVICES') or os.environ.get('CUDA_VISIBLE_DEVICES')
    ids = [v.strip() for v in visible.split(',') if v.strip()] if visible else [str(i) for i in range(n)]
    if len(ids) < n:
        raise SystemExit(f'--gpus {n} but only {len(ids)} devices are visible ({visible})')
    port = _free_port()
    procs = []
    import tempfile
    # rank 0's stdout goes to a file, not a pipe: whatever it prints (library warnings, NCCL_DEBUG=INFO ...) can never fill a pipe buffer
    # and block it in write() while the other ranks wait in a barrier
    out_file = tempfile.TemporaryFile()
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK='0', WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HIP_VISIBLE_DEVICES=ids[r], TDS_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        env.pop('CUDA_VISIBLE_DEVICES', None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv + ['--worker'], env=env,
                                      stdout=out_file if r == 0 else subprocess.DEVNULL))
    out0 = b''
    deadline = time.time() + args.launch_timeout
    failed = None
    try:
        # poll: a worker that dies must not leave the others waiting in a barrier until the timeout
        while failed is None and any(pr.poll() is None for pr in procs):
            for r, pr in enumerate(procs):
                rc = pr.poll()
                if rc is not None and rc != 0:
                    failed = (r, rc)
            if time.time() > deadline:
                failed = ('timeout', -1)
            time.sleep(0.2)
        for r, pr in enumerate(procs):
            if failed is None and pr.poll() not in (None, 0):
                failed = (r, pr.poll())
    finally:
        for pr in procs:                        # exact PIDs of the children this process started, nothing else
            if pr.poll() is None:
                pr.kill()
        for pr in procs:
            try:
                pr.wait(timeout=10)
            except Exception:                   # noqa: BLE001
                pass
        out_file.seek(0)
        out0 = out_file.read()
        out_file.close()
    for ln in out0.decode().splitlines():       # rank 0's JSON line to stdout, anything else it printed to stderr
        (sys.stdout if ln.lstrip().startswith('{') else sys.stderr).write(ln + '\n')
    sys.stdout.flush()
    if failed is not None:
        raise SystemExit(f'bench worker {failed[0]} failed (exit code {failed[1]})')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='scenes per GPU')
    ap.add_argument('--agents', type=int, default=64)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help="skip the extra `configs` / `roofline_u8` entries (BASELINE.json's configs 2, 3, 5; N = 1 only)")
    ap.add_argument('--cpu-scenes', type=int, default=768, help='scenes of the CPU-baseline sample (all usable cores, about 10 s at 16 of them); the single-thread run uses 16')
    ap.add_argument('--overlap', choices=('reserved', 'stream', 'off'), default='reserved',
                    help="where compute_collision / compute_offroad run: 'reserved' (default) -- the loop runs on a stream that is kept off 4 CUs per XCD and the "
                         "metrics on a stream confined to those 32 CUs, beside the raster launch (Simulator.overlap_infractions = 'reserved'); 'stream' -- a plain "
                         "second stream (slower: nothing fits beside the persistent launch); 'off' -- behind the launch on the caller's stream")
    ap.add_argument('--no-default-path', action='store_true', help='skip the `default_path` entry (the loop without out=; N = 1 only)')
    ap.add_argument('--ring-candidates', type=int, default=4, help='output allocations probed at most for the two-buffer image ring')
    ap.add_argument('--dry-run', action='store_true', help='no GPU, no kernels: exercises launch, barrier, reduction and the JSON line only')
    ap.add_argument('--launch-timeout', type=float, default=1500.0)
    ap.add_argument('--worker', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus > 1 and world == 1 and not args.worker:
        return launch(args, [a for a in sys.argv[1:]])
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    distributed = world > 1
    backend = os.environ.get('TDS_BENCH_BACKEND', 'nccl')
    if args.dry_run:
        if os.environ.get('TDS_BENCH_DRY_RUN_FAIL_RANK') == str(rank):      # tests/test_bench_launcher.py: a worker that dies at start-up
            raise SystemExit(3)
        # (TDS_BENCH_DRY_RUN_TRY_NCCL=1: keep asking for RCCL -- on a CPU it cannot come up, which exercises the fallback to gloo below)
        device, backend = torch.device('cpu'), ('nccl' if os.environ.get('TDS_BENCH_DRY_RUN_TRY_NCCL') == '1' else 'gloo')
    else:
        n_dev = torch.cuda.device_count()
        if n_dev == 0:
            raise SystemExit('bench.py needs an MI355X (use --dry-run for the launch path alone)')
        if local_rank >= n_dev:
            # fewer devices than local ranks: only when the multi-process path is exercised on a smaller box (tests/test_gpu_bench_workers.py
            # runs two ranks on the one GPU it has); on the 8-GPU node every rank has its own device
            sys.stderr.write(f'[bench] rank {rank}: local rank {local_rank} but {n_dev} visible device(s): sharing device {local_rank % n_dev}\n')
        local_rank = local_rank % n_dev
        torch.cuda.set_device(local_rank)
        device = torch.device('cuda', local_rank)
    rccl_group = None
    if distributed:
        import datetime
        import torch.distributed as dist
        # the transports announce themselves on fd 1 ("[Gloo] Rank 0 is connected to ..."): stdout carries the ONE JSON line and nothing else
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')        # one node: gloo must not depend on the host name resolving
            # No collective is on the data path: the process group carries the barrier around the timed region and the reductions of the
            # timings.  Those reductions ALWAYS run over a gloo group created first (host tensors).  Under an external launcher the barrier is
            # then moved to RCCL -- if and only if RCCL came up on EVERY rank, which the ranks agree on over gloo: a rank on which RCCL fails
            # fast can no longer leave the others inside an RCCL rendezvous, and the attempt is bounded by a short timeout.
            dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=600))
            dist.barrier()
            if backend == 'nccl':
                os.environ.setdefault('TORCH_NCCL_BLOCKING_WAIT', '1')            # a timeout raises in this thread instead of aborting the process
                os.environ.setdefault('TORCH_NCCL_ASYNC_ERROR_HANDLING', '0')
                ok, why = 1, ''
                try:
                    rccl_group = dist.new_group(backend='nccl', timeout=datetime.timedelta(seconds=float(os.environ.get('TDS_BENCH_RCCL_TIMEOUT', '90'))))
                    dist.barrier(group=rccl_group, device_ids=[local_rank])
                    torch.cuda.synchronize(device)
                except Exception as exc:                                 # noqa: BLE001 -- whatever RCCL raises
                    ok, why = 0, f'{type(exc).__name__}: {exc}'
                vote = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(vote, op=dist.ReduceOp.MIN)              # over gloo
                if int(vote.item()) == 0:
                    sys.stderr.write(f'[bench] rank {rank}: RCCL process group ' + (f'failed ({why})' if not ok else 'failed on another rank') +
                                     '; falling back to gloo for the barrier\n')
                    rccl_group, backend = None, 'gloo'
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    red_dev = 'cpu'        # the (timing-only) reductions live on the host, over gloo

    B, A = args.batch, args.agents
    from torchdrivesim_amd import parallel
    sink = {}
    if args.dry_run:
        host = None

        def step(i):
            sink['i'] = i
    else:
        sim, actions, host = build_simulator(B, A, device, seed=1234 + rank)
        from torchdrivesim_amd.utils import Resolution
        res = Resolution(RES, RES)
        from torchdrivesim_amd import _ops

        # The images go to two caller-owned buffers, allocated once and used in turn (step i renders while the consumer of step i - 1 still
        # holds the other one): a training loop owns its observation ring.  The reference allocates per call (rendering/cv2.py:52).  What the
        # write stream of the launch reaches depends on the PHYSICAL pages under the buffer (about one 51.5 GB hipMalloc in three is served
        # at 7/8 of the rate for as long as it lives; DESIGN_HISTORY.md section 4): the buffers come from the library's allocator, which spreads the
        # pages out (csrc/alloc.hip), and allocate_image_ring still measures every candidate against the fill_ rate of the same run; what
        # it saw is reported per rank (`roofline.ring_probe`, `per_rank`).  The reference-shaped call without `out=` is measured after the
        # timed region and reported beside (`default_path`).
        from torchdrivesim_amd.rendering import allocate_image_ring
        # The metrics run BESIDE the raster launch: the loop's stream (sim.raster_stream()) is kept off four CUs per XCD -- the write-bound launch loses
        # nothing on 224 of 256 CUs -- and the metric kernels, foreseen from the previous step, are enqueued on a stream confined to those 32 CUs
        # (tds_stream_create, hipExtStreamCreateWithCUMask; DESIGN_HISTORY.md section 4).  Same kernels, same bits; `--overlap off` puts them behind the launch.
        sim.overlap_infractions = {'reserved': 'reserved', 'stream': True, 'off': False}[args.overlap]
        loop_stream = sim.raster_stream() if args.overlap == 'reserved' else torch.cuda.current_stream(device)
        torch.cuda.synchronize(device)
        torch.cuda.set_stream(loop_stream)
        bufs, ring_probe = allocate_image_ring(lambda out: sim.render_egocentric(res=res, fov=FOV, out=out), (B, A, 3, RES, RES), torch.float32, device,
                                               count=2, candidates=args.ring_candidates)
        first_touch_ms = float(np.mean(ring_probe['first_touch_ms']))

        def step(i):
            sim.step(actions[i % actions.shape[0]])
            sink['img'] = sim.render_egocentric(res=res, fov=FOV, out=bufs[i % 2])
            sink['col'] = sim.compute_collision()
            sink['off'] = sim.compute_offroad()

    def barrier():
        # dist.barrier() when there is a process group (over RCCL where it came up, else gloo), then torch.cuda.synchronize()
        parallel.barrier(device, group=rccl_group, device_ids=None if rccl_group is None else [local_rank])

    for i in range(args.warmup):
        step(i)
    barrier()
    first_call = None
    if not args.dry_run:
        _ops.raster_events = []             # HIP events around every raster launch of the timed region
        first_call = _ops.raster_calls      # which raster launches of this process the timed region is (tools/make_profiles.py cuts a trace to them)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    mine = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(mine, red_dev)        # slowest rank
    per_rank = parallel.gather_over_ranks(mine, red_dev)
    raster_ms = None
    if not args.dry_run:
        raster_ms = float(np.mean([a.elapsed_time(b) for a, b in _ops.raster_events])) if _ops.raster_events else None
        _ops.raster_events = None
    # every rank's own figures (a rank on slow memory would otherwise show only as a low rate): reporting only, over gloo
    reserved_ok = None
    if not args.dry_run and args.overlap == 'reserved':
        reserved_ok = bool(sim._reserved_usable(device))
    mine_report = dict(rank=rank, seconds=mine, agent_steps_per_s=B * A * args.steps / mine, avg_launch_ms=raster_ms, device_identity=device_identity(device, reserved_ok),
                       ring_probe=None if args.dry_run else dict(launch_ms=ring_probe['launch_ms'], fill_ms=ring_probe['fill_ms'], fast=ring_probe['fast'],
                                                                kept=ring_probe['kept'], aliased=ring_probe['aliased']))
    all_reports = parallel.gather_objects_over_ranks(mine_report)

    if rank == 0:
        value = world * B * A * args.steps / elapsed
        achieved = (B * A * ALGO_BYTES_PER_IMAGE) / (raster_ms * 1e-3) / 1e9 if raster_ms else None
        traffic, traffic_note = stamped_traffic(B, A)
        same_run = {}
        if not args.dry_run:
            same_run = same_run_write_roofs(sim, bufs[0], res, device)
            same_run['first_touch_ms'] = first_touch_ms
            same_run['ring_probe'] = ring_probe
            same_run['timed_raster_calls'] = [first_call, first_call + args.steps]     # [first, end) among this process's raster launches
        line = dict(
            metric='agent-steps/sec (whole node) at B=1024xA=64, 256x256 BEV', value=value, unit='agent-steps/s', n_gpus=world,
            steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / max(args.steps, 1), higher_is_better=True, scaling='weak',
            vs_baseline=None, dtype='f32', data='synthetic',
            config=dict(workload=f'B={B}xA={A} per GPU, carla_Town01 mesh (30750 faces): KinematicBicycle.step + render_egocentric {RES}x{RES} '
                                 f'fp32 fov {FOV:g} m + compute_collision(iou) + compute_offroad', global_batch=world * B, agents=A, res=RES,
                        parallelism=f'scene-batch sharding x{world}, no collectives'),
            roofline=dict(bound='hbm', achieved=achieved, peak=HBM_PEAK_GBS, unit='GB/s', frac=None if achieved is None else achieved / HBM_PEAK_GBS,
                          traffic=traffic, traffic_source=traffic_note,
                          kernel='raster_scene_bits_kernel', avg_launch_ms=raster_ms, algorithmic_bytes_per_launch=B * A * ALGO_BYTES_PER_IMAGE,
                          **same_run))
        line['methodology'] = ('images are rendered into a two-buffer ring the loop owns (rendering.allocate_image_ring: buffers from the library\'s '
                               'allocator, each measured against the fill_ rate of the same run; `roofline.ring_probe`); `default_path` is the same '
                               'loop with the reference-shaped call render_egocentric() without out= (a fresh tensor per step from the image pool); '
                               + {'reserved': 'the loop runs on a stream kept off 4 CUs per XCD (224 of 256 CUs for the raster launch) and the metric kernels on a '
                                              'stream confined to those 32 CUs, beside the launch (Simulator.overlap_infractions = \'reserved\'; --overlap off: behind it)',
                                  'stream': 'the metric kernels on a plain second stream', 'off': 'the metric kernels behind the raster launch on the same stream'}[args.overlap])
        line['overlap'] = args.overlap
        line['distinct_devices'] = distinct_devices(all_reports)          # N ranks on N GPUs <=> distinct_devices == n_gpus
        if world == 1:
            line['device_identity'] = all_reports[0]['device_identity']
        if world > 1:
            line['per_rank_agent_steps_per_s'] = [B * A * args.steps / t for t in per_rank]
            line['per_rank'] = all_reports
            line['launcher'] = ('bench.py self-launch, one child per GPU' if args.worker else 'external launcher (torch.distributed.run)') + \
                (', RCCL barrier' if backend == 'nccl' else ', gloo barrier on 127.0.0.1')
        if args.dry_run:
            line['dry_run'] = True
            line['data'] = 'none (dry run: no kernels were launched, the value is meaningless)'
        if world == 1 and not args.dry_run:
            if not args.no_configs:
                line['mixed_maps'] = mixed_maps_mode(device, args.steps, args.warmup, B, A, args.overlap, line['ms_per_step'], bufs)
                torch.cuda.set_stream(loop_stream)
            if not args.no_default_path:
                # the reference-shaped call: no `out=`, a fresh image tensor per step (rendering/cv2.py:52 allocates per call); the previous
                # image is still held when the next one is allocated, as a consumer would
                del bufs
                sink.clear()
                torch.cuda.empty_cache()

                def default_step(i):
                    sim.step(actions[i % actions.shape[0]])
                    sink['img'] = sim.render_egocentric(res=res, fov=FOV)
                    sink['col'] = sim.compute_collision()
                    sink['off'] = sim.compute_offroad()

                for i in range(args.warmup):
                    default_step(i)
                torch.cuda.synchronize(device)
                _ops.raster_events = []
                t1 = time.perf_counter()
                for i in range(args.steps):
                    default_step(args.warmup + i)
                torch.cuda.synchronize(device)
                dt = (time.perf_counter() - t1) / args.steps
                ms = [a.elapsed_time(b) for a, b in _ops.raster_events]
                _ops.raster_events = None
                line['default_path'] = dict(what='the same loop with render_egocentric() without out=: a fresh image tensor per step',
                                            ms_per_step=1e3 * dt, agent_steps_per_s=B * A / dt, avg_launch_ms=float(np.mean(ms)),
                                            min_launch_ms=float(np.min(ms)), max_launch_ms=float(np.max(ms)),
                                            image_allocation='torch memory pool over tds_torch_alloc (csrc/alloc.hip)' if _ops.use_image_pool else 'torch.empty')
                bufs = None
            torch.cuda.synchronize(device)
            torch.cuda.set_stream(torch.cuda.default_stream(device))      # the other entries build their own simulators on the default stream, all CUs
            if not args.no_configs:
                line['roofline_u8'] = u8_mode(device, args.steps, args.warmup, B, A)
                line['roofline_128'] = low_res_mode(device, args.steps, args.warmup, B, A)
                line['roofline_64'] = low_res_mode(device, args.steps, args.warmup, B, A, res=64)
                del sim, bufs
                sink.clear()
                _ops.release_image_pool()
                torch.cuda.empty_cache()
                line['configs'] = other_configs(device, args.steps, args.warmup, overlap=args.overlap)
            if not args.no_cpu_baseline:
                line['cpu_baseline'] = cpu_baseline(host, min(args.cpu_scenes, B), A)
        print(json.dumps(line), flush=True)
    if distributed:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
