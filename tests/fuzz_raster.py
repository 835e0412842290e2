#!/usr/bin/env python3
"""Large randomised parity run of the K3 scene rasteriser against the oracle (a script, not collected by pytest: minutes of CPU time).
   python tests/fuzz_raster.py [--seeds 8] [--seed0 0] [--batch 8] [--agents 24] [--res 256] [--map town01|town02|mixed] [--u8]
   --map mixed (round 6): every scene of a batch is on Town01 or Town02 at random -- ONE launch through a map set (tds_raster_scene_multi), the oracle
   renders every scene with its own town's mesh.
   --res 0: a third family -- every seed draws its own resolution from 4 .. 60 (multiples of 4 or not: the one-pixel-per-thread
   write-out) and its own field of view from 5 .. 200 m."""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))      # this file lives in tests/: the oracle is test infrastructure
from test_gpu_parity import actor_keys, dev, make_map, oracle_static, render_both, sc_np    # noqa: E402
from torchdrivesim_amd import _ops as ops                                        # noqa: E402
from oracle import oracle                                                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seeds', type=int, default=8); ap.add_argument('--batch', type=int, default=8); ap.add_argument('--agents', type=int, default=24)
ap.add_argument('--res', type=int, default=256); ap.add_argument('--fov', type=float, default=35.0)
ap.add_argument('--map', default='town01'); ap.add_argument('--u8', action='store_true')
ap.add_argument('--seed0', type=int, default=0, help='first seed (a second run with other scenes)')
a = ap.parse_args()
def load_town(name):
    t = np.load(os.path.join(ROOT, 'tests', 'golden', f'{name}_mesh.npz'))
    town = dict(verts=t['verts'], faces=t['faces'], vert_category=t['vert_category'], categories=[str(c) for c in t['categories']])
    return (make_map(ops, town['verts'], town['faces'], town['vert_category'], town['categories']),
            oracle_static(oracle, town['verts'], town['faces'], town['vert_category'], town['categories']),
            town['verts'][town['vert_category'] == town['categories'].index('road')])


towns = [load_town(n) for n in (('town01', 'town02') if a.map == 'mixed' else (a.map,))]
smap, static, road = towns[0]
bad_total = n_img = 0
t0 = time.time()
for seed in range(a.seeds):
    gen = np.random.default_rng(1000 + a.seed0 + seed)
    B, A = a.batch, a.agents
    which = gen.integers(0, len(towns), B)
    anchor = np.stack([towns[w][2][gen.integers(0, len(towns[w][2]))] for w in which])[:, None]
    xy = anchor + gen.uniform(-25, 25, (B, A, 2))
    state = np.concatenate([xy, gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 10, (B, A, 1))], -1).astype(np.float32)
    size = np.concatenate([gen.uniform(3.5, 12, (B, A, 1)), gen.uniform(1.6, 3.0, (B, A, 1))], -1).astype(np.float32)
    present = gen.uniform(size=(B, A)) < 0.85
    mask = np.ascontiguousarray(present[:, None, :] & (gen.uniform(size=(B, A, A)) < 0.95))
    cam_sc = sc_np(ops.heading_sc(dev(state)[..., 2]))
    res, fov = (a.res, a.fov) if a.res > 0 else (int(gen.integers(4, 61)), float(np.exp(gen.uniform(np.log(5.0), np.log(200.0)))))
    dtype = torch.uint8 if a.u8 else torch.float32
    if len(towns) == 1:
        img, ref = render_both(ops, oracle, smap, static, state, size, mask, state[..., :2].copy(), cam_sc, fov, res, dtype)
    else:
        mset = ops.StaticMapSet([tw[0] for tw in towns], torch.from_numpy(which.astype(np.int32)))
        sd = dev(state)
        agent_sc = ops.heading_sc(sd[..., 2])
        img = ops.raster_scene(mset, sd, agent_sc, dev(oracle.actor_template(size)), actor_keys(mset, B, A), dev(mask), dev(state[..., :2].copy()), dev(cam_sc), fov, res, dtype).cpu().numpy()
        ref = np.empty(img.shape, np.float32)
        for w in range(len(towns)):
            idx = np.nonzero(which == w)[0]
            if len(idx):
                sv, sa, sf = towns[w][1]
                ref[idx] = oracle.render_scenes(state[idx], size[idx], np.ascontiguousarray(mask[idx]), state[idx][..., :2].copy(), cam_sc[idx], sv, sa, sf, fov, res, agent_sc=sc_np(agent_sc)[idx])
    bad = (img.astype(np.float32) != ref)
    per_img = bad.reshape(B * A, -1).any(1).sum()
    bad_total += int(bad.sum()); n_img += B * A
    print(f'seed {seed}: {B * A} images {res} x {res} fov {fov:.1f}, {int(bad.sum())} differing values in {int(per_img)} images ({time.time() - t0:.0f} s)', flush=True)
print('TOTAL', n_img, 'images,', bad_total, 'differing values')
sys.exit(1 if bad_total else 0)
