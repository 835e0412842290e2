#!/usr/bin/env python3
"""Randomised parity run of the K3 scene rasteriser in the modes tests/fuzz_raster.py does not reach (a script, not collected by pytest):
three agent types (seven distinct keys: at 256 x 256 the bit-plane kernel then renders the whole image in two 8-wave workgroups per CU -- round 6;
two half-image strips before --, differentiable calls in 4-wave ones), with and without the trim
rule (CV2RendererConfig.trim_mesh_before_rendering), and the key-index slices of differentiable calls decoded back into the image.
   python tests/fuzz_raster_modes.py [--seeds 6] [--batch 4] [--agents 24]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))      # this file lives in tests/: the oracle is test infrastructure
import test_gpu_parity as T                                                      # noqa: E402
from test_gpu_parity import dev, sc_np, pack                                     # noqa: E402
from torchdrivesim_amd import _ops as ops                                        # noqa: E402
from oracle import oracle                                                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seeds', type=int, default=6); ap.add_argument('--batch', type=int, default=4); ap.add_argument('--agents', type=int, default=24)
a = ap.parse_args()
TYPES = dict(vehicle=(4, (32, 74, 135)), bicycle=(5, (255, 150, 40)), pedestrian=(6, (255, 64, 180)))
levels = sorted(set(T.LEVEL_TABLE) | {float(z) for z, _ in TYPES.values()}, reverse=True)
t = np.load(os.path.join(ROOT, 'tests', 'golden', 'town01_mesh.npz'))
cats = [str(c) for c in t['categories']]
cat = t['vert_category'][t['faces'][:, 0]]
smap = ops.StaticMap(t['verts'], t['faces'], np.array([T.LEVELS[cats[c]] for c in cat], np.float32),
                     np.array([pack(T.COLORS[cats[c]]) for c in cat], np.uint32), levels, device=T.DEV)
static = oracle.static_mesh_arrays(t['verts'], t['faces'], t['vert_category'], cats, colors={**oracle.DEFAULT_COLORS, **T.COLORS}, levels={**oracle.DEFAULT_LEVELS, **T.LEVELS})
road = t['verts'][t['vert_category'] == cats.index('road')]
names = list(TYPES)
bad_total = n_img = 0
t0 = time.time()
for seed in range(a.seeds):
    for res, fov in ((256, 35.0), (128, 50.0), (320, 35.0)):
        gen = np.random.default_rng(7000 + seed)
        B, A = a.batch, a.agents
        anchor = road[gen.integers(0, len(road), (B, 1))]
        state = np.concatenate([anchor + gen.uniform(-25, 25, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
        size = np.concatenate([gen.uniform(1.0, 9, (B, A, 1)), gen.uniform(0.6, 3.0, (B, A, 1))], -1).astype(np.float32)
        kind = gen.integers(0, 3, (B, A))
        mask = np.ascontiguousarray((gen.uniform(size=(B, 1, A)) < 0.85) & (gen.uniform(size=(B, A, A)) < 0.95))
        body = np.array([(smap.rank_of(TYPES[n][0]) << 24) | pack(TYPES[n][1]) for n in names], np.int64)
        dkey = (smap.rank_of(T.LEVELS['direction']) << 24) | pack(T.COLORS['direction'])
        keys = torch.from_numpy(np.stack([body[kind], np.full_like(kind, dkey)], -1)).to(torch.int32).to(T.DEV)
        lev = np.stack([np.array([TYPES[n][0] for n in names], np.float32)[kind], np.full(kind.shape, T.LEVELS['direction'], np.float32)], -1)
        col = np.stack([np.array([TYPES[n][1] for n in names], np.float32)[kind], np.broadcast_to(np.array(T.COLORS['direction'], np.float32), kind.shape + (3,))], -2) / np.float32(255.0)
        sd = dev(state)
        agent_sc = ops.heading_sc(sd[..., 2])
        tmpl = dev(oracle.actor_template(size))
        for trim in (True, False):
            img, slices, ktab = ops.raster_scene(smap, sd, agent_sc, tmpl, keys, dev(mask), dev(state[..., :2].copy()), agent_sc, fov, res, trim=trim,
                                                  index_slices=True)
            oracle.set_trim_mesh(trim)
            try:
                ref = oracle.render_scenes(state, size, mask, state[..., :2].copy(), sc_np(agent_sc), *static, fov, res, agent_sc=sc_np(agent_sc),
                                           actor_levels=lev, actor_colors=col.astype(np.float32))
            finally:
                oracle.set_trim_mesh(True)
            out = img.cpu().numpy()
            bad = int((out != ref).sum())
            # the same scene without index slices, float32 and uint8: at 256 x 256 seven keys take the whole image in 8-WAVE workgroups (round 6),
            # the differentiable call above in 4-wave ones
            for dt in (torch.float32, torch.uint8):
                plain = ops.raster_scene(smap, sd, agent_sc, tmpl, keys, dev(mask), dev(state[..., :2].copy()), agent_sc, fov, res, dt, trim=trim)
                bad += int((plain.cpu().numpy().astype(np.float32) != ref).sum())
            # the slices decode to the same image
            dec_bad = -1
            if slices is not None:
                wpr, nb = (res + 31) // 32, 4 if len(ktab) > 7 else (3 if len(ktab) > 3 else 2)
                sl = slices.view(B * A, wpr, res // 4, 4, 4).cpu().numpy().astype(np.uint32)
                xs = np.arange(res)
                idx = np.zeros((B * A, res, res), np.int64)
                for bit in range(nb):
                    words = sl[:, :, :, bit, :].reshape(B * A, wpr, res)
                    idx |= (((words[:, xs // 32, :] >> (xs % 32)[None, :, None].astype(np.uint32)) & 1).astype(np.int64)) << bit
                table = np.array([[0, 0, 0]] + [[(k >> 16) & 255, (k >> 8) & 255, k & 255] for k in ktab], np.float32)
                dec_bad = int((table[idx].transpose(0, 3, 1, 2).reshape(out.shape) != out).sum())
            bad_total += bad + max(dec_bad, 0); n_img += B * A
            print(f'seed {seed} res {res} trim {trim}: {B * A} images, {len(ktab or [])} keys, {bad} differing values, slices {dec_bad} ({time.time() - t0:.0f} s)', flush=True)
print('TOTAL', n_img, 'images,', bad_total, 'differing values')
sys.exit(1 if bad_total else 0)
