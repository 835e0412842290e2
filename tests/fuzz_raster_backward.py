#!/usr/bin/env python3
"""Randomised run of the two K3 backward kernels against each other on Town01 scenes (a script, not collected by pytest): the gradients
computed from the forward's key-index slices (tds_raster_scene_bwd_idx_f32: sparse reads next to colour boundaries) must equal those
computed from the forward image (tds_raster_scene_bwd_f32: the round-1 kernel, itself checked against the numpy definition), and the colour
gradient must equal the per-key sums over the decoded slices exactly.
   python tests/fuzz_raster_backward.py [--seeds 10] [--batch 3] [--agents 16]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))      # this file lives in tests/: the oracle is test infrastructure
import test_gpu_parity as T                                                      # noqa: E402
from test_gpu_parity import dev, actor_keys, make_map                            # noqa: E402
from torchdrivesim_amd import _ops as ops                                        # noqa: E402
from oracle import oracle                                                        # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--seeds', type=int, default=10); ap.add_argument('--batch', type=int, default=3); ap.add_argument('--agents', type=int, default=16)
a = ap.parse_args()
t = np.load(os.path.join(ROOT, 'tests', 'golden', 'town01_mesh.npz'))
cats = [str(c) for c in t['categories']]
smap = make_map(ops, t['verts'], t['faces'], t['vert_category'], cats)
road = t['verts'][t['vert_category'] == cats.index('road')]
worst, n_img, t0 = 0.0, 0, time.time()
for seed in range(a.seeds):
    for res, fov in ((64, 35.0), (128, 50.0), (256, 35.0), (320, 35.0)):
        gen = np.random.default_rng(9000 + seed)
        B, A = a.batch, a.agents
        anchor = road[gen.integers(0, len(road), (B, 1))]
        state = np.concatenate([anchor + gen.uniform(-20, 20, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
        size = np.concatenate([gen.uniform(3.5, 9, (B, A, 1)), gen.uniform(1.6, 3.0, (B, A, 1))], -1).astype(np.float32)
        mask = np.ascontiguousarray((gen.uniform(size=(B, 1, A)) < 0.9) & (gen.uniform(size=(B, A, A)) < 0.95))
        tmpl = dev(oracle.actor_template(size))
        keys = actor_keys(smap, B, A)
        grads, gout = [], None
        for use in (True, False):
            ops.use_index_slices = use
            try:
                st = dev(state).requires_grad_(True)
                cxy = dev(state[..., :2].copy()).requires_grad_(True)
                sc = ops.heading_sc(st[..., 2]).detach().requires_grad_(True)
                csc = sc.detach().clone().requires_grad_(True)
                img = ops.raster_scene_diff(smap, st, sc, tmpl, keys, dev(mask), cxy, csc, fov, res)
                if gout is None:
                    gout = torch.randn(img.shape, device=T.DEV, generator=torch.Generator(device=T.DEV).manual_seed(seed))
                img.backward(gout)
                grads.append([x.grad.double().cpu().numpy() for x in (st, sc, cxy, csc)])
            finally:
                ops.use_index_slices = True
        err = 0.0
        for x, y in zip(*grads):
            err = max(err, float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-9)))
        # colour gradient: integer-valued incoming gradient -> exact sums
        img, slices, ktab = ops.raster_scene(smap, dev(state), ops.heading_sc(dev(state)[..., 2]), tmpl, keys, dev(mask), dev(state[..., :2].copy()),
                                              ops.heading_sc(dev(state)[..., 2]), fov, res, index_slices=True)
        kc = torch.tensor([[(k >> 16) & 255, (k >> 8) & 255, k & 255] for k in ktab], dtype=torch.float32, device=T.DEV, requires_grad=True)
        st2 = dev(state).requires_grad_(True)
        out = ops.raster_scene_diff(smap, st2, ops.heading_sc(st2[..., 2]), tmpl, keys, dev(mask), dev(state[..., :2].copy()), ops.heading_sc(dev(state)[..., 2]),
                                    fov, res, key_colors=kc, color_keys=ktab)
        gi = torch.randint(-3, 4, out.shape, device=T.DEV, generator=torch.Generator(device=T.DEV).manual_seed(seed + 1)).float()
        out.backward(gi)
        table = torch.tensor([[0, 0, 0]] + [[(k >> 16) & 255, (k >> 8) & 255, k & 255] for k in ktab], dtype=torch.float32, device=T.DEV)
        # per key: sum of gi over the pixels that show the key's colour AND rank -- decoded from the image (distinct colours here)
        exp = torch.stack([torch.stack([gi[:, :, ch][(out.detach() == table[i + 1].view(1, 1, 3, 1, 1)).all(2)].sum() for ch in range(3)]) for i in range(len(ktab))])
        cbad = int((kc.grad != exp).sum())
        worst = max(worst, err); n_img += B * A
        print(f'seed {seed} res {res}: {B * A} cameras, max relative difference idx vs image backward {err:.2e}, colour gradient mismatches {cbad} ({time.time() - t0:.0f} s)', flush=True)
        if cbad:
            worst = 1.0
print('TOTAL', n_img, 'cameras, worst relative difference', f'{worst:.2e}')
sys.exit(1 if worst > 5e-4 else 0)
