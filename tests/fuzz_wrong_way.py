#!/usr/bin/env python3
"""Randomised parity run of the wrong-way kernel against the oracle's agent-by-agent restatement (a script, not collected by pytest: the
oracle is pure Python).   python tests/fuzz_wrong_way.py [--agents 1500]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lanelet_oracle as lo                                           # noqa: E402  (this file lives in tests/)
from torchdrivesim_amd import lanelet2 as L                                       # noqa: E402
from torchdrivesim_amd.infractions import lanelet_orientation_loss                # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--agents', type=int, default=1500)
a = ap.parse_args()
dev = torch.device('cuda', 0)
worst, total, t0 = 0.0, 0, time.time()
for town in ('carla_Town01', 'carla_Town02'):
    for align in (True, False):
        lanes = L.load_lanelet_map(os.path.join(ROOT, 'tests', 'golden', town + '.osm.gz'), origin=(0.0, 0.0), align_borders=align)
        # The oracle reads the file by itself (own XML walk, projection, alignment, centre lines) when the bounds are aligned as Lanelet2's loader
        # leaves them.  In FILE order (align=False) every lanelet of these maps has its left bound on the right: the centre-line construction then
        # sits on a knife edge (the first connection starts exactly on the entry gate; whether it "leaves through the gate" is decided by the last
        # bit of the midpoint) and the 6e-11 m by which the two readers' points differ flips whole centre lines -- so for that setting the oracle's
        # lanelets are built from the product's bound arrays (its own centre-line and query code still run on them).
        if align:
            own = lo.load_osm(os.path.join(ROOT, 'tests', 'golden', town + '.osm.gz'), origin=(0.0, 0.0), align=True)
        else:
            own = lo.OracleMap([lo.OracleLanelet(l.id, l.left, l.right, l.attributes) for l in lanes.laneletLayer])
        cl = np.concatenate([l.centerline for l in lanes.laneletLayer])
        for k, (tol, thr) in enumerate(((1.0, np.pi / 2), (0.0, np.pi / 2), (0.25, 2.2), (2.5, np.pi / 2))):
            g = np.random.default_rng(hash((town, align, k)) % 2 ** 32)
            n = a.agents
            xy = cl[g.integers(0, len(cl), n), :2] + g.normal(0, 2.0, (n, 2)) * g.choice([0.05, 1.0, 3.0], (n, 1))
            state = np.concatenate([xy, g.uniform(-np.pi, np.pi, (n, 1)), np.ones((n, 1))], -1).astype(np.float32)[None]
            ref = lo.lanelet_orientation_loss([own], state, None, thr, tol)
            out = lanelet_orientation_loss([lanes], torch.from_numpy(state).to(dev), direction_angle_threshold=thr, lanelet_dist_tolerance=tol).cpu().numpy()
            err = float(np.abs(out - ref).max())
            worst, total = max(worst, err), total + n
            print(f'{town} align_borders={align} tol={tol} thr={thr:.3f}: {n} agents, {int((ref > 0).sum())} wrong-way, max |hip - oracle| = {err:.2e} '
                  f'({time.time() - t0:.0f} s)', flush=True)
print('TOTAL', total, 'agents, max difference', f'{worst:.2e}')
sys.exit(1 if worst > 2e-6 else 0)
