#!/usr/bin/env python3
"""How often is the union of the two fills of a lane-marking QUAD (two same-colour triangles that share an edge) what a quad-shaped rule
would paint?  (four fifths of a view's faces are such pairs, set up and painted twice over.)  CPU only: the oracle's
call lists (what the reference hands to cv2.fillConvexPoly, pixel coordinates) of Town01 views at 64 / 128 / 256 pixels; every pair of
consecutive same-colour faces that share two (pixel-space) vertices is filled triangle by triangle with the oracle's cv::fillConvexPoly and
compared, row by row, with the interval [min of the left ends, max of the right ends]: "simple" = the union is that interval in every row.
   python tools/quad_union_stats.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
verts, faces, vcat, cats = bench.load_town01()
B, A = 1, 64
state, size, present, actions = bench.synth_agents(B, A, verts[vcat == cats.index('road')], 1234)
sv, sa, sf = orc.static_mesh_arrays(verts, faces, vcat, cats)
sc = np.stack([np.sin(state[..., 2]), np.cos(state[..., 2])], -1).astype(np.float32)
mask = np.ascontiguousarray(np.broadcast_to(present[:, None, :], (B, A, A)))
for res in (64, 128, 256):
    _, tris, cols, cnt = orc.render_scenes(state, size, mask, state[..., :2].copy(), sc, sv, sa, sf, 35.0, res, agent_sc=sc, record=True, images=False)
    n_faces = n_pairs = simple = gaps = inside_pairs = quads = quad_equal = quad_missing = quad_extra = 0
    for i in range(B * A):
        t = tris[i, :cnt[i]].reshape(-1, 3, 2)
        c = cols[i, :cnt[i]]
        n_faces += len(t)
        j = 0
        while j + 1 < len(t):
            a, b = t[j], t[j + 1]
            shared = len({tuple(p) for p in a} & {tuple(p) for p in b})
            if shared >= 2 and tuple(c[j]) == tuple(c[j + 1]):
                n_pairs += 1
                pts = np.concatenate([a, b])
                x0, y0, x1, y1 = pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()
                if x0 >= 0 and y0 >= 0 and x1 < res and y1 < res:
                    inside_pairs += 1
                    w, h = x1 - x0 + 3, y1 - y0 + 3
                    off = np.array([x0 - 1, y0 - 1])
                    ia = orc.fill_convex_poly(np.zeros((h, w, 3), np.float32), a - off, (1, 0, 0))[..., 0] > 0
                    ib = orc.fill_convex_poly(np.zeros((h, w, 3), np.float32), b - off, (1, 0, 0))[..., 0] > 0
                    un = ia | ib
                    ok = True
                    for y in range(h):
                        xs = np.nonzero(un[y])[0]
                        if len(xs) and not un[y, xs[0]:xs[-1] + 1].all():
                            ok = False
                            gaps += 1
                            break
                    simple += ok
                    # the same pair as ONE call of cv::fillConvexPoly with the four points in order around the quad (p, s1, q, s2)
                    sa_, sb_ = [tuple(p_) for p_ in a], [tuple(p_) for p_ in b]
                    sh = [p_ for p_ in sa_ if p_ in sb_]
                    pa = [p_ for p_ in sa_ if p_ not in sh]
                    pb = [p_ for p_ in sb_ if p_ not in sh]
                    if len(sh) == 2 and len(pa) == 1 and len(pb) == 1:
                        quad = np.array([pa[0], sh[0], pb[0], sh[1]], np.int32) - off
                        iq = orc.fill_convex_poly(np.zeros((h, w, 3), np.float32), quad, (1, 0, 0))[..., 0] > 0
                        quads += 1
                        quad_equal += bool(np.array_equal(iq, un))
                        quad_missing += int((un & ~iq).sum()); quad_extra += int((iq & ~un).sum())
                j += 2
            else:
                j += 1
    print(f'res {res}: {n_faces / (B * A):.0f} faces per view, {2 * n_pairs / n_faces:.2f} of them in same-colour pairs that share an edge; of the pairs {inside_pairs / max(n_pairs, 1):.2f} '
          f'lie inside the image; of those the union of the two fills is ONE interval in every row for {simple / max(inside_pairs, 1):.4f} ({gaps} pairs with a gap); '
          f'cv::fillConvexPoly of the quad (4 points) equals the union for {quad_equal} of {quads} proper quads ({quad_missing} pixels missing, {quad_extra} extra in all)')
