"""
nograd_oracle.py -- CPU restatement of the `nograd` collision metric.  TEST INFRASTRUCTURE ONLY (like tds_oracle.c: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does).

Reference path, followed function by function in numpy with the reference's dtypes (float32 states -> float32 corners):
    Simulator._compute_collision_of_multi_agents    simulator.py:1111-1149   (present agents only, NPCs not counted)
    compute_agent_collisions_metric                  infractions.py:352-375
    get_all_intersections                            infractions.py:429-474   (upper triangle, `intersection(...).area != 0`)
    rectangle_vertices                               infractions.py:476-500
The one call that leaves the reference is `shapely.geometry.Polygon.intersection(...).area` (shapely is a pip dependency, unpinned in
pyproject.toml:21, absent here and not under /root/reference; no reference test exercises this metric): PARITY UNPINNED.  It is restated
twice, and the tests require the two to agree on every pair they use:
  * `clip_area`      -- the area of the intersection polygon by convex clipping (Sutherland-Hodgman) in float64, the arithmetic GEOS
                        works in; the verdict is `area != 0` exactly as infractions.py:462-464;
  * `shares_area`    -- the exact answer to "do the two rectangles share area": rational arithmetic (fractions.Fraction) on the float32
                        corners, two convex polygons share area iff no edge line of either has the whole other polygon on its outer
                        side or on the line.  Touching rectangles (a common edge, a corner on an edge) share none.
"""
from fractions import Fraction

import numpy as np


def rectangle_vertices(cx, cy, w, h, angle):
    """infractions.py:476-500, same expressions, same dtype as the inputs (float32 in the simulator's call)"""
    dx = w / 2
    dy = h / 2
    dxcos = dx * np.cos(angle)
    dxsin = dx * np.sin(angle)
    dycos = dy * np.cos(angle)
    dysin = dy * np.sin(angle)
    return np.stack([
        np.concatenate([cx, cy], axis=-1) + np.concatenate([-dxcos - -dysin, -dxsin + -dycos], axis=-1),
        np.concatenate([cx, cy], axis=-1) + np.concatenate([dxcos - -dysin, dxsin + -dycos], axis=-1),
        np.concatenate([cx, cy], axis=-1) + np.concatenate([dxcos - dysin, dxsin + dycos], axis=-1),
        np.concatenate([cx, cy], axis=-1) + np.concatenate([-dxcos - dysin, -dxsin + dycos], axis=-1)
    ], axis=1)


def clip_area(p, q) -> float:
    """area of the intersection of two convex counter-clockwise polygons, float64 Sutherland-Hodgman + shoelace"""
    out = [(float(x), float(y)) for x, y in p]
    q = [(float(x), float(y)) for x, y in q]
    for k in range(len(q)):
        if not out:
            return 0.0
        (ax, ay), (bx, by) = q[k], q[(k + 1) % len(q)]
        side = lambda pt: (bx - ax) * (pt[1] - ay) - (by - ay) * (pt[0] - ax)
        src, out = out, []
        for i in range(len(src)):
            cur, nxt = src[i], src[(i + 1) % len(src)]
            sc, sn = side(cur), side(nxt)
            if sc >= 0:
                out.append(cur)
            if (sc > 0 and sn < 0) or (sc < 0 and sn > 0):
                t = sc / (sc - sn)
                out.append((cur[0] + t * (nxt[0] - cur[0]), cur[1] + t * (nxt[1] - cur[1])))
    if len(out) < 3:
        return 0.0
    s = 0.0
    for i in range(len(out)):
        (x0, y0), (x1, y1) = out[i], out[(i + 1) % len(out)]
        s += x0 * y1 - x1 * y0
    return abs(s) / 2.0


def shares_area(p, q) -> bool:
    """exact: the interiors of two convex counter-clockwise quadrilaterals intersect"""
    P = [(Fraction(float(x)), Fraction(float(y))) for x, y in p]
    Q = [(Fraction(float(x)), Fraction(float(y))) for x, y in q]

    def separated(a, b):
        for e in range(len(a)):
            (ax, ay), (bx, by) = a[e], a[(e + 1) % len(a)]
            if all((bx - ax) * (vy - ay) - (by - ay) * (vx - ax) <= 0 for vx, vy in b):
                return True
        return False
    return not (separated(P, Q) or separated(Q, P))


def get_all_intersections(rects: np.ndarray, predicate='clip') -> np.ndarray:
    """infractions.py:429-474 (ego_idx None): upper-triangular 0/1 matrix over the given rectangles (m x 5, float32)"""
    m = len(rects)
    polys = rectangle_vertices(*np.split(rects, rects.shape[-1], axis=-1)) if m else np.zeros((0, 4, 2), rects.dtype)
    inter = np.zeros((m, m))
    for i in range(m):
        for j in range(i):
            hit = (clip_area(polys[j], polys[i]) != 0) if predicate == 'clip' else shares_area(polys[j], polys[i])
            if hit:
                inter[j, i] = 1
    return inter


def compute_agent_collisions_metric(all_rects, collision_masks, present_masks, predicate='clip') -> np.ndarray:
    """infractions.py:352-375"""
    all_scores = []
    agent_count = present_masks.shape[1]
    for batch, (rect, mask_i) in enumerate(zip(all_rects, collision_masks)):
        intersects = get_all_intersections(rect, predicate)
        intersects[~mask_i] = 0
        intersects = intersects + intersects.T - np.diag(np.diag(intersects))
        all_agent_scores = intersects.sum(axis=-1)
        padded = np.zeros(agent_count)
        padded[present_masks[batch]] = all_agent_scores
        all_scores.append(padded)
    return np.array(all_scores)


def nograd_collision(state, size, present, predicate='clip') -> np.ndarray:
    """simulator.py:1111-1149 for the default mask: (B,A,4) float32, (B,A,2) float32, (B,A) bool -> (B,A) float64 counts"""
    state, size, present = np.asarray(state, np.float32), np.asarray(size, np.float32), np.asarray(present, bool)
    boxes = [np.concatenate([state[b][present[b]][:, :2], size[b][present[b]], state[b][present[b]][:, 2:3]], axis=-1) for b in range(len(state))]
    masks = [(present[b] * present[b])[present[b]] for b in range(len(state))]
    return compute_agent_collisions_metric(boxes, masks, present, predicate)
