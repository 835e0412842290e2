"""
lanelet_oracle.py -- CPU restatement of the wrong-way path.  TEST INFRASTRUCTURE ONLY (like tds_oracle.c: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does).

Follows the reference's own control flow, agent by agent and lanelet by lanelet, in numpy float64 where the reference sits in
Lanelet2 (double) and numpy float32 where it sits in torch:
    lanelet_orientation_loss     reference infractions.py:232-304
    find_lanelet_directions      reference lanelet2.py:108-141
    find_direction               reference lanelet2.py:144-180
plus the three Lanelet2 library functions those call, restated from the library's published sources [UNVERIFIED-UPSTREAM -- Lanelet2
is a pip dependency of the reference (CI_cpu.yml:34, unpinned), not installed here and not under /root/reference]:
    lanelet2.geometry.findWithin2d(laneletLayer, p, d)   distance from p to the lanelet's outline polygon (0 inside), <= d, sorted
    lanelet2.geometry.project(linestring, p)             closest point of the line, 3-D
    lanelet.centerline                                   lanelet2_core Lanelet.cpp calculateCenterline (pure Python loops here)

The oracle also reads maps BY ITSELF (`load_osm`: XML, transverse Mercator in complex arithmetic, bound alignment, outline polygons, its
own centre lines) so that a comparison with the product shares neither the product's reader nor its centre-line code.

Pinning: the known answers of the reference's own tests for this path (tests/simulator/test_util.py:17-44: direction pi/4 on the
line (0,0)-(1,1)-(2,1); losses [[0,1],[0,1]] for the two-agent scene, all zeros once the lanelet is tagged `parking`) --
tests/test_lanelet2.py.  Everything else about this path is PARITY UNPINNED (no Lanelet2 here to generate vectors with).
"""
import cmath
import math

import numpy as np


def _seg_intersect(p1, p2, q1, q2) -> bool:
    """closed segments p1p2 and q1q2 share a point (2-D)"""
    def orient(a, b, c):
        return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])

    def on(a, b, c):
        return min(a[0], b[0]) <= c[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= c[1] <= max(a[1], b[1])
    o1, o2, o3, o4 = orient(p1, p2, q1), orient(p1, p2, q2), orient(q1, q2, p1), orient(q1, q2, p2)
    if ((o1 > 0) != (o2 > 0)) and ((o3 > 0) != (o4 > 0)) and o1 != 0 and o2 != 0 and o3 != 0 and o4 != 0:
        return True
    return (o1 == 0 and on(p1, p2, q1)) or (o2 == 0 and on(p1, p2, q2)) or (o3 == 0 and on(q1, q2, p1)) or (o4 == 0 and on(q1, q2, p2))


class _BoundChecker:
    """Does a candidate segment cross the left bound, the right bound, the entry or the exit of the lanelet."""

    def __init__(self, left: np.ndarray, right: np.ndarray):
        self.left = [tuple(p) for p in left[:, :2]]
        self.right = [tuple(p) for p in right[:, :2]]
        self.entry = (self.right[0], self.left[0])
        self.exit = (self.left[-1], self.right[-1])

    @staticmethod
    def _crosses(line, seg, skip_first_equal=True) -> bool:
        for a, b in zip(line[:-1], line[1:]):
            if _seg_intersect(seg[0], seg[1], a, b):
                if a != seg[0] and b != seg[0] if skip_first_equal else True:       # segments leaving from seg[0] do not count
                    return True
        return False

    def intersects(self, seg) -> bool:
        """the centre-line candidate (from the last centre point) touches a bound or leaves through the entry / exit"""
        return (self._crosses(self.left, seg) or self._crosses(self.right, seg)
                or self._crosses_gate(self.entry, seg) or self._crosses_gate(self.exit, seg))

    @staticmethod
    def _crosses_gate(gate, seg) -> bool:
        """the inside of the lanelet is on the right of both gates (entry = right.front -> left.front, exit = left.back ->
        right.back): a segment leaves through a gate when its end lies strictly on the outer side and it meets the gate"""
        (ax, ay), (bx, by) = gate
        outside = (bx - ax) * (seg[1][1] - ay) - (by - ay) * (seg[1][0] - ax) > 0
        return outside and _seg_intersect(seg[0], seg[1], gate[0], gate[1])

    def second_crosses_bounds(self, seg, left: bool) -> bool:
        """a connection between the bounds crosses the bound its SECOND point lies on, other than in that point"""
        line = self.left if left else self.right
        for a, b in zip(line[:-1], line[1:]):
            if a != seg[1] and b != seg[1] and _seg_intersect(seg[0], seg[1], a, b):
                return True
        return False


def calculate_centerline(left: np.ndarray, right: np.ndarray) -> np.ndarray:
    """
    Centre line of a lanelet as (k,3) float64.  Starting from the midpoint of the two first bound points, the next point of
    either bound is paired with the current point of the other bound; among the points ahead, the one closest to the other
    bound's current point whose connection stays inside the lanelet is taken, the shorter of the left / right candidates
    wins (left on ties), and the midpoint of the pair is appended.  The midpoint of the two last points always ends the line.
    """
    left = np.asarray(left, np.float64)
    right = np.asarray(right, np.float64)
    if len(left) == 0 or len(right) == 0:
        return np.zeros((0, 3))
    bounds = _BoundChecker(left, right)
    pts = [0.5 * (left[0] + right[0])]
    il, ir = 0, 0

    def closest(line, cur, other_pt, is_left):
        best, best_d = None, None
        last = tuple(pts[-1][:2])
        other2 = tuple(other_pt[:2])
        d_last_other = math.dist(other2, last)
        order = sorted(range(cur + 1, len(line)), key=lambda k: math.dist(tuple(line[k, :2]), other2))
        for k in order:
            cand2 = tuple(line[k, :2])
            d = math.dist(cand2, other2) / 2.0
            if best_d is not None and d - d_last_other > best_d:
                break                                                          # no closer point can follow
            if best_d is not None and best_d <= d:
                continue
            centre = (0.5 * (cand2[0] + other2[0]), 0.5 * (cand2[1] + other2[1]))
            if (not bounds.intersects((last, centre)) and not bounds.second_crosses_bounds((other2, cand2), is_left)
                    and not bounds.second_crosses_bounds((cand2, other2), not is_left)):
                best, best_d = k, d
        return best, best_d

    while il < len(left) - 1 or ir < len(right) - 1:
        kl, dl = closest(left, il, right[ir], True)
        kr, dr = closest(right, ir, left[il], False)
        if dl is not None and (dr is None or dl <= dr):
            pts.append(0.5 * (left[kl] + right[ir]))
            il = kl
        elif dr is not None:
            pts.append(0.5 * (left[il] + right[kr]))
            ir = kr
        else:
            break
    if not (il == len(left) - 1 and ir == len(right) - 1):
        pts.append(0.5 * (left[-1] + right[-1]))
    return np.stack(pts, 0)


def polygon_distance(poly: np.ndarray, x: float, y: float) -> float:
    """boost::geometry::distance(point, polygon): 0 inside (even-odd crossing rule), else the distance to the closest edge"""
    a = poly
    b = np.roll(poly, -1, axis=0)
    inside = False
    for (ax, ay), (bx, by) in zip(a, b):
        if (ay > y) != (by > y) and x < ax + (y - ay) * (bx - ax) / (by - ay):
            inside = not inside
    if inside:
        return 0.0
    d = b - a
    l2 = (d * d).sum(1)
    t = np.clip(((np.array([x, y]) - a) * d).sum(1) / np.where(l2 > 0, l2, 1.0), 0.0, 1.0)
    foot = a + t[:, None] * d
    return float(np.sqrt(((foot - np.array([x, y])) ** 2).sum(1).min()))


def find_within_2d(lanelets, x: float, y: float, max_dist: float):
    """[(distance, lanelet)] sorted by distance; `lanelets`: objects with `.polygon2d()`"""
    found = []
    for l in lanelets:
        d = polygon_distance(l.polygon2d(), x, y)
        if d <= max_dist:
            found.append((d, l))
    found.sort(key=lambda t: t[0])
    return found


class LaneletError(RuntimeError):
    pass


def project(ls: np.ndarray, p: np.ndarray) -> np.ndarray:
    best, out = float('inf'), None
    for a, b in zip(ls[:-1], ls[1:]):
        d = b - a
        l2 = float((d * d).sum())
        t = min(max(float(((p - a) * d).sum()) / l2, 0.0), 1.0) if l2 > 0 else 0.0
        f = a + t * d
        d2 = float(((f - p) ** 2).sum())
        if d2 < best:
            best, out = d2, f
    return out


def find_direction(ls: np.ndarray, location3d: np.ndarray) -> float:
    """lanelet2.py:144-180, line by line"""
    projected = project(ls, location3d)
    first, second = float('inf'), float('inf')
    closest, second_closest = 0, 0
    for i, point in enumerate(ls):
        dist = float(np.sqrt(((projected - point) ** 2).sum()))
        if dist < first:
            second = first
            first = dist
            second_closest = closest
            closest = i
        elif dist < second:
            second = dist
            second_closest = i
    if not abs(closest - second_closest) == 1:
        raise LaneletError('Failed to find direction of the linestring at a given point')
    if closest > second_closest:
        a, b = ls[second_closest], ls[closest]
    else:
        b, a = ls[second_closest], ls[closest]
    return float(np.arctan2(b[1] - a[1], b[0] - a[0]))


def find_lanelet_directions(lanelets, centerlines, x, y, tags_to_exclude=(), tol=1.0):
    """lanelet2.py:108-141; `centerlines`: dict id(lanelet) -> (k,3) array"""
    loc3 = np.array([float(x), float(y), 0.0])
    directions = []
    for _, l in find_within_2d(lanelets, float(x), float(y), tol):
        c = centerlines[id(l)]
        if len(c) < 2:
            continue
        if any(t in l.attributes for t in tags_to_exclude):
            directions = []
            break
        directions.append(find_direction(c, loc3))
    return directions


# ------------------------------------------------------------------------------------------------------------------------
# the oracle's own lane table: OSM file -> lanelets.  Independent of torchdrivesim_amd/lanelet2.py (no import, different code);
# both restate lanelet2_io's OsmHandlerLoad + lanelet2_projection's UtmProjector [UNVERIFIED-UPSTREAM], and the product's reader is
# separately pinned against the mesh the reference ships (tests/test_lanelet2.py).
# ------------------------------------------------------------------------------------------------------------------------
class OracleLanelet:
    """left / right bound (k,3) float64 in travel order + OSM tags"""

    def __init__(self, ident, left, right, attributes=None):
        self.id = ident
        self.left = np.asarray(left, np.float64).reshape(-1, 3)
        self.right = np.asarray(right, np.float64).reshape(-1, 3)
        self.attributes = dict(attributes or {})

    def polygon2d(self):
        """Lanelet::polygon2d: the left bound, then the right bound backwards"""
        return np.concatenate([self.left[:, :2], self.right[::-1, :2]], 0)


class _LazyCenterlines(dict):
    """id(lanelet) -> centre line, computed when first asked for (the pure-Python construction takes 70 ms per lanelet)"""

    def __init__(self, lanelets, fn):
        super().__init__()
        self._by_id, self._fn = {id(l): l for l in lanelets}, fn

    def __missing__(self, key):
        l = self._by_id[key]
        self[key] = self._fn(l.left, l.right)
        return self[key]


class OracleMap:
    def __init__(self, lanelets):
        self.laneletLayer = list(lanelets)
        self._centerlines = None

    def centerlines(self, centerline_fn=None):
        if centerline_fn is not None:
            return _LazyCenterlines(self.laneletLayer, centerline_fn)
        if self._centerlines is None:
            self._centerlines = _LazyCenterlines(self.laneletLayer, calculate_centerline)
        return self._centerlines


def transverse_mercator(lat_deg, lon_deg, lon0_deg):
    """(easting - 500 km, northing) / metres of a WGS84 position on the central meridian lon0: Krueger's series in the complex
    form  zeta = zeta' + sum_j alpha_j sin(2 j zeta'),  zeta' = xi' + i eta',  alpha_j to n^6 (Karney 2011, eq. 35; the series
    GeographicLib::TransverseMercator evaluates)."""
    a, f, k0 = 6378137.0, 1.0 / 298.257223563, 0.9996
    n = f / (2.0 - f)
    e = math.sqrt(f * (2.0 - f))
    A = a / (1.0 + n) * (1.0 + n ** 2 / 4.0 + n ** 4 / 64.0 + n ** 6 / 256.0)
    alpha = (n * (1 / 2 + n * (-2 / 3 + n * (5 / 16 + n * (41 / 180 + n * (-127 / 288 + n * 7891 / 37800))))),
             n ** 2 * (13 / 48 + n * (-3 / 5 + n * (557 / 1440 + n * (281 / 630 - n * 1983433 / 1935360)))),
             n ** 3 * (61 / 240 + n * (-103 / 140 + n * (15061 / 26880 + n * 167603 / 181440))),
             n ** 4 * (49561 / 161280 + n * (-179 / 168 + n * 6601661 / 7257600)),
             n ** 5 * (34729 / 80640 - n * 3418889 / 1995840),
             n ** 6 * 212378941 / 319334400)
    phi, lam = math.radians(lat_deg), math.radians(lon_deg - lon0_deg)
    tau = math.tan(phi)
    sigma = math.sinh(e * math.atanh(e * tau / math.sqrt(1.0 + tau * tau)))
    tau_p = tau * math.sqrt(1.0 + sigma * sigma) - sigma * math.sqrt(1.0 + tau * tau)
    zeta_p = complex(math.atan2(tau_p, math.cos(lam)), math.asinh(math.sin(lam) / math.hypot(tau_p, math.cos(lam))))
    zeta = zeta_p
    for j, al in enumerate(alpha, 1):
        zeta += al * cmath.sin(2 * j * zeta_p)
    return k0 * A * zeta.imag, k0 * A * zeta.real


def _utm_zone(lat, lon):
    lon_i = int(math.floor(lon))
    lon_i = lon_i - 360 if lon_i >= 180 else (lon_i + 360 if lon_i < -180 else lon_i)
    zone = (lon_i + 186) // 6
    band = max(-10, min(9, (int(math.floor(lat)) + 80) // 8 - 10))
    if band == 7 and zone == 31 and lon_i >= 3:
        return 32                                                              # south-west Norway
    if band == 9 and 0 <= lon_i < 42:
        return 2 * ((lon_i + 183) // 12) + 1                                   # Svalbard
    return zone


def _side(line, p):
    """> 0: p on the left of the poly-line, < 0: on its right, judged at the closest segment"""
    best, side = float('inf'), 0.0
    for (ax, ay), (bx, by) in zip(line[:-1, :2], line[1:, :2]):
        dx, dy = bx - ax, by - ay
        l2 = dx * dx + dy * dy
        t = 0.0 if l2 == 0 else min(1.0, max(0.0, ((p[0] - ax) * dx + (p[1] - ay) * dy) / l2))
        d2 = (ax + t * dx - p[0]) ** 2 + (ay + t * dy - p[1]) ** 2
        if d2 < best:
            best, side = d2, dx * (p[1] - ay) - dy * (p[0] - ax)
    return side


def load_osm(path, origin=(0.0, 0.0), align=True):
    """lanelet2.io.load(path, UtmProjector(Origin(*origin))) restricted to what the wrong-way query reads: the lanelets' bounds in the
    UTM zone of the origin, relative to the origin, each bound turned so that the right one lies on the right of the left one
    (OsmHandlerLoad alignLaneletBorders; align=False keeps the file's order)."""
    import gzip
    import xml.etree.ElementTree as ET
    with (gzip.open if path.endswith('.gz') else open)(path, 'rb') as f:
        osm = ET.parse(f).getroot()
    lon0 = 6.0 * _utm_zone(*origin) - 183.0
    e0, n0 = transverse_mercator(origin[0], origin[1], lon0)
    node = {}
    for nd in osm.findall('node'):
        ele = [float(t.get('v')) for t in nd.findall('tag') if t.get('k') == 'ele']
        e, n = transverse_mercator(float(nd.get('lat')), float(nd.get('lon')), lon0)
        node[nd.get('id')] = (e - e0, n - n0, ele[-1] if ele else 0.0)
    way = {w.get('id'): np.array([node[r.get('ref')] for r in w.findall('nd')], np.float64).reshape(-1, 3) for w in osm.findall('way')}
    out = []
    for rel in osm.findall('relation'):
        tags = {t.get('k'): t.get('v') for t in rel.findall('tag')}
        if tags.get('type') != 'lanelet':
            continue
        side = {m.get('role'): way[m.get('ref')] for m in rel.findall('member') if m.get('type') == 'way' and m.get('role') in ('left', 'right')}
        if len(side) != 2:
            continue
        left, right = side['left'], side['right']
        if align and len(left) > 1 and len(right) > 1:
            # the right bound must begin on the right of the left bound, and the left bound on the left of the right bound
            flip_left, flip_right = _side(left, right[0]) > 0, _side(right, left[0]) < 0
            left = left[::-1].copy() if flip_left else left
            right = right[::-1].copy() if flip_right else right
        out.append(OracleLanelet(int(rel.get('id')), left, right, tags))
    return OracleMap(out)


def lanelet_orientation_loss(lanelet_maps, agents_state, recenter_offset=None, thr=math.pi / 2, tol=1.0, centerline_fn=None,
                             tags_to_exclude=('parking',)):
    """infractions.py:232-304.  lanelet_maps: list of B objects with `.laneletLayer` (or None); agents_state (B,A,4) float32.
    `centerline_fn(left, right)`: defaults to this file's calculate_centerline."""
    centerline_fn = centerline_fn or calculate_centerline
    f32 = np.float32
    out = np.zeros(agents_state.shape[:2], f32)
    cache = {}
    for b, m in enumerate(lanelet_maps):
        if m is None:
            continue
        if id(m) not in cache:
            if isinstance(m, OracleMap):
                cache[id(m)] = m.centerlines(None if centerline_fn is calculate_centerline else centerline_fn)
            else:
                cache[id(m)] = {id(l): centerline_fn(l.left, l.right) for l in m.laneletLayer}
        cls = cache[id(m)]
        for a in range(agents_state.shape[1]):
            x, y, psi = f32(agents_state[b, a, 0]), f32(agents_state[b, a, 1]), f32(agents_state[b, a, 2])
            if recenter_offset is not None:
                x = f32(x + f32(recenter_offset[b, 0]))
                y = f32(y + f32(recenter_offset[b, 1]))
            try:
                dirs = find_lanelet_directions(m.laneletLayer, cls, float(x), float(y), tags_to_exclude, tol)
            except LaneletError:
                continue
            if dirs:
                d = np.array(dirs, np.float64).astype(f32) - psi                       # torch.tensor(directions) - agent_psi
                d = np.remainder(d + f32(np.pi), f32(2 * np.pi)) - f32(np.pi)           # utils.normalize_angle, utils.py:31-37
                losses = -np.cos(d) * (np.abs(d) > f32(thr)).astype(f32)
                out[b, a] = losses.min()
    return out
