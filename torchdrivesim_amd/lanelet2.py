"""
Lane maps for the wrong-way query: the part of the reference's `lanelet2.py` that sits on the path
`Simulator.compute_wrong_way -> infractions.lanelet_orientation_loss -> find_lanelet_directions -> find_direction`
(reference simulator.py:607-630, infractions.py:232-304, lanelet2.py:88-180), without the `lanelet2` package.

The reference keeps a `lanelet2.core.LaneletMap` (C++ objects behind boost-python) and queries it agent by agent from a triple
Python loop.  Here a map is read ONCE into flat float64 arrays (this module, host side, numpy) and uploaded as a lane table;
the per-agent query is one HIP kernel launch for the whole batch (`csrc/lanes.hip`, `tds_wrong_way_f32`).

What is restated from the (absent, unpinned -- the reference's CI does `pip install lanelet2`) Lanelet2 library, from its
published sources, all marked [UNVERIFIED-UPSTREAM] because the library cannot be run here:
  * the OSM reader (lanelet2_io OsmFile / OsmHandlerLoad): nodes `lat lon` + optional `ele` tag, ways, relations with
    `type=lanelet` and members of role `left` / `right`; bounds are inverted where needed so that the right bound lies on the
    right of the left bound (`alignLaneletBorders`);
  * `UtmProjector(Origin(lat, lon))`: transverse Mercator (Krueger series to n^6 as in GeographicLib), zone of the origin,
    coordinates relative to the origin's;
  * the centre line of a lanelet (lanelet2_core Lanelet.cpp `calculateCenterline`): greedy pairing of the next point of either
    bound with the current point of the other one, rejecting connections that cross a bound.
PINNED here against data the reference ships: the projected points and the road / lane-marking meshes built from
`carla_Town01.osm` equal the reference's own `carla_Town01_mesh.json` (generated upstream with the real Lanelet2), see
`tests/test_lanelet2.py`; the query is pinned by the known answers of the reference's tests/simulator/test_util.py:17-44.
"""
import gzip
import math
import random
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from .mesh import BaseMesh, BirdviewMesh, rendering_mesh

is_available = True        # the reference exports this flag (lanelet2.py:21-27); this module needs no external package


@dataclass
class LaneFeatures:
    """Lane markers handed to policies next to the image (lanelet2.py:30-69): dense `[B, M, D]` and sparse `[B, N, D]` feature rows with
    their masks.  Pure batch-axis plumbing; the simulator only carries them."""
    dense_lane_features: Optional[Tensor] = None
    dense_lane_features_mask: Optional[Tensor] = None
    sparse_lane_features: Optional[Tensor] = None
    sparse_lane_features_mask: Optional[Tensor] = None

    def _map(self, f) -> 'LaneFeatures':
        return LaneFeatures(*[None if x is None else f(x) for x in (self.dense_lane_features, self.dense_lane_features_mask,
                                                                    self.sparse_lane_features, self.sparse_lane_features_mask)])

    def to(self, device) -> 'LaneFeatures':
        return self._map(lambda x: x.to(device))

    def copy(self) -> 'LaneFeatures':
        return self._map(lambda x: x)

    def extend(self, n: int) -> 'LaneFeatures':
        return self._map(lambda x: x.unsqueeze(1).expand((x.shape[0], n) + x.shape[1:]).reshape((n * x.shape[0],) + x.shape[1:]))

    def select_batch_elements(self, idx) -> 'LaneFeatures':
        return self._map(lambda x: x[idx])


class Lanelet2NotFound(ImportError):
    """Kept for interface compatibility (lanelet2.py:72-76); never raised here."""


class LaneletError(RuntimeError):
    """Some function related to lane maps failed (lanelet2.py:79-83)."""


# ------------------------------------------------------------------------------------------------------------------------
# UTM projection  [UNVERIFIED-UPSTREAM: lanelet2_projection UTM.cpp, GeographicLib TransverseMercator.cpp]
# ------------------------------------------------------------------------------------------------------------------------
_WGS84_A = 6378137.0
_WGS84_F = 1.0 / 298.257223563
_K0 = 0.9996


def _krueger_alpha(n: float) -> Tuple[float, List[float]]:
    n2, n3, n4, n5, n6 = n * n, n ** 3, n ** 4, n ** 5, n ** 6
    b1 = (1.0 + n2 / 4.0 + n4 / 64.0 + n6 / 256.0) / (1.0 + n)
    alp = [
        n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800,
        13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360,
        61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440,
        49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600,
        34729 * n5 / 80640 - 3418889 * n6 / 1995840,
        212378941 * n6 / 319334400,
    ]
    return b1, alp


def _standard_zone(lat: float, lon: float) -> int:
    """UTM zone of a position, with the Norway / Svalbard exceptions (GeographicLib UTMUPS::StandardZone)."""
    ilon = int(math.floor(lon))
    if ilon >= 180:
        ilon -= 360
    elif ilon < -180:
        ilon += 360
    zone = (ilon + 186) // 6
    band = max(-10, min(9, (int(math.floor(lat)) + 80) // 8 - 10))
    if band == 7 and zone == 31 and ilon >= 3:
        zone = 32
    elif band == 9 and 0 <= ilon < 42:
        zone = 2 * ((ilon + 183) // 12) + 1
    return zone


def utm_forward(lat, lon, zone: int):
    """Easting / northing (metres, false easting 500 km, northern-hemisphere northing) of WGS84 positions in a FIXED zone."""
    lat = np.asarray(lat, np.float64)
    lon = np.asarray(lon, np.float64)
    f = _WGS84_F
    e2 = f * (2.0 - f)
    es = math.sqrt(e2)
    n = f / (2.0 - f)
    b1, alp = _krueger_alpha(n)
    a1 = b1 * _WGS84_A
    lon0 = 6.0 * zone - 183.0
    dlon = np.deg2rad(lon - lon0)
    phi = np.deg2rad(lat)
    tau = np.tan(phi)
    sig = np.sinh(es * np.arctanh(es * tau / np.hypot(1.0, tau)))
    taup = np.hypot(1.0, sig) * tau - sig * np.hypot(1.0, tau)
    xip = np.arctan2(taup, np.cos(dlon))
    etap = np.arcsinh(np.sin(dlon) / np.hypot(taup, np.cos(dlon)))
    xi, eta = xip.copy(), etap.copy()
    for j, a in enumerate(alp, start=1):
        xi = xi + a * np.sin(2 * j * xip) * np.cosh(2 * j * etap)
        eta = eta + a * np.cos(2 * j * xip) * np.sinh(2 * j * etap)
    return 500000.0 + _K0 * a1 * eta, _K0 * a1 * xi


class UtmProjector:
    """`lanelet2.projection.UtmProjector(lanelet2.io.Origin(lat, lon))` (reference lanelet2.py:103): positions are
    projected in the UTM zone of the origin and expressed relative to the origin's own projection."""

    def __init__(self, origin: Tuple[float, float] = (0.0, 0.0)):
        self.origin = (float(origin[0]), float(origin[1]))
        self.zone = _standard_zone(*self.origin)
        self.north = self.origin[0] >= 0
        x0, y0 = utm_forward(self.origin[0], self.origin[1], self.zone)
        self.x0, self.y0 = float(x0), float(y0) + (0.0 if self.north else 10000000.0)

    def forward(self, lat, lon):
        x, y = utm_forward(lat, lon, self.zone)
        if not self.north:                         # northing counted in the hemisphere of the origin
            y = y + 10000000.0
        return x - self.x0, y - self.y0


# ------------------------------------------------------------------------------------------------------------------------
# map objects (plain arrays)
# ------------------------------------------------------------------------------------------------------------------------
@dataclass
class Lanelet:
    """One lanelet: left / right bound as (n,3) float64 arrays (travel direction = order of the points), ids of the bound
    points, the OSM tags, and the centre line (computed on first use, like `lanelet.centerline`)."""
    id: int
    left: np.ndarray
    right: np.ndarray
    left_ids: np.ndarray
    right_ids: np.ndarray
    attributes: Dict[str, str] = field(default_factory=dict)
    _centerline: Optional[np.ndarray] = None

    @property
    def leftBound(self) -> np.ndarray:
        return self.left

    @property
    def rightBound(self) -> np.ndarray:
        return self.right

    @property
    def centerline(self) -> np.ndarray:
        if self._centerline is None:
            self._centerline = calculate_centerline(self.left, self.right)
        return self._centerline

    def polygon2d(self) -> np.ndarray:
        """left bound followed by the reversed right bound (lanelet2 `Lanelet::polygon2d`)"""
        return np.concatenate([self.left[:, :2], self.right[::-1, :2]], 0)

    def invert(self) -> 'Lanelet':
        """The same surface driven the other way (`lanelet2.core.Lanelet.invert`): the reversed right bound becomes the left one."""
        return Lanelet(self.id, self.right[::-1].copy(), self.left[::-1].copy(), self.right_ids[::-1].copy(), self.left_ids[::-1].copy(),
                       dict(self.attributes))


class LaneletMap:
    """The two layers of a Lanelet2 map the reference touches: `pointLayer` (ids + coordinates, in file order) and
    `laneletLayer`."""

    def __init__(self, point_ids: Sequence[int], points: np.ndarray, lanelets: List[Lanelet]):
        self.point_ids = np.asarray(point_ids, np.int64)
        self.points = np.asarray(points, np.float64).reshape(-1, 3)
        self.laneletLayer: List[Lanelet] = list(lanelets)
        self._tables = {}                  # device lane tables, one per (device index)

    @property
    def pointLayer(self) -> np.ndarray:
        return self.points

    def add(self, lanelet: Lanelet) -> None:
        self.laneletLayer.append(lanelet)
        self._tables = {}

    def table(self, device, tags_to_exclude: Sequence[str] = (), tolerance: float = 1.0):
        """The device lane table of this map (built on first use per device / tag list; rebuilt for a larger tolerance)."""
        from . import _ops
        device = torch.device(device)
        key = (device.type, device.index, tuple(tags_to_exclude))
        h = self._tables.get(key)
        if h is None or h.max_tolerance < tolerance:
            h = _ops.LaneTableHandle(lane_table(self, tags_to_exclude), device, max_tolerance=max(float(tolerance), 1.0))
            self._tables[key] = h
        return h

    def __bool__(self) -> bool:           # `if not lanelet_map` in the reference (infractions.py:266)
        return True


def revert_map(lanelet_map: 'LaneletMap') -> 'LaneletMap':
    """Every lanelet inverted -- what the reference's examples/lanelet2_to_birdview_mesh.py:20-36 does to CARLA maps after loading
    ("Fixing for Carla left-handed coordinates"): a loader that works in number space turns the lanelets of a map authored in a
    left-handed frame around, and this turns them back."""
    return LaneletMap(lanelet_map.point_ids, lanelet_map.points, [l.invert() for l in lanelet_map.laneletLayer if len(l.left) and len(l.right)])


def make_lanelet(lanelet_id: int, left, right, attributes: Optional[Dict[str, str]] = None) -> Lanelet:
    """A lanelet from two point lists, `lanelet2.core.Lanelet(id, left_bound, right_bound)`; points are (x, y) or (x, y, z)."""
    def pts(p):
        p = np.asarray(p, np.float64)
        return np.concatenate([p, np.zeros((len(p), 1))], 1) if p.shape[1] == 2 else p
    l, r = pts(left), pts(right)
    return Lanelet(int(lanelet_id), l, r, -np.arange(1, len(l) + 1), -np.arange(len(l) + 1, len(l) + len(r) + 1), dict(attributes or {}))


# ------------------------------------------------------------------------------------------------------------------------
# OSM reader  [UNVERIFIED-UPSTREAM: lanelet2_io OsmFile.cpp, OsmHandlerLoad.cpp]
# ------------------------------------------------------------------------------------------------------------------------
def _signed_side(line: np.ndarray, p: np.ndarray) -> float:
    """> 0 when `p` lies on the left of the poly-line `line` (2-D), judged at the closest segment: the sign convention of
    lanelet2 `geometry::signedDistance`."""
    a, b = line[:-1, :2], line[1:, :2]
    d = b - a
    l2 = (d * d).sum(1)
    t = np.clip(((p[:2] - a) * d).sum(1) / np.where(l2 > 0, l2, 1.0), 0.0, 1.0)
    foot = a + t[:, None] * d
    k = int(np.argmin(((foot - p[:2]) ** 2).sum(1)))
    return float(d[k, 0] * (p[1] - a[k, 1]) - d[k, 1] * (p[0] - a[k, 0]))


def _align_borders(left: np.ndarray, lids: np.ndarray, right: np.ndarray, rids: np.ndarray):
    """Invert a bound that runs against the lanelet: the right bound must start on the right of the left bound and the left
    bound on the left of the right bound (OsmHandlerLoad `alignLaneletBorders`)."""
    if len(left) < 2 or len(right) < 2:
        return left, lids, right, rids
    inv_left = _signed_side(left, right[0]) > 0
    inv_right = _signed_side(right, left[0]) < 0
    if inv_left:
        left, lids = left[::-1].copy(), lids[::-1].copy()
    if inv_right:
        right, rids = right[::-1].copy(), rids[::-1].copy()
    return left, lids, right, rids


def load_lanelet_map(map_path: str, origin: Tuple[float, float] = (0, 0), align_borders: bool = True) -> LaneletMap:
    """
    Load a Lanelet2 map from an OSM file on disk (reference lanelet2.py:86-105; `.osm` or `.osm.gz`).

    Args:
        map_path: local path to the OSM file
        origin: latitude and longitude of the origin to use with the UTM projector
        align_borders: orient the bounds like Lanelet2's loader (`geometry::align`: the right bound ends up on the right of the left
            one, which fixes the lanelet's direction).  False keeps the order of the file.  See DESIGN.md, "Wrong-way": the shipped
            CARLA maps store every lanelet against that rule, so the two settings give OPPOSITE lane directions on them.
    Raises:
        FileNotFoundError: if the file does not exist
    """
    import os
    if not os.path.exists(map_path):
        raise FileNotFoundError(map_path)
    opener = gzip.open if map_path.endswith('.gz') else open
    with opener(map_path, 'rb') as f:
        root = ET.parse(f).getroot()
    proj = UtmProjector(origin)
    ids, lat, lon, ele = [], [], [], []
    for nd in root.iter('node'):
        ids.append(int(nd.get('id')))
        lat.append(float(nd.get('lat')))
        lon.append(float(nd.get('lon')))
        z = 0.0
        for tag in nd.iter('tag'):
            if tag.get('k') == 'ele':
                z = float(tag.get('v'))
        ele.append(z)
    x, y = proj.forward(np.array(lat), np.array(lon))
    points = np.stack([x, y, np.array(ele)], 1) if ids else np.zeros((0, 3))
    index = {i: k for k, i in enumerate(ids)}
    ways = {}
    for w in root.iter('way'):
        refs = [int(nd.get('ref')) for nd in w.iter('nd')]
        ways[int(w.get('id'))] = np.array(refs, np.int64)
    lanelets = []
    for rel in root.iter('relation'):
        tags = {t.get('k'): t.get('v') for t in rel.iter('tag')}
        if tags.get('type') != 'lanelet':
            continue
        sides = {}
        for m in rel.iter('member'):
            if m.get('type') == 'way' and m.get('role') in ('left', 'right'):
                sides[m.get('role')] = ways[int(m.get('ref'))]
        if 'left' not in sides or 'right' not in sides:
            continue
        lids, rids = sides['left'], sides['right']
        left = points[[index[i] for i in lids]]
        right = points[[index[i] for i in rids]]
        if align_borders:
            left, lids, right, rids = _align_borders(left, lids, right, rids)
        attrs = {k: v for k, v in tags.items() if k != 'type'}
        attrs['type'] = 'lanelet'
        lanelets.append(Lanelet(int(rel.get('id')), left, right, lids, rids, attrs))
    return LaneletMap(ids, points, lanelets)


# ------------------------------------------------------------------------------------------------------------------------
# centre line  [UNVERIFIED-UPSTREAM: lanelet2_core/src/Lanelet.cpp calculateCenterline / BoundChecker]
# ------------------------------------------------------------------------------------------------------------------------
def calculate_centerline(left: np.ndarray, right: np.ndarray) -> np.ndarray:
    """
    Centre line of a lanelet as (k,3) float64, computed by the native library (`tds_lanelet_centerline_f64`, csrc/lanes.hip; host
    C++ like Lanelet2's own).  Starting from the midpoint of the two first bound points, the next point of either bound is
    paired with the current point of the other bound; among the points ahead, the one closest to the other bound's current point
    whose connection stays inside the lanelet is taken, the shorter of the left / right candidates wins (left on ties), and the
    midpoint of the pair is appended.  The midpoint of the two last points always ends the line.
    """
    import ctypes
    from . import _native as nat
    left = np.ascontiguousarray(left, np.float64).reshape(-1, 3)
    right = np.ascontiguousarray(right, np.float64).reshape(-1, 3)
    out = np.zeros((len(left) + len(right) + 1, 3), np.float64)
    n = ctypes.c_int(0)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    nat.check(nat.lib().tds_lanelet_centerline_f64(vp(left), len(left), vp(right), len(right), vp(out), ctypes.byref(n)), 'tds_lanelet_centerline_f64')
    return out[:n.value].copy()


# ------------------------------------------------------------------------------------------------------------------------
# meshes built from a lane map (reference lanelet2.py:211-377), host side, once per map
# ------------------------------------------------------------------------------------------------------------------------
def road_mesh_from_lanelet_map(lanelet_map: LaneletMap, lanelets: Optional[List[int]] = None) -> BaseMesh:
    """
    Creates a road mesh by triangulating all lanelets in a given map (reference lanelet2.py:211-256): every point of the map
    is a vertex; a lanelet with bounds of n and m points gives n + m - 2 faces zipped along the two bounds.
    """
    verts = torch.from_numpy(lanelet_map.points[:, :2].astype(np.float32))
    index = {int(i): k for k, i in enumerate(lanelet_map.point_ids)}
    out = []
    for l in lanelet_map.laneletLayer:
        if lanelets is not None and l.id not in lanelets:
            continue
        lb = [index[int(i)] for i in l.left_ids]
        rb = [index[int(i)] for i in l.right_ids]
        n_faces = len(lb) + len(rb) - 2
        if n_faces < 1:
            continue
        faces = np.zeros((n_faces, 3), np.int64)
        i = j = 0
        while i + j < n_faces:
            if i < len(lb) - 1:
                faces[i + j] = [lb[i], rb[j], lb[i + 1]]
                i += 1
            if j < len(rb) - 1:
                faces[i + j] = [lb[i], rb[j], rb[j + 1]]
                j += 1
        out.append(torch.from_numpy(faces))
    faces = torch.cat(out, 0) if out else torch.zeros((0, 3), dtype=torch.int64)
    return BaseMesh(verts=verts.unsqueeze(0), faces=faces.unsqueeze(0))


def line_segments_to_mesh(points: Tensor, line_width: float = 0.3, eps: float = 1e-6) -> BaseMesh:
    """
    `BxNx2x2` line segments as strips of 6 vertices / 4 faces each, `line_width` metres to either side
    (reference lanelet2.py:259-287).
    """
    batch_size, n = points.shape[0], points.shape[1]
    d = points[:, :, 1] - points[:, :, 0]
    d_hat = d / (torch.norm(d, p=2, dim=2, keepdim=True) + eps)
    d_perp = torch.stack([-d_hat[:, :, 1], d_hat[:, :, 0]], dim=2).unsqueeze(2)
    verts = torch.cat([points + d_perp * line_width, points, points - d_perp * line_width], dim=2).reshape(batch_size, -1, 2)
    strip = torch.tensor([[0, 1, 2], [1, 2, 3], [2, 3, 4], [3, 4, 5]], dtype=torch.int32, device=points.device)
    faces = (strip[None] + 6 * torch.arange(n, dtype=torch.int32, device=points.device)[:, None, None]).reshape(1, -1, 3)
    return BaseMesh(verts=verts, faces=faces.expand(batch_size, -1, -1).contiguous())


def lanelet_map_to_lane_mesh(lanelet_map: LaneletMap, left_handed: bool = False, batch_size: int = 50000,
                             left_right_marking_join_threshold: float = 0.1, lanelets: Optional[List[int]] = None,
                             lane_boundary_width: float = 0.275) -> BirdviewMesh:
    """
    Lane-marking mesh of a map (reference lanelet2.py:290-377): the distinct segments of all left bounds and of all right
    bounds; a left segment whose two ends lie within `left_right_marking_join_threshold` of the ends of some right segment
    is a `joint_lane` marking, the rest are `left_lane` / `right_lane` strips (swapped for left-handed maps).
    """
    import scipy.spatial
    verts = lanelet_map.points[:, :2].astype(np.float32)
    index = {int(i): k for k, i in enumerate(lanelet_map.point_ids)}
    left_set, right_set = {}, {}
    for l in lanelet_map.laneletLayer:
        if lanelets is not None and l.id not in lanelets:
            continue
        for ids, dst in ((l.right_ids, right_set), (l.left_ids, left_set)):
            for a, b in zip(ids[:-1], ids[1:]):
                dst[tuple(sorted((int(a), int(b))))] = None          # a set that keeps insertion order
    def seg_points(segs):
        return np.stack([np.stack([verts[index[a]], verts[index[b]]]) for a, b in segs], 0) if segs else np.zeros((0, 2, 2), np.float32)
    lp, rp = seg_points(list(left_set)), seg_points(list(right_set))

    def near(a, b):
        return scipy.spatial.distance.cdist(a, b) < left_right_marking_join_threshold
    joint = np.zeros((len(lp), len(rp)), bool)
    for i in range(0, len(lp), batch_size):
        for j in range(0, len(rp), batch_size):
            a, b = lp[i:i + batch_size], rp[j:j + batch_size]
            joint[i:i + batch_size, j:j + batch_size] = (near(a[:, 0], b[:, 0]) & near(a[:, 1], b[:, 1])) | (near(a[:, 0], b[:, 1]) & near(a[:, 1], b[:, 0]))
    left_common, right_common = joint.any(1), joint.any(0)
    left_points, right_points, joint_points = lp[~left_common], rp[~right_common], lp[left_common]
    if left_handed:
        left_points, right_points = right_points, left_points
    as_t = lambda p: torch.tensor(p).unsqueeze(0)
    if joint_points.shape[0] > 0:
        joint_mesh = rendering_mesh(line_segments_to_mesh(as_t(joint_points), line_width=lane_boundary_width), category='joint_lane')
    else:
        joint_mesh = BirdviewMesh.empty(dim=2, batch_size=1)
    left_mesh = rendering_mesh(line_segments_to_mesh(as_t(left_points), line_width=lane_boundary_width), category='left_lane')
    right_mesh = rendering_mesh(line_segments_to_mesh(as_t(right_points), line_width=lane_boundary_width), category='right_lane')
    return BirdviewMesh.concat([joint_mesh, left_mesh, right_mesh])


# ------------------------------------------------------------------------------------------------------------------------
# flat lane table (what the kernel and the oracle read)
# ------------------------------------------------------------------------------------------------------------------------
@dataclass
class LaneTable:
    """All lanelets of a map as flat arrays.
    poly_xy (P,2) f64 : outlines (left bound, then the right bound reversed), lanelet after lanelet
    poly_start (L+1) i32
    cl_xyz (C,3) f64 : centre lines;  cl_start (L+1) i32
    flags (L) i32 : bit 0 = carries an excluded tag
    """
    poly_xy: np.ndarray
    poly_start: np.ndarray
    cl_xyz: np.ndarray
    cl_start: np.ndarray
    flags: np.ndarray


def lane_table(lanelet_map: LaneletMap, tags_to_exclude: Optional[Sequence[str]] = None) -> LaneTable:
    tags = list(tags_to_exclude or [])
    polys, cls, ps, cs, flags = [], [], [0], [0], []
    for l in lanelet_map.laneletLayer:
        p, c = l.polygon2d(), l.centerline
        polys.append(p)
        cls.append(c)
        ps.append(ps[-1] + len(p))
        cs.append(cs[-1] + len(c))
        flags.append(1 if any(t in l.attributes for t in tags) else 0)
    cat = lambda xs, w: np.ascontiguousarray(np.concatenate(xs, 0), np.float64) if xs else np.zeros((0, w))
    return LaneTable(cat(polys, 2), np.array(ps, np.int32), cat(cls, 3), np.array(cs, np.int32), np.array(flags, np.int32))


# ------------------------------------------------------------------------------------------------------------------------
# point queries with the reference's signatures; both run the batch kernel on one point
# ------------------------------------------------------------------------------------------------------------------------
def _query_device(device):
    """where a point query runs: the given device, else the current MI355X -- there is no CPU implementation"""
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        raise RuntimeError('lane-map queries run on an MI355X; no GPU is visible and there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def find_lanelet_directions(lanelet_map: LaneletMap, x: float, y: float, tags_to_exclude: Optional[List[str]] = None,
                            lanelet_dist_tolerance: float = 1.0, device=None) -> List[float]:
    """
    For a given point, find local orientations of all lanelets within `lanelet_dist_tolerance` of it, nearest lanelet first
    (reference lanelet2.py:108-141).  Raises LaneletError where the reference's `find_direction` does.
    """
    from . import _ops
    dev = _query_device(device)
    lanes = lanelet_map.table(dev, tags_to_exclude or [], lanelet_dist_tolerance)
    pts = torch.tensor([[float(x), float(y)]], dtype=torch.float64, device=dev)
    dirs, dists, count, status = _ops.lanelet_directions([lanes], None, pts, float(lanelet_dist_tolerance))
    if int(status[0]) & 2:
        return []                                             # a lanelet with an excluded tag: lanelet2.py:133-135
    if int(status[0]) & 1:
        raise LaneletError('Failed to find direction of the linestring at a given point')
    k = int(count[0])
    if k > dirs.shape[1]:
        raise LaneletError(f'{k} lanelets within tolerance, more than the {dirs.shape[1]} the query returns')
    order = torch.argsort(dists[0, :k], stable=True)          # findWithin2d sorts by distance
    return [float(v) for v in dirs[0, :k][order].cpu()]


def find_direction(linestring, location3d, device=None) -> float:
    """
    Local orientation of a line string next to a point (reference lanelet2.py:144-180): the point is projected onto the line,
    the two vertices closest to the projection must be neighbours (else LaneletError), and the direction runs from the earlier
    to the later one.  Runs the same kernel as the batch query on a one-lanelet table whose outline contains the point.
    """
    from . import _ops
    ls = np.asarray(linestring, np.float64)
    if ls.shape[1] == 2:
        ls = np.concatenate([ls, np.zeros((len(ls), 1))], 1)
    loc = np.zeros(3)
    loc[:len(location3d)] = np.asarray(location3d, np.float64)
    ls = ls - np.array([0.0, 0.0, loc[2]])                     # the kernel's query point has z = 0
    lo = np.minimum(ls[:, :2].min(0), loc[:2]) - 1.0
    hi = np.maximum(ls[:, :2].max(0), loc[:2]) + 1.0
    box = np.array([[lo[0], lo[1]], [hi[0], lo[1]], [hi[0], hi[1]], [lo[0], hi[1]]])
    table = LaneTable(box, np.array([0, 4], np.int32), np.ascontiguousarray(ls), np.array([0, len(ls)], np.int32), np.zeros(1, np.int32))
    dev = _query_device(device)
    lanes = _ops.LaneTableHandle(table, dev, max_tolerance=0.0)
    dirs, _, count, status = _ops.lanelet_directions([lanes], None, torch.tensor([[loc[0], loc[1]]], dtype=torch.float64, device=dev), 0.0)
    if int(status[0]) & 1 or int(count[0]) != 1:
        raise LaneletError('Failed to find direction of the linestring at a given point')
    return float(dirs[0, 0])


def pick_random_point_and_orientation(lanelet_map: LaneletMap) -> Tuple[float, float, float]:
    """A random point on a random centre line and the local orientation there (reference lanelet2.py:183-208)."""
    lanelet = random.choice(list(lanelet_map.laneletLayer))
    c = lanelet.centerline
    seg = np.sqrt(((c[1:] - c[:-1]) ** 2).sum(1))
    cum = np.concatenate([[0.0], np.cumsum(seg)])

    def at(s):
        k = int(np.clip(np.searchsorted(cum, s, side='right') - 1, 0, len(seg) - 1))
        t = (s - cum[k]) / seg[k] if seg[k] > 0 else 0.0
        return c[k] + t * (c[k + 1] - c[k])
    s = random.uniform(0, cum[-1])
    p, q = at(s), at(min(s + 1, cum[-1]))
    return float(p[0]), float(p[1]), float(np.arctan2(q[1] - p[1], q[0] - p[0]))
