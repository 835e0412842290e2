"""
Triangle-mesh containers with the reference's names and semantics (torchdrivesim/mesh.py): `BaseMesh` ->
`AttributeMesh` -> `RGBMesh`, `BirdviewMesh` (per-vertex categories, colours and rendering levels resolved late) and
`BirdviewRGBMeshGenerator`.  They are host-side plumbing around the hot path: the MI355X renderer consumes a
`BirdviewMesh` once (to build the device-resident static map) and generates the actor mesh inside the raster kernel;
`BirdviewRGBMeshGenerator.generate` exists for the reference's generic `render_frame(rgb_mesh, ...)` dataflow.
On-disk formats (`*_mesh.json`, pickles) follow mesh.py:259-297,700-719 so existing map files load unchanged.
"""
import copy
import dataclasses
import json
import os
import pickle
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.utils.rnn import pad_sequence

from torchdrivesim_amd.utils import is_inside_polygon, rotate, transform

Color = Union[Tensor, Tuple[int, int, int]]


def tensor_color(color: Color, device='cpu', dtype=torch.float) -> Tensor:
    """int 3-tuple in [0,255] or (3,) tensor in [0,1] -> (3,) tensor in [0,1] (mesh.py:32-47)"""
    if not isinstance(color, Tensor):
        color = torch.tensor(color, device=device, dtype=dtype) / 255.0
    return color


class BadMeshFormat(RuntimeError):
    pass


def _expand(x: Tensor, batch: int, size: int) -> Tensor:
    return x.unsqueeze(1).expand((batch, size, *x.shape[1:])).flatten(0, 1)


@dataclass
class BaseMesh:
    """Triangles in a Dim-dimensional space with exactly one batch dimension (mesh.py:57-369)."""
    verts: Tensor      #: BxVxDim
    faces: Tensor      #: BxFx3 indices into verts
    _verts_fill: float = dataclasses.field(default=0.0, init=False)
    _faces_fill: int = dataclasses.field(default=0, init=False)      # padded faces are [0,0,0]

    _tensor_fields = ('verts', 'faces')

    @property
    def dim(self) -> int:
        return self.verts.shape[-1]

    @property
    def batch_size(self) -> int:
        return max(self.verts.shape[0], self.faces.shape[0])

    @property
    def verts_count(self) -> int:
        return self.verts.shape[-2]

    @property
    def faces_count(self) -> int:
        return self.faces.shape[-2]

    @property
    def device(self) -> torch.device:
        return self.verts.device

    @property
    def center(self) -> Tensor:
        if self.verts_count > 0:
            return (self.verts.max(dim=-2).values + self.verts.min(dim=-2).values) / 2
        return torch.zeros((self.batch_size, 2), dtype=self.verts.dtype, device=self.device)

    def _map(self, f):
        return dataclasses.replace(self, **{k: f(getattr(self, k)) for k in self._tensor_fields})

    def to(self, device):
        return self._map(lambda x: x.to(device))

    def clone(self):
        return copy.deepcopy(self)

    def expand(self, size: int):
        b = self.batch_size
        return self._map(lambda x: _expand(x, b, size))

    def select_batch_elements(self, idx):
        return self._map(lambda x: x[idx])

    def __getitem__(self, item):
        return self.select_batch_elements(item)

    def pad(self, pad_size: int):
        return self._map(lambda x: torch.cat([x, torch.zeros((pad_size, *x.shape[1:]), device=x.device, dtype=x.dtype)], dim=0))

    def translate(self, xy: Tensor, inplace: bool = True):
        mesh = self if inplace else self.clone()
        mesh.verts = mesh.verts.clone()
        mesh.verts[..., :2] += xy.unsqueeze(1)
        return mesh

    def offset(self, offset: Tensor):
        if offset.shape[-1] < self.dim:
            offset = torch.cat([offset, torch.zeros(offset.shape[:-1] + (self.dim - offset.shape[-1],), dtype=offset.dtype, device=offset.device)])
        return dataclasses.replace(self, verts=self.verts + offset)

    @classmethod
    def collate(cls, meshes):
        verts = pad_sequence([m.verts.squeeze(0) for m in meshes], batch_first=True, padding_value=cls._verts_fill)
        faces = pad_sequence([m.faces.squeeze(0) for m in meshes], batch_first=True, padding_value=cls._faces_fill)
        return cls(verts=verts, faces=faces)

    @classmethod
    def concat(cls, meshes):
        verts = torch.cat([m.verts for m in meshes], dim=-2)
        offsets = np.concatenate([[0], np.cumsum([m.verts_count for m in meshes])[:-1]]).astype(int) if meshes else []
        faces = torch.cat([m.faces + int(o) for m, o in zip(meshes, offsets)], dim=-2)
        return cls(verts=verts, faces=faces)

    def merge(self, other):
        return self.concat([self, other])

    # ---- persistence (mesh.py:238-297)
    def pickle(self, mesh_file_path: str):
        os.makedirs(os.path.dirname(mesh_file_path) or '.', exist_ok=True)
        with open(mesh_file_path, 'wb') as f:
            pickle.dump(self, f)

    @classmethod
    def unpickle(cls, mesh_file_path: str, pickle_module: Any = pickle):
        with open(mesh_file_path, 'rb') as f:
            mesh = pickle_module.Unpickler(f).load()
        if isinstance(mesh, cls._unpickle_type()):
            return mesh
        raise BadMeshFormat

    @classmethod
    def _unpickle_type(cls):
        return BaseMesh

    def serialize(self):
        return {'verts': self.verts.tolist(), 'faces': self.faces.tolist()}

    def save(self, file_save_path: str):
        directory = os.path.dirname(file_save_path)
        if directory:
            os.makedirs(directory, exist_ok=True)
        with open(file_save_path, 'w') as file:
            json.dump(self.serialize(), file)

    @classmethod
    def _deserialize_tensors(cls, data: Dict) -> Dict:
        out = dict(data)
        out.update(verts=torch.tensor(data['verts']), faces=torch.tensor(data['faces']))
        return out

    @classmethod
    def deserialize(cls, data: Dict):
        return cls(**cls._deserialize_tensors(data))

    @classmethod
    def load(cls, filepath):
        try:
            with open(filepath, 'r') as file:
                data = json.load(file)
            return cls.deserialize(data)
        except Exception as e:
            raise BadMeshFormat(str(e))

    @classmethod
    def empty(cls, dim: int = 2, batch_size: int = 1):
        return cls(verts=torch.zeros((batch_size, 0, dim), dtype=torch.float), faces=torch.zeros((batch_size, 0, 3), dtype=torch.int))

    # ---- trimming (mesh.py:308-369): drop faces with all 3 vertices outside, renumber, pad with [0,0,0]
    def _trim_and_return_verts_and_faces(self, vertices_to_keep: Tensor, trim_face_only=False):
        assert vertices_to_keep.dim() == 2, 'Batch dimension needed for polygon tensor.'
        verts, faces = self.verts, self.faces.long()
        keep_face = torch.gather(vertices_to_keep.long().unsqueeze(1).expand(-1, faces.shape[1], -1), index=faces, dim=-1).any(dim=-1)
        new_faces = pad_sequence([faces[i, k] for i, k in enumerate(keep_face)], batch_first=True)
        if trim_face_only:
            return verts, new_faces, None
        if new_faces.shape[1] < 1:
            used = torch.zeros(new_faces.shape[0], 0, device=new_faces.device, dtype=torch.long)
        else:
            used = pad_sequence([x.unique() for x in new_faces.flatten(start_dim=1)], batch_first=True)
        new_verts = torch.gather(verts, 1, used.unsqueeze(-1).expand(-1, -1, verts.shape[-1]))
        remap = torch.zeros(verts.shape[:2], dtype=torch.long, device=verts.device)
        remap.scatter_(1, used, torch.arange(used.shape[1], device=used.device).unsqueeze(0).expand(remap.shape[0], -1))
        new_faces = torch.gather(remap.unsqueeze(-1).expand(-1, -1, 3), dim=1, index=new_faces)
        return new_verts, new_faces, used

    def _trim_extra(self, used: Optional[Tensor]) -> Dict[str, Tensor]:
        return {}

    def trim(self, polygon: Tensor, trim_face_only: bool = False):
        """Crop to a convex BxPx2 polygon: faces with every vertex outside are removed even if they cross it."""
        if self.verts.shape[-1] < 2:
            raise NotImplementedError
        inside = is_inside_polygon(self.verts[..., :2], polygon)
        v, f, used = self._trim_and_return_verts_and_faces(inside, trim_face_only)
        return dataclasses.replace(self, verts=v, faces=f, **self._trim_extra(used))


@dataclass
class AttributeMesh(BaseMesh):
    """Every vertex carries an attribute vector (mesh.py:372-521)."""
    attrs: Tensor = None      #: BxVxAttr
    _attrs_fill: float = dataclasses.field(default=0.0, init=False)

    _tensor_fields = ('verts', 'faces', 'attrs')

    @property
    def attr_dim(self) -> int:
        return self.attrs.shape[-1]

    @classmethod
    def set_attr(cls, mesh: BaseMesh, attr: Tensor):
        assert attr.dim() == 1
        return cls(verts=mesh.verts, faces=mesh.faces, attrs=attr.expand(mesh.verts.shape[:-1] + attr.shape))

    @classmethod
    def concat(cls, meshes):
        base = BaseMesh.concat(meshes)
        return cls(verts=base.verts, faces=base.faces, attrs=torch.cat([m.attrs for m in meshes], dim=-2))

    @classmethod
    def collate(cls, meshes):
        base = BaseMesh.collate(meshes)
        attrs = pad_sequence([m.attrs.squeeze(0) for m in meshes], batch_first=True, padding_value=cls._attrs_fill)
        return cls(verts=base.verts, faces=base.faces, attrs=attrs)

    @classmethod
    def _unpickle_type(cls):
        return AttributeMesh

    def serialize(self):
        data = super().serialize()
        data['attrs'] = self.attrs.tolist()
        return data

    @classmethod
    def _deserialize_tensors(cls, data: Dict) -> Dict:
        out = super()._deserialize_tensors(data)
        out['attrs'] = torch.tensor(data['attrs'])
        return out

    @classmethod
    def empty(cls, dim=2, batch_size=1, attr_dim=3):
        return cls(verts=torch.zeros((batch_size, 0, dim), dtype=torch.float), faces=torch.zeros((batch_size, 0, 3), dtype=torch.int),
                   attrs=torch.zeros((batch_size, 0, attr_dim), dtype=torch.float))

    def _trim_extra(self, used):
        if used is None:
            return {}
        return dict(attrs=torch.gather(self.attrs, 1, used.unsqueeze(-1).expand(-1, -1, self.attrs.shape[-1])))


class RGBMesh(AttributeMesh):
    """AttributeMesh whose attribute is an RGB colour in [0,1] (mesh.py:524-538)."""

    @classmethod
    def set_color(cls, mesh: BaseMesh, color: Color):
        return cls.set_attr(mesh=mesh, attr=tensor_color(color, device=mesh.device, dtype=mesh.verts.dtype))


@dataclass
class BirdviewMesh(BaseMesh):
    """2-D mesh whose vertices belong to named categories; each category has a colour and a rendering level z
    (lower renders on top), possibly supplied later (mesh.py:541-758)."""
    categories: List[str] = None
    colors: Dict[str, Tensor] = None
    zs: Dict[str, float] = None
    vert_category: Tensor = None      #: BxV indices into categories
    _cat_fill: int = 0

    _tensor_fields = ('verts', 'faces', 'vert_category')

    @property
    def num_categories(self) -> int:
        return len(self.categories)

    @classmethod
    def set_properties(cls, mesh: BaseMesh, category: str, color: Optional[Color] = None, z: Optional[float] = None):
        cat = torch.zeros((mesh.batch_size, mesh.verts_count), dtype=mesh.faces.dtype, device=mesh.device)
        return cls(verts=mesh.verts, faces=mesh.faces, categories=[category], vert_category=cat,
                   colors={category: tensor_color(color)} if color is not None else {}, zs={category: z} if z is not None else {})

    @classmethod
    def unify(cls, meshes):
        """Re-index categories so that all meshes share one category list."""
        device = meshes[0].device if meshes else 'cpu'
        categories = []
        for m in meshes:
            categories += [c for c in m.categories if c not in categories]
        colors, zs = {}, {}
        for m in reversed(meshes):      # earlier meshes win
            colors.update(m.colors)
            zs.update(m.zs)
        out = []
        for m in meshes:
            lut = torch.tensor([categories.index(c) for c in m.categories], dtype=torch.int, device=device)
            out.append(dataclasses.replace(m, categories=categories, vert_category=lut[m.vert_category.to(torch.int64)], colors=colors, zs=zs))
        return out

    @classmethod
    def concat(cls, meshes):
        meshes = cls.unify(meshes)
        base = BaseMesh.concat(meshes)
        first = meshes[0] if meshes else None
        return cls(verts=base.verts, faces=base.faces, categories=first.categories if first else [],
                   vert_category=torch.cat([m.vert_category.to(torch.int64) for m in meshes], dim=-1),
                   colors=first.colors if first else {}, zs=first.zs if first else {})

    @classmethod
    def collate(cls, meshes):
        meshes = cls.unify(meshes)
        base = BaseMesh.collate(meshes)
        first = meshes[0] if meshes else None
        cat = pad_sequence([m.vert_category.squeeze(0).to(torch.int64) for m in meshes], batch_first=True, padding_value=cls._cat_fill)
        return cls(verts=base.verts, faces=base.faces, categories=first.categories if first else [], vert_category=cat,
                   colors=first.colors if first else {}, zs=first.zs if first else {})

    def fill_attr(self) -> RGBMesh:
        """Explicit per-vertex colour, and z = rendering level appended to the vertices (mesh.py:663-683)."""
        missing = [c for c in self.categories if c not in self.colors]
        if missing:
            raise RuntimeError(f'Missing color values for the following categories: {missing}')
        missing = [c for c in self.categories if c not in self.zs]
        if missing:
            raise RuntimeError(f'Missing z values for the following categories: {missing}')
        cat = self.vert_category.to(torch.int64)
        zs = torch.tensor([self.zs[k] for k in self.categories], dtype=self.verts.dtype, device=self.device)[cat].unsqueeze(-1)
        if self.categories:
            colors = torch.stack([self.colors[k] for k in self.categories]).to(self.verts.dtype).to(self.device)[cat]
        else:
            colors = torch.zeros((self.batch_size, 0, 3), dtype=self.verts.dtype, device=self.device)
        return RGBMesh(verts=torch.cat([self.verts[..., :2], zs], dim=-1), faces=self.faces, attrs=colors)

    @classmethod
    def _unpickle_type(cls):
        return BirdviewMesh

    def serialize(self):
        data = super().serialize()
        data.update(categories=self.categories, colors={k: v.tolist() for k, v in self.colors.items()}, zs=self.zs,
                    vert_category=self.vert_category.tolist(), _cat_fill=self._cat_fill)
        return data

    @classmethod
    def _deserialize_tensors(cls, data: Dict) -> Dict:
        out = super()._deserialize_tensors(data)
        out.update(categories=data['categories'], colors={k: torch.tensor(v) for k, v in data['colors'].items()}, zs=data['zs'],
                   vert_category=torch.tensor(data['vert_category']), _cat_fill=data['_cat_fill'])
        return out

    @classmethod
    def empty(cls, dim=2, batch_size=1):
        return cls(verts=torch.zeros((batch_size, 0, dim), dtype=torch.float), faces=torch.zeros((batch_size, 0, 3), dtype=torch.int),
                   vert_category=torch.zeros([batch_size, 0], dtype=torch.int), categories=[], colors=dict(), zs=dict())

    def _trim_extra(self, used):
        if used is None:
            return {}
        return dict(vert_category=torch.gather(self.vert_category, 1, used))

    def separate_by_category(self) -> Dict[str, BaseMesh]:
        out = {}
        for i, category in enumerate(self.categories):
            v, f, _ = self._trim_and_return_verts_and_faces(self.vert_category == i, trim_face_only=False)
            out[category] = BaseMesh(verts=v, faces=f)
        return out


def rendering_mesh(mesh: BaseMesh, category: str) -> BirdviewMesh:
    return BirdviewMesh.set_properties(BaseMesh(verts=mesh.verts, faces=mesh.faces), category=category)


def set_colors_with_defaults(mesh: BirdviewMesh, color_map: Dict[str, Tuple[int, int, int]], rendering_levels: Dict[str, float]) -> RGBMesh:
    """Fill in missing category colours / levels from the renderer's tables, then `fill_attr` (mesh.py:1170-1178)."""
    for k in mesh.categories:
        if k not in mesh.colors:
            mesh.colors[k] = tensor_color(color_map[k])
        if k not in mesh.zs:
            mesh.zs[k] = rendering_levels[k]
    return mesh.fill_attr()


def build_verts_faces_from_bounding_box(bbs: Tensor) -> Tuple[Tensor, Tensor]:
    """...xAx4x2 corners -> vertices ...x4Ax2 and faces ...x2Ax3, every box cut along the corner 1 - corner 3 diagonal (mesh.py:1274-1290)"""
    batch, n = bbs.shape[:-3], bbs.shape[-3]
    quad = torch.tensor([[0, 1, 3], [1, 3, 2]], dtype=torch.long, device=bbs.device)
    faces = quad + 4 * torch.arange(n, dtype=torch.long, device=bbs.device).reshape(n, 1, 1)
    return bbs.reshape(*batch, -1, 2), faces.reshape(2 * n, 3).expand(*batch, 2 * n, 3)


#: template faces of one actor: body quad as two triangles, then the direction triangle (mesh.py:955, 933-935)
ACTOR_FACES = ((0, 1, 3), (1, 3, 2), (4, 5, 6))
ACTOR_VERTS = 7


def actor_template(lenwid: Tensor, direction_size: float = 0.3) -> Tensor:
    """(...,2) length/width -> (...,7,2) actor vertices in the agent frame: body corners (l,w),(l,-w),(-l,-w),(-l,w) x 0.5
    and the direction triangle (tip at +l/2, base at l*(0.5-size)) -- mesh.py:911-996 with render_agent_direction=True."""
    length, width = lenwid[..., 0], lenwid[..., 1]
    zero = torch.zeros_like(length)
    body = torch.stack([torch.stack([x, y], dim=-1) for x, y in ((length, width), (length, -width), (-length, -width), (-length, width))], dim=-2) * 0.5
    off = length * (0.5 - direction_size)
    tri = torch.stack([torch.stack([length * direction_size + off, zero + zero], dim=-1),
                       torch.stack([zero + off, width * 0.5 + zero], dim=-1),
                       torch.stack([zero + off, -width * 0.5 + zero], dim=-1)], dim=-2)
    return torch.cat([body, tri], dim=-2)


def generate_disc_mesh(radius: float = 2, num_triangles: int = 10, device='cpu') -> Tuple[Tensor, Tensor]:
    """A fan of `num_triangles` triangles around the origin: verts (T+1) x 2 = centre, then the rim, every rim point being the
    previous one rotated by 360/T degrees (the rounding accumulates exactly as in mesh.py:1243-1271); faces T x 3."""
    T = int(num_triangles)
    step = torch.deg2rad(torch.tensor([[360 / T]], dtype=torch.float32, device=device))
    rim = [torch.tensor([[radius, 0]], dtype=torch.float32, device=device)]
    for _ in range(max(T, 2) - 1):
        rim.append(rotate(rim[-1], step))
    verts = torch.cat([torch.zeros(1, 2, dtype=torch.float32, device=device)] + rim, dim=0)
    k = torch.arange(1, T + 1, dtype=torch.long, device=device)
    nxt = k % T + 1 if T > 1 else k + 1
    return verts, torch.stack([torch.zeros_like(k), k, nxt], dim=-1)


class BirdviewRGBMeshGenerator:
    """Keeps the static background and the per-agent templates and produces, on request, the explicit per-camera RGB
    mesh the reference's renderers consume (mesh.py:761-1157).  The fused MI355X path does NOT call `generate`: it
    reads `background_mesh` once and the templates every step."""

    def __init__(self, background_mesh: BirdviewMesh, color_map: Dict[str, Tuple[int, int, int]], rendering_levels: Dict[str, float],
                 world_center: Optional[Tensor] = None, agent_attributes: Optional[Tensor] = None, agent_types: Optional[Tensor] = None,
                 agent_type_names: Optional[List[str]] = None, render_agent_direction: bool = True, traffic_controls=None,
                 waypoint_radius: float = 2.0, waypoint_num_triangles: int = 10):
        self.color_map = color_map
        self.rendering_levels = rendering_levels
        self.render_agent_direction = render_agent_direction
        self.initialize_background_mesh(background_mesh, world_center)
        self.actor_mesh = None
        self.actor_lenwid = self.actor_types = self.actor_type_names = None
        if agent_attributes is not None:
            assert agent_types is not None and agent_type_names is not None
            self.initialize_actors_mesh(agent_attributes, agent_types, agent_type_names, render_agent_direction)
        self.static_traffic_controls_mesh = self.traffic_lights_mesh = self.traffic_light_colors = None
        if traffic_controls:
            self.initialize_traffic_controls_mesh(traffic_controls)
        self.initialize_waypoint_mesh(waypoint_radius, waypoint_num_triangles)

    def initialize_waypoint_mesh(self, waypoint_radius: float = 2.0, waypoint_num_triangles: int = 10) -> None:
        """one disc per scene, category `goal_waypoint` (mesh.py:885-909)"""
        self.waypoint_radius, self.waypoint_num_triangles = waypoint_radius, waypoint_num_triangles
        B, dev = self.background_mesh.batch_size, self.background_mesh.device
        verts, faces = generate_disc_mesh(radius=waypoint_radius, num_triangles=waypoint_num_triangles, device=dev)
        disc = rendering_mesh(BaseMesh(verts=verts[None].expand(B, -1, -1).clone(), faces=faces[None].expand(B, -1, -1).clone()), 'goal_waypoint')
        self.waypoint_mesh = set_colors_with_defaults(disc, color_map=self.color_map, rendering_levels=self.rendering_levels)

    def initialize_background_mesh(self, background_mesh, world_center: Optional[Tensor] = None):
        if world_center is None:
            if getattr(background_mesh, 'categories', None) and 'road' in background_mesh.categories:
                world_center = self._category_center(background_mesh, 'road')
            else:
                world_center = background_mesh.center
        self.world_center = world_center.to(background_mesh.device)
        if isinstance(background_mesh, BirdviewMesh):
            background_mesh = set_colors_with_defaults(background_mesh.clone(), color_map=self.color_map, rendering_levels=self.rendering_levels)
        self.background_mesh = background_mesh

    @staticmethod
    def _category_center(mesh: BirdviewMesh, category: str) -> Tensor:
        """Centre of the bounding box of the vertices used by the faces of one category -- what
        `mesh.separate_by_category()[category].center` gives (mesh.py:860-866), without the per-scene Python loop."""
        idx = mesh.categories.index(category)
        faces = mesh.faces.long()
        if faces.shape[1] == 0:
            return torch.zeros((mesh.batch_size, 2), dtype=mesh.verts.dtype, device=mesh.device)
        is_cat = mesh.vert_category == idx                                         # B x V
        keep_face = torch.gather(is_cat.unsqueeze(1).expand(-1, faces.shape[1], -1), 2, faces).any(-1)      # B x F
        used = torch.zeros_like(is_cat, dtype=torch.int32).scatter_add_(1, faces.flatten(1), keep_face.unsqueeze(-1).expand(-1, -1, 3).flatten(1).int()) > 0
        big = torch.finfo(mesh.verts.dtype).max
        v = mesh.verts[..., :2]
        lo = torch.where(used.unsqueeze(-1), v, torch.full_like(v, big)).min(dim=1).values
        hi = torch.where(used.unsqueeze(-1), v, torch.full_like(v, -big)).max(dim=1).values
        # unused batch elements (no such face): padded vertex 0 semantics of the reference -> centre of vertex 0
        none = ~used.any(dim=1)
        center = (hi + lo) / 2
        return torch.where(none.unsqueeze(-1), v[:, 0] if v.shape[1] else torch.zeros_like(center), center)

    def add_static_meshes(self, meshes: List[BirdviewMesh]) -> None:
        self.add_static_rgb_meshes([set_colors_with_defaults(m.clone(), color_map=self.color_map, rendering_levels=self.rendering_levels) for m in meshes])

    def add_static_rgb_meshes(self, meshes: List[RGBMesh]) -> None:
        self.background_mesh = RGBMesh.concat([self.background_mesh] + meshes)

    def initialize_actors_mesh(self, agent_attributes: Tensor, agent_types: Tensor, agent_type_names: List[str], render_agent_direction: bool = True):
        self.render_agent_direction = render_agent_direction
        self.actor_lenwid, self.actor_types, self.actor_type_names = agent_attributes[..., :2], agent_types.long(), list(agent_type_names)
        B, N = self.actor_lenwid.shape[:2]
        tmpl = actor_template(self.actor_lenwid)                               # B x N x 7 x 2
        nv = ACTOR_VERTS if render_agent_direction else 4
        faces_one = torch.tensor(ACTOR_FACES[:3 if render_agent_direction else 2], dtype=torch.long, device=tmpl.device)
        faces = (faces_one[None, None] + nv * torch.arange(N, device=tmpl.device)[None, :, None, None]).expand(B, N, -1, 3).reshape(B, -1, 3)
        cats = self.actor_type_names + (['direction'] if render_agent_direction else [])
        vc = self.actor_types.unsqueeze(-1).expand(B, N, 4)
        if render_agent_direction:
            vc = torch.cat([vc, torch.full((B, N, 3), len(self.actor_type_names), dtype=torch.long, device=tmpl.device)], dim=-1)
        mesh = BirdviewMesh(verts=tmpl[..., :nv, :].reshape(B, N * nv, 2), faces=faces, categories=cats, vert_category=vc.reshape(B, N * nv),
                            colors=dict(), zs=dict())
        self.actor_mesh = set_colors_with_defaults(mesh, color_map=self.color_map, rendering_levels=self.rendering_levels)

    # ---- traffic controls (mesh.py:1007-1051): stop lines as quads; lights are coloured by their state at generate() time
    @classmethod
    def _create_traffic_controls_mesh(cls, traffic_controls, selected: List[str]) -> BirdviewMesh:
        batch_size = max([c.corners.shape[0] for c in traffic_controls.values()], default=1)
        meshes = []
        for kind, control in traffic_controls.items():
            if kind not in selected or control.corners.shape[-3] == 0:
                continue
            verts, faces = build_verts_faces_from_bounding_box(control.corners)
            if kind == 'traffic_light':
                meshes.append(BirdviewMesh(verts=verts, faces=faces, categories=[f'{kind}_{st}' for st in control.allowed_states],
                                           vert_category=control.state.unsqueeze(-1).expand(control.state.shape + (4,)).flatten(-2, -1),
                                           zs=dict(), colors=dict()))
            else:
                meshes.append(rendering_mesh(BaseMesh(verts=verts, faces=faces), category=kind))
        return BirdviewMesh.concat(meshes) if meshes else BirdviewMesh.empty(dim=2, batch_size=batch_size)

    def initialize_traffic_controls_mesh(self, traffic_controls) -> None:
        mk = lambda sel: set_colors_with_defaults(self._create_traffic_controls_mesh(traffic_controls, sel), color_map=self.color_map,
                                                  rendering_levels=self.rendering_levels).to(self.background_mesh.device)     # an absent kind is an
        #                                                                                               empty mesh, created on the host
        self.static_traffic_controls_mesh = mk(['stop_sign', 'yield_sign'])
        self.traffic_lights_mesh = mk(['traffic_light'])
        self.traffic_light_colors = None
        if 'traffic_light' in traffic_controls:
            tl = traffic_controls['traffic_light']
            cols = torch.stack([tensor_color(self.color_map[f'traffic_light_{st}'], device=self.traffic_lights_mesh.device) for st in tl.allowed_states])
            self.traffic_light_colors = cols.reshape(1, 1, -1, 3).expand(self.traffic_lights_mesh.batch_size, tl.state.shape[1], -1, -1)

    def add_static_meshes(self, meshes: List[BirdviewMesh]) -> None:
        """more static elements behind the actors: coloured by category and appended to the background (mesh.py:870-883)"""
        self.add_static_rgb_meshes([set_colors_with_defaults(m.clone(), color_map=self.color_map, rendering_levels=self.rendering_levels) for m in meshes])

    def add_static_rgb_meshes(self, meshes: List[RGBMesh]) -> None:
        self.background_mesh = self.background_mesh.concat([self.background_mesh] + list(meshes))

    # ---- batch plumbing
    def to(self, device):
        self.background_mesh = self.background_mesh.to(device)
        self.world_center = self.world_center.to(device)
        if self.actor_mesh is not None:
            self.actor_mesh = self.actor_mesh.to(device)
            self.actor_lenwid, self.actor_types = self.actor_lenwid.to(device), self.actor_types.to(device)
        for name in ('static_traffic_controls_mesh', 'traffic_lights_mesh', 'traffic_light_colors', 'waypoint_mesh'):
            if getattr(self, name, None) is not None:
                setattr(self, name, getattr(self, name).to(device))
        return self

    def _clone_with(self, f, bg_f):
        other = self.__class__.__new__(self.__class__)
        other.color_map, other.rendering_levels = self.color_map.copy(), self.rendering_levels.copy()
        other.render_agent_direction = self.render_agent_direction
        other.background_mesh = bg_f(self.background_mesh)
        other.world_center = f(self.world_center)
        other.actor_mesh = bg_f(self.actor_mesh) if self.actor_mesh is not None else None
        other.actor_lenwid = f(self.actor_lenwid) if self.actor_lenwid is not None else None
        other.actor_types = f(self.actor_types) if self.actor_types is not None else None
        other.actor_type_names = self.actor_type_names
        other.static_traffic_controls_mesh = bg_f(self.static_traffic_controls_mesh) if self.static_traffic_controls_mesh is not None else None
        other.traffic_lights_mesh = bg_f(self.traffic_lights_mesh) if self.traffic_lights_mesh is not None else None
        other.traffic_light_colors = f(self.traffic_light_colors) if self.traffic_light_colors is not None else None
        other.waypoint_radius, other.waypoint_num_triangles = self.waypoint_radius, self.waypoint_num_triangles
        other.waypoint_mesh = bg_f(self.waypoint_mesh)
        return other

    def copy(self):
        return self.expand(1)

    def expand(self, n: int):
        return self._clone_with(lambda x: x.unsqueeze(1).expand((x.shape[0], n) + x.shape[1:]).reshape((n * x.shape[0],) + x.shape[1:]),
                                lambda m: m.expand(n))

    def select_batch_elements(self, idx):
        return self._clone_with(lambda x: x[idx], lambda m: m[idx])

    def generate(self, num_cameras: int, agent_state: Optional[Tensor] = None, present_mask: Optional[Tensor] = None,
                 traffic_lights=None, waypoints: Optional[Tensor] = None, waypoints_rendering_mask: Optional[Tensor] = None,
                 custom_agent_colors: Optional[Tensor] = None) -> RGBMesh:
        """Explicit (B*Nc)-batched RGB mesh = background expanded per camera || posed actors (mesh.py:1053-1157).
        Faces of masked agents are zeroed before the concat, hence alias the first actor vertex."""
        meshes = [self.background_mesh.expand(num_cameras)]
        if agent_state is not None and self.actor_mesh is not None:
            assert agent_state.shape[1] == num_cameras
            actor = self.actor_mesh.expand(num_cameras)
            st = agent_state.flatten(0, 1)
            EB, N, _ = st.shape
            xy = transform(actor.verts[..., :2].reshape(EB * N, -1, 2), st[..., :3].reshape(EB * N, 3)).reshape(EB, -1, 2)
            verts = torch.cat([xy, actor.verts[..., 2:3]], dim=-1)
            faces = actor.faces
            if present_mask is not None:
                pm = present_mask.flatten(0, 1)
                per = faces.shape[1] // pm.shape[1]
                faces = faces * pm[..., None].expand(pm.shape + (per,)).flatten(1, 2)[..., None]
            attrs = actor.attrs
            if custom_agent_colors is not None:
                attrs = attrs.clone()
                step = ACTOR_VERTS if self.render_agent_direction else 4
                cc = custom_agent_colors.flatten(0, 1)
                for i in range(4):
                    attrs[:, i::step] = cc
            meshes.append(dataclasses.replace(actor, verts=verts, faces=faces, attrs=attrs))
        if self.static_traffic_controls_mesh is not None:
            meshes.append(self.static_traffic_controls_mesh.expand(num_cameras))
        if traffic_lights is not None and self.traffic_lights_mesh is not None and self.traffic_lights_mesh.faces_count > 0:
            # colour of every light = colour of its current state, on its four vertices (mesh.py:1105-1118)
            lights = self.traffic_lights_mesh.expand(num_cameras)
            assert traffic_lights.state.shape[0] == lights.batch_size
            table = self.traffic_light_colors[:, None].repeat_interleave(num_cameras, dim=1).flatten(0, 1)
            cur = torch.gather(table, 2, traffic_lights.state[..., None, None].expand(-1, -1, -1, 3))
            meshes.append(dataclasses.replace(lights, attrs=cur.expand(-1, -1, 4, -1).reshape(lights.batch_size, -1, 3)))
        if waypoints is not None:
            # one disc per (camera, waypoint), moved to the waypoint; discs of masked waypoints keep their vertices but their faces are
            # zeroed, so after the concat they alias the first waypoint vertex of the camera (mesh.py:1120-1145)
            assert waypoints.shape[1] == num_cameras
            disc = self.waypoint_mesh
            B, M, nv = disc.batch_size, waypoints.shape[2], disc.verts.shape[1]
            xy = disc.verts[:, None, None, :, :2] + waypoints[..., None, :].to(disc.verts.dtype)                   # B x Nc x M x nv x 2
            z = disc.verts[:, None, None, :, 2:3].expand(-1, num_cameras, M, -1, -1)
            faces = disc.faces[:, None, None] + nv * torch.arange(M, device=disc.device)[None, None, :, None, None]     # B x 1 x M x T x 3
            faces = faces.expand(-1, num_cameras, -1, -1, -1)
            if waypoints_rendering_mask is not None:
                faces = faces * waypoints_rendering_mask.reshape(B, num_cameras, M, 1, 1)
            attrs = disc.attrs[:, None, None].expand(-1, num_cameras, M, -1, -1)
            meshes.append(dataclasses.replace(disc, verts=torch.cat([xy, z], dim=-1).reshape(B * num_cameras, M * nv, 3),
                                              faces=faces.reshape(B * num_cameras, -1, 3), attrs=attrs.reshape(B * num_cameras, M * nv, 3)))
        return RGBMesh.concat(meshes)
