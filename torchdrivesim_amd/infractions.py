"""
Infraction metrics with the reference's names and signatures (torchdrivesim/infractions.py), computed by the K2 HIP
kernels: oriented-box IoU, 5-disc overlap, and off-road distance to the driving-surface mesh.
"""
from typing import List, Optional, Union

import os
import numpy as np
import torch
from torch import Tensor

from torchdrivesim_amd import _ops
from torchdrivesim_amd.mesh import BaseMesh


def iou_differentiable(box1: Tensor, box2: Tensor, fast: bool = True) -> Tensor:
    """Approximate IoU of oriented boxes, element-wise over BxAx5 tensors [x,y,length,width,psi] (infractions.py:307-324,
    _iou_utils.py:344-367).  Forward only; `Simulator.compute_collision` is the differentiable entry point."""
    if not fast:
        raise NotImplementedError('only the fast Rotated-IoU variant exists (the reference imports the slow one from a private package)')
    return _ops.pairwise_overlap(box1, box2, 'iou')


def collision_detection_with_discs(box1: Tensor, box2: Tensor, num_discs: int = 5, backend: str = 'torch') -> Tensor:
    """TrafficSim-style overlap of `num_discs` discs per box, relu(1 - d / (r1 + r2)) (infractions.py:503-545); `num_discs` odd,
    3 .. 25 (torch.cdist, which the reference calls, changes its formulation above 25 points per set)."""
    assert isinstance(num_discs, int) and num_discs > 1 and num_discs % 2 != 0          # bbox2discs, infractions.py:391
    if backend != 'torch':
        raise ValueError('Unknown backend framework.')
    return _ops.pairwise_overlap(box1, box2, 'discs', num_discs=num_discs)


def box2corners_th(box: Tensor) -> Tensor:
    """(B,N,5) boxes -> (B,N,4,2) corners (_iou_utils.py:270-299)"""
    return _ops.box2corners(box)


#: grid cell (metres) of the geometry-only maps built for the off-road query: point location wants cells of about a face's size
#: (the rasteriser's own maps keep the library default, 8 m, which suits its row scans)
OFFROAD_CELL_SIZE = float(os.environ.get('TDS_OFFROAD_CELL', 3.0))


def _static_maps_for(mesh: BaseMesh, device) -> List:
    """The geometry-only device map(s) of a batch of meshes -- ONE per distinct batch element (`_ops.group_rows`), taken from the process-wide
    content cache or built once -- remembered on the mesh object: [(StaticMap or StaticMapSet, None)]."""
    key = (mesh.verts.data_ptr(), mesh.faces.data_ptr(), tuple(mesh.verts.shape), tuple(mesh.faces.shape), str(device))
    cache = getattr(mesh, '_tds_offroad_maps', None)
    if cache is not None and cache[0] == key:
        return cache[1]
    dev = torch.device(device)
    tensors = [t.detach() if t.is_cuda else t.detach().to(dev) for t in (mesh.verts, mesh.faces)]
    scene_map, reps, hashes = _ops.group_rows(tensors)
    per_group = []
    for r, h in zip(reps, hashes):
        rows = [t[r] for t in tensors]
        ckey = ('offroad', str(dev), h, tuple(tuple(x.shape) for x in rows), OFFROAD_CELL_SIZE)
        per_group.append(_ops.map_cache.get(ckey, rows, lambda rows=rows: _ops.StaticMap(rows[0][..., :2], rows[1], device=dev, cell_size=OFFROAD_CELL_SIZE)))
    if len(per_group) == 1:
        maps = [(per_group[0], None)]
    else:
        maps = [(_ops.StaticMapSet(per_group, torch.from_numpy(scene_map)), None)]           # one launch for the batch
    try:
        object.__setattr__(mesh, '_tds_offroad_maps', (key, maps))
    except Exception:
        pass
    return maps


def offroad_infraction_loss(agent_states: Tensor, lenwid: Tensor, driving_surface_mesh: Union[BaseMesh, _ops.StaticMap],
                            threshold: float = 0, use_pytorch3d: Optional[bool] = None) -> Tensor:
    """Sum over the 4 agent corners of the thresholded SQUARED distance to the mesh (infractions.py:176-229, the pure
    torch path: `F.threshold(d, threshold, 0)` keeps d).  agent_states BxAx4, lenwid BxAx2 or Bx2 -> BxA."""
    if use_pytorch3d:
        raise NotImplementedError('pytorch3d is not part of this framework; the HIP kernel follows the pure-torch path')
    B, A = agent_states.shape[:2]
    if isinstance(driving_surface_mesh, _ops.StaticMap):
        maps = [(driving_surface_mesh, None)]
        n_faces = driving_surface_mesh.n_faces
    else:
        n_faces = driving_surface_mesh.faces_count
        maps = None
    if A == 0 or n_faces == 0:
        return torch.zeros_like(agent_states[..., 0])
    if lenwid.dim() == 2:
        lenwid = lenwid.unsqueeze(-2).expand((lenwid.shape[0], A, lenwid.shape[1]))
    if maps is None:
        maps = _static_maps_for(driving_surface_mesh, agent_states.device)
    if maps[0][1] is None:
        return _ops.offroad(maps[0][0], agent_states, lenwid, threshold=threshold)
    return torch.cat([_ops.offroad(m, agent_states[b:b + 1], lenwid[b:b + 1], threshold=threshold) for m, b in maps], dim=0)


LANELET_TAGS_TO_EXCLUDE = ['parking']          # infractions.py:21


def lane_table_set(lanelet_maps, device, lanelet_dist_tolerance: float = 1.0) -> Optional[_ops.LaneTableSet]:
    """The device lane tables of a batch of maps (one table per DISTINCT map object, entries `None` -> no table)."""
    uniq, index = [], {}
    scene_map = []
    for m in lanelet_maps:
        if m is None:
            scene_map.append(-1)
            continue
        if id(m) not in index:
            index[id(m)] = len(uniq)
            uniq.append(m)
        scene_map.append(index[id(m)])
    if not uniq:
        return None
    tables = [m.table(device, LANELET_TAGS_TO_EXCLUDE, lanelet_dist_tolerance) for m in uniq]
    return _ops.LaneTableSet(tables, torch.tensor(scene_map, dtype=torch.int32))


def lanelet_orientation_loss(lanelet_maps, agents_state: Tensor, recenter_offset: Optional[Tensor] = None,
                             direction_angle_threshold: float = np.pi / 2, lanelet_dist_tolerance: float = 1.0,
                             lane_set: Optional[_ops.LaneTableSet] = None) -> Tensor:
    """
    Wrong-way loss (infractions.py:232-304): for every agent the lanelets within `lanelet_dist_tolerance` of its position, the
    local direction of each one's centre line, d = normalize_angle(direction - psi), loss = min over them of
    -cos(d) * [|d| > direction_angle_threshold]; 0 without such a lanelet, for scenes whose map is None, near a lanelet tagged
    'parking', and where the reference's `find_direction` raises.  One kernel launch for the batch (csrc/lanes.hip) instead of the
    reference's Python loop over scenes, agents and lanelets.  `lanelet_maps`: a list of B `torchdrivesim_amd.lanelet2.LaneletMap`
    or None; `lane_set`: the prepared tables of these maps (callers that query every step keep it).
    """
    assert len(lanelet_maps) == agents_state.shape[0]
    if recenter_offset is not None:
        assert len(lanelet_maps) == recenter_offset.shape[0]
    assert direction_angle_threshold >= np.pi / 2, 'direction_angle_threshold smaller than pi / 2 will produce false positives'
    if lane_set is None:
        lane_set = lane_table_set(lanelet_maps, agents_state.device, lanelet_dist_tolerance)
    if lane_set is None or agents_state.shape[1] == 0:
        return torch.zeros(agents_state.shape[:2], dtype=torch.float, device=agents_state.device)
    return _ops.wrong_way(lane_set, agents_state, recenter_offset, None, direction_angle_threshold, lanelet_dist_tolerance)
