"""
`Simulator` facade with the reference's surface (torchdrivesim/simulator.py:280-1194): constructor injection of a
`KinematicModel` and a `BirdviewRenderer`, `TorchDriveConfig.collision_metric` dispatch, `step`, `render`,
`render_egocentric`, `compute_collision`, `compute_offroad`, `compute_wrong_way`, state accessors and the batch plumbing
(`copy`, `extend`, `select_batch_elements`, `to`).  Every hot call goes to a HIP kernel:

    step               -> K1 (kinematic.py)                                  reference: simulator.py:841-861
    render[_egocentric]-> K3 fused scene path (rendering/hip.py)             reference: simulator.py:920-1033
    compute_collision  -> K2a, all agents of all scenes in one launch        reference: simulator.py:1064-1194 (A launches)
    compute_offroad    -> K2b over the device-resident map grid              reference: simulator.py:1035-1044

    compute_wrong_way  -> lane tables of `lanelet_map` (lanelet2.py), one launch  reference: simulator.py:607-630 (Python triple loop)

Also carried: traffic controls, waypoint goals (state + rendering), observation noise and lane features (plumbing only).  Out of scope
(SURVEY.md section 8): lanelet-derived lane features (the tensors are supplied by the caller).
"""
import logging
from dataclasses import dataclass, field
from enum import Enum
from typing import Any, Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from torchdrivesim_amd import _ops
from torchdrivesim_amd.infractions import lane_table_set
from torchdrivesim_amd.kinematic import KinematicModel
from torchdrivesim_amd.mesh import BirdviewMesh, BirdviewRGBMeshGenerator, actor_template, set_colors_with_defaults
from torchdrivesim_amd.rendering import BirdviewRenderer, RendererConfig, renderer_from_config, HipRenderer
from torchdrivesim_amd.utils import Resolution, is_inside_polygon, relative, assert_equal

logger = logging.getLogger(__name__)


class CollisionMetric(Enum):
    """How collisions between agents are measured (simulator.py:27-34)."""
    iou = 'iou'                            #: approximate differentiable IoU of oriented rectangles
    discs = 'discs'                        #: differentiable overlap of rectangles approximated by 5 discs (default)
    nograd = 'nograd'                      #: number of other present agents whose rectangle overlaps (no gradient)
    nograd_pytorch3d = 'nograd-pytorch3d'  #: not available here


@dataclass
class TorchDriveConfig:
    """Top-level simulator configuration (simulator.py:37-51)."""
    renderer: RendererConfig = field(default_factory=lambda: RendererConfig())
    single_agent_rendering: bool = False
    collision_metric: CollisionMetric = field(default_factory=lambda: CollisionMetric.discs)
    offroad_threshold: float = 0.5
    left_handed_coordinates: bool = False
    wrong_way_angle_threshold: float = np.pi / 2
    lanelet_inclusion_tolerance: float = 1.0
    waypoint_removal_threshold: float = 2.0


def _enlarge(x: Tensor, n: int) -> Tensor:
    return x.unsqueeze(1).expand((x.shape[0], n) + x.shape[1:]).reshape((n * x.shape[0],) + x.shape[1:])


class SpawnController:
    """Despawns NPCs that leave `exit_boundary` and spawns scheduled ones (simulator.py:54-124)."""

    def __init__(self, exit_boundary: Optional[Tensor] = None, spawn_states: Optional[Tensor] = None, spawn_masks: Optional[Tensor] = None):
        self.exit_boundary, self.spawn_states, self.spawn_masks = exit_boundary, spawn_states, spawn_masks
        self.time = 0

    def spawn_despawn_npcs(self, simulator: 'Simulator') -> None:
        ctrl = simulator.npc_controller
        mask, states = ctrl.npc_present_mask, ctrl.npc_state
        if self.exit_boundary is not None:
            mask = mask & is_inside_polygon(states[..., :2], self.exit_boundary)
        if self.spawn_states is not None and self.spawn_masks is not None:
            to_spawn = self.spawn_masks[..., self.time] & ~mask
            mask = mask | to_spawn
            states = self.spawn_states[..., self.time, :].where(to_spawn.unsqueeze(-1), states)
        ctrl.npc_present_mask, ctrl.npc_state = mask, states
        self.time += 1

    def _map(self, f):
        for k in ('exit_boundary', 'spawn_states', 'spawn_masks'):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, f(v))
        return self

    def to(self, device):
        return self._map(lambda x: x.to(device))

    def copy(self):
        return self.__class__(self.exit_boundary, self.spawn_states, self.spawn_masks)

    def extend(self, n, in_place=True):
        return (self if in_place else self.copy())._map(lambda x: _enlarge(x, n))

    def select_batch_elements(self, idx, in_place=True):
        return (self if in_place else self.copy())._map(lambda x: x[idx])


class NPCController:
    """Non-playable agents; the base class leaves them where they are (simulator.py:128-203)."""

    def __init__(self, npc_size: Tensor, npc_state: Tensor, npc_present_mask: Optional[Tensor] = None, npc_types: Optional[Tensor] = None,
                 agent_type_names: Optional[List[str]] = None, spawn_controller: Optional[SpawnController] = None):
        self.npc_size, self.npc_state = npc_size, npc_state
        self.npc_present_mask = npc_present_mask if npc_present_mask is not None else torch.ones_like(npc_state[..., 0], dtype=torch.bool)
        self.npc_types = npc_types if npc_types is not None else torch.zeros_like(self.npc_present_mask).long()
        self.agent_type_names = agent_type_names if agent_type_names is not None else ['vehicle']
        self.spawn_controller = spawn_controller if spawn_controller is not None else SpawnController()

    def get_npc_state(self):
        return self.npc_state

    def get_npc_size(self):
        return self.npc_size

    def get_npc_types(self):
        return self.npc_types

    def get_npc_present_mask(self):
        return self.npc_present_mask

    def spawn_despawn_npcs(self, simulator: 'Simulator') -> None:
        self.spawn_controller.spawn_despawn_npcs(simulator)

    def advance_npcs(self, simulator: 'Simulator') -> None:
        self.spawn_despawn_npcs(simulator)

    def _map(self, f):
        self.npc_size, self.npc_state = f(self.npc_size), f(self.npc_state)
        self.npc_present_mask, self.npc_types = f(self.npc_present_mask), f(self.npc_types)
        return self

    def to(self, device):
        self.spawn_controller.to(device)
        return self._map(lambda x: x.to(device))

    def copy(self):
        return self.__class__(self.npc_size, self.npc_state, self.npc_present_mask, self.npc_types, self.agent_type_names, self.spawn_controller.copy())

    def extend(self, n, in_place=True):
        me = self if in_place else self.copy()
        me.spawn_controller.extend(n, in_place=True)
        return me._map(lambda x: _enlarge(x, n))

    def select_batch_elements(self, idx, in_place=True):
        me = self if in_place else self.copy()
        me.spawn_controller.select_batch_elements(idx, in_place=True)
        return me._map(lambda x: x[idx])


class CompoundNPCController(NPCController):
    """
    Several NPC controllers over one set of NPCs: `controller_indices` (BxNpc) says which controller owns which NPC
    (simulator.py:206-278).  After every advance the owners' results are merged and the merged tensors handed back to ALL controllers,
    so that each of them sees the whole scene.  Deviation: `extend` repeats `controller_indices` per scene like every other tensor
    (the reference's `expand(n, -1)` only works for a batch of one).
    """

    def __init__(self, controllers: List[NPCController], controller_indices: Tensor):
        B, n = controller_indices.shape
        dev = controller_indices.device
        super().__init__(torch.zeros((B, n, 2), device=dev), torch.zeros((B, n, 4), device=dev), torch.zeros((B, n), device=dev, dtype=torch.bool),
                         None if controllers[0].npc_types is None else torch.zeros((B, n), device=dev, dtype=torch.long), controllers[0].agent_type_names)
        self.controllers = controllers
        self.controller_indices = controller_indices
        self.gather_npc_states()

    def gather_npc_states(self):
        for i, c in enumerate(self.controllers):
            mine = self.controller_indices == i
            self.npc_size = c.npc_size.where(mine.unsqueeze(-1), self.npc_size)
            self.npc_state = c.npc_state.where(mine.unsqueeze(-1), self.npc_state)
            self.npc_present_mask = c.npc_present_mask.where(mine, self.npc_present_mask)
            self.npc_types = c.npc_types.where(mine, self.npc_types)
        for c in self.controllers:
            c.npc_size, c.npc_state, c.npc_present_mask, c.npc_types = self.npc_size, self.npc_state, self.npc_present_mask, self.npc_types

    def advance_npcs(self, simulator: 'Simulator') -> None:
        for c in self.controllers:
            c.advance_npcs(simulator)
        self.gather_npc_states()

    def to(self, device):
        super().to(device)
        self.controller_indices = self.controller_indices.to(device)
        for c in self.controllers:
            c.to(device)
        return self

    def copy(self):
        return self.__class__([c.copy() for c in self.controllers], self.controller_indices.clone())

    def extend(self, n, in_place=True):
        me = self if in_place else self.copy()
        NPCController.extend(me, n, in_place=True)
        me.controller_indices = _enlarge(me.controller_indices, n)
        for c in me.controllers:
            c.extend(n, in_place=True)
        return me

    def select_batch_elements(self, idx, in_place=True):
        me = self if in_place else self.copy()
        NPCController.select_batch_elements(me, idx, in_place=True)
        me.controller_indices = me.controller_indices[idx]
        for c in me.controllers:
            c.select_batch_elements(idx, in_place=True)
        return me


class Simulator:
    """Batched 2-D driving simulator; see the module docstring.  Arguments follow simulator.py:283-309."""

    def __init__(self, road_mesh: BirdviewMesh, kinematic_model: KinematicModel, agent_size: Tensor, initial_present_mask: Tensor,
                 cfg: TorchDriveConfig, renderer: Optional[BirdviewRenderer] = None, lanelet_map=None, recenter_offset: Optional[Tensor] = None,
                 birdview_mesh_generator: Optional[BirdviewRGBMeshGenerator] = None, internal_time: int = 0, traffic_controls=None,
                 waypoint_goals=None, agent_types: Optional[Tensor] = None, agent_type_names: Optional[List[str]] = None,
                 npc_controller: Optional[NPCController] = None, agent_lr: Optional[Tensor] = None, lane_features=None,
                 observation_noise_model=None, action_model_extras: Optional[Dict[str, Any]] = None):
        self._lane_set = None                            # device lane tables of `lanelet_map`, made on the first compute_wrong_way
        self.road_mesh = road_mesh
        self.lanelet_map = lanelet_map
        self.recenter_offset = recenter_offset
        self.kinematic_model = kinematic_model
        self.agent_size = agent_size
        self.present_mask = initial_present_mask
        self.action_model_extras = action_model_extras
        self.traffic_controls = traffic_controls        # Dict[str, BaseTrafficControl]: state and violations; not rendered by the fused path
        self.waypoint_goals = waypoint_goals             # WaypointGoal: ticked off in step(), drawn as discs by render_egocentric
        self.lane_features = lane_features               # lanelet2.LaneFeatures: carried for policies, not used by any kernel
        if observation_noise_model is None:
            from torchdrivesim_amd.observation_noise import ObservationNoise
            observation_noise_model = ObservationNoise()
        self.observation_noise_model = observation_noise_model

        if not agent_type_names:
            agent_type_names = ['vehicle']
        if agent_types is None:
            agent_types = torch.zeros_like(initial_present_mask).long()
        if len(agent_types) == 1:
            agent_types = agent_types.expand_as(initial_present_mask)
        if agent_lr is None:
            agent_lr = torch.zeros_like(initial_present_mask).to(agent_size.dtype)
        if len(agent_lr) == 1:
            agent_lr = agent_lr.expand_as(initial_present_mask)
        self._agent_types = agent_type_names
        self._batch_size = self.road_mesh.batch_size
        self.agent_type = agent_types
        self.agent_lr = agent_lr

        self.npc_controller = npc_controller
        if self.npc_controller is None:
            dev = initial_present_mask.device
            self.npc_controller = NPCController(
                npc_size=torch.zeros((self._batch_size, 0, 2), dtype=self.agent_size.dtype, device=dev),
                npc_state=torch.zeros((self._batch_size, 0, 4), dtype=self.get_state().dtype, device=dev),
                npc_present_mask=torch.zeros((self._batch_size, 0), dtype=torch.bool, device=dev),
                npc_types=torch.zeros((self._batch_size, 0), dtype=torch.long, device=dev), agent_type_names=agent_type_names)
        self.validate_tensor_shapes()

        self.cfg: TorchDriveConfig = cfg
        if renderer is None:
            cfg.renderer.left_handed_coordinates = cfg.left_handed_coordinates
            self.renderer: BirdviewRenderer = renderer_from_config(cfg=cfg.renderer)
        else:
            self.renderer = renderer
        if cfg.left_handed_coordinates:
            self.kinematic_model.left_handed = cfg.left_handed_coordinates
        self.warned_no_lanelet = False
        self.internal_time = internal_time

        if birdview_mesh_generator is None:
            self.birdview_mesh_generator = BirdviewRGBMeshGenerator(background_mesh=self.road_mesh, color_map=self.renderer.color_map,
                                                                    rendering_levels=self.renderer.rendering_levels)
            self.birdview_mesh_generator.initialize_actors_mesh(self.get_all_agent_size(), self.get_all_agent_type(), self.agent_types)
        else:
            self.birdview_mesh_generator = birdview_mesh_generator
        if self.traffic_controls and getattr(self.birdview_mesh_generator, 'traffic_lights_mesh', None) is None:
            self.birdview_mesh_generator.initialize_traffic_controls_mesh(self.traffic_controls)      # simulator.py:373-374
        self._scene_cache = None        # device-resident static maps + actor templates/keys, rebuilt lazily
        self._fork = None               # (event, stamp, stream, results) recorded right before the last raster launch, see _beside_render
        self._fork_used = []            # metrics asked for since then: enqueued ahead of the next raster launch

    # ------------------------------------------------------------------------------------------------- properties
    @property
    def agent_types(self) -> Optional[List[str]]:
        return self._agent_types

    @property
    def action_size(self) -> int:
        return self.kinematic_model.action_size

    @property
    def batch_size(self) -> int:
        return self._batch_size

    @property
    def agent_count(self) -> int:
        return self.get_agent_size().shape[-2]

    @property
    def npc_count(self) -> int:
        return self.get_npc_size().shape[-2]

    # ------------------------------------------------------------------------------------------------- plumbing
    def to(self, device):
        self.road_mesh = self.road_mesh.to(device)
        self.recenter_offset = self.recenter_offset.to(device) if self.recenter_offset is not None else None
        self.agent_size, self.agent_type = self.agent_size.to(device), self.agent_type.to(device)
        self.agent_lr, self.present_mask = self.agent_lr.to(device), self.present_mask.to(device)
        self.kinematic_model = self.kinematic_model.to(device)
        self.birdview_mesh_generator = self.birdview_mesh_generator.to(device)
        self.npc_controller = self.npc_controller.to(device)
        if self.traffic_controls is not None:
            self.traffic_controls = {k: v.to(device) for k, v in self.traffic_controls.items()}
        self.waypoint_goals = self.waypoint_goals.to(device) if self.waypoint_goals is not None else None
        self.lane_features = self.lane_features.to(device) if self.lane_features is not None else None
        self._scene_cache = None
        return self

    def copy(self):
        """Independent simulator sharing the tensors (shallow, simulator.py:421-442)."""
        other = self.__class__(
            road_mesh=self.road_mesh, kinematic_model=self.kinematic_model.copy(), agent_size=self.agent_size,
            initial_present_mask=self.present_mask, cfg=self.cfg, renderer=self.renderer.copy(), lanelet_map=self.lanelet_map,
            birdview_mesh_generator=self.birdview_mesh_generator.copy(), recenter_offset=self.recenter_offset, internal_time=self.internal_time,
            agent_types=self.agent_type, agent_type_names=self.agent_types, agent_lr=self.agent_lr, npc_controller=self.npc_controller.copy(),
            traffic_controls={k: v.copy() for k, v in self.traffic_controls.items()} if self.traffic_controls is not None else None,
            waypoint_goals=self.waypoint_goals.copy() if self.waypoint_goals is not None else None,
            observation_noise_model=self.observation_noise_model, lane_features=self.lane_features.copy() if self.lane_features is not None else None)
        other._scene_cache = self._scene_cache          # static maps are immutable and can be shared
        return other

    def extend(self, n: int, in_place: bool = True):
        if not in_place:
            other = self.copy()
            other.extend(n, in_place=True)
            return other
        self.road_mesh = self.road_mesh.expand(n)
        self.agent_size, self.agent_type = _enlarge(self.agent_size, n), _enlarge(self.agent_type, n)
        self.agent_lr, self.present_mask = _enlarge(self.agent_lr, n), _enlarge(self.present_mask, n)
        self.recenter_offset = _enlarge(self.recenter_offset, n) if self.recenter_offset is not None else None
        self.lanelet_map = [m for m in self.lanelet_map for _ in range(n)] if self.lanelet_map is not None else None
        self._lane_set = None
        self.lane_features = self.lane_features.extend(n) if self.lane_features is not None else None
        self.kinematic_model.extend(n)
        self._batch_size *= n
        self.birdview_mesh_generator = self.birdview_mesh_generator.expand(n)
        self.npc_controller = self.npc_controller.extend(n)
        if self.traffic_controls is not None:
            self.traffic_controls = {k: v.extend(n) for k, v in self.traffic_controls.items()}
        if self.waypoint_goals is not None:
            self.waypoint_goals = self.waypoint_goals.extend(n)
        self._scene_cache = None
        return self

    def select_batch_elements(self, idx, in_place=True):
        if not in_place:
            other = self.copy()
            other.select_batch_elements(idx, in_place=True)
            return other
        self.road_mesh = self.road_mesh[idx]
        self.recenter_offset = self.recenter_offset[idx] if self.recenter_offset is not None else None
        self.lanelet_map = [self.lanelet_map[i] for i in idx] if self.lanelet_map is not None else None
        self._lane_set = None
        self.lane_features = self.lane_features.select_batch_elements(idx) if self.lane_features is not None else None
        self.agent_size, self.agent_type = self.agent_size[idx], self.agent_type[idx]
        self.agent_lr, self.present_mask = self.agent_lr[idx], self.present_mask[idx]
        self.kinematic_model.select_batch_elements(idx)
        self._batch_size = len(idx)
        self.birdview_mesh_generator = self.birdview_mesh_generator.select_batch_elements(idx)
        self.npc_controller = self.npc_controller.select_batch_elements(idx)
        if self.traffic_controls is not None:
            self.traffic_controls = {k: v.select_batch_elements(idx) for k, v in self.traffic_controls.items()}
        if self.waypoint_goals is not None:
            self.waypoint_goals = self.waypoint_goals.select_batch_elements(idx)
        self._scene_cache = None
        return self

    def __getitem__(self, item):
        return self.select_batch_elements(item, in_place=False)

    def validate_agent_types(self):
        return

    def validate_tensor_shapes(self):
        assert_equal(len(self.kinematic_model.get_state().shape), 3)
        assert_equal(len(self.agent_size.shape), 3)
        assert_equal(len(self.agent_type.shape), 2)
        assert_equal(len(self.agent_lr.shape), 2)
        assert_equal(len(self.present_mask.shape), 2)
        b = self.batch_size
        for t in (self.kinematic_model.get_state(), self.agent_size, self.agent_type, self.agent_lr, self.present_mask):
            assert_equal(t.shape[0], b)
        assert_equal(self.road_mesh.batch_size, b)
        n = self.agent_count
        assert_equal(self.kinematic_model.get_state().shape[-2], n)
        assert_equal(self.agent_type.shape[-1], n)
        assert_equal(self.agent_lr.shape[-1], n)
        assert_equal(self.present_mask.shape[-1], n)

    # ------------------------------------------------------------------------------------------------- accessors
    def get_action_model_extras(self) -> Dict[str, Any]:
        return dict(self.action_model_extras) if self.action_model_extras else {}

    def get_world_center(self) -> Tensor:
        return self.birdview_mesh_generator.world_center

    def get_state(self) -> Tensor:
        return self.kinematic_model.get_state()

    def get_waypoints(self, count: int = 1):
        """B x A x count*M x 2 current waypoints of the agents, or None (simulator.py:589-593)"""
        return self.waypoint_goals.get_waypoints(count=count) if self.waypoint_goals is not None else None

    def get_waypoints_state(self):
        return self.waypoint_goals.state if self.waypoint_goals is not None else None

    def get_waypoints_mask(self, count: int = 1):
        return self.waypoint_goals.get_masks(count=count) if self.waypoint_goals is not None else None

    def get_agent_size(self) -> Tensor:
        return self.agent_size

    def get_agent_type(self) -> Tensor:
        return self.agent_type

    def get_agent_type_names(self) -> List[str]:
        return self._agent_types

    def get_agent_lr(self) -> Tensor:
        return self.agent_lr

    def get_present_mask(self) -> Tensor:
        return self.present_mask

    def get_npc_state(self) -> Tensor:
        return self.npc_controller.get_npc_state()

    def get_npc_size(self) -> Tensor:
        return self.npc_controller.get_npc_size()

    def get_npc_present_mask(self) -> Tensor:
        return self.npc_controller.get_npc_present_mask()

    def get_npc_types(self) -> Tensor:
        return self.npc_controller.get_npc_types()

    def get_all_agent_state(self) -> Tensor:
        return self.get_state() if self.npc_count == 0 else torch.cat([self.get_state(), self.get_npc_state()], dim=-2)

    def _heading_sc(self) -> Tensor:
        """[sin psi, cos psi] of ALL agents, computed once per state tensor and shared by render / collision / off-road (each of them
        would otherwise launch its own sin and cos); when gradients are being recorded, one shared autograd node.  The cache is keyed on the
        IDENTITY of the state tensor (which it keeps alive, so the allocator cannot hand its block to a later state) and on its
        version counter; a kinematic model whose get_state() builds a fresh tensor per call (CompoundKinematicModel) or a scene
        with NPCs (torch.cat per call) therefore never hits it."""
        state = self.get_all_agent_state()
        if state.requires_grad and torch.is_grad_enabled():
            # gradients are being recorded: ONE autograd node per state tensor (_ops.state_heading_sc), shared by render / collision / off-road
            # like the plain cache below -- autograd sums the three consumers' gradients before the node's closed-form backward runs once
            cached = getattr(self, '_sc_cache_grad', None)
            if cached is not None and cached[0] is state and cached[1] == state._version:
                return cached[2]
            sc = _ops.state_heading_sc(state)
            self._sc_cache_grad = (state, state._version, sc)
            return sc
        cached = getattr(self, '_sc_cache', None)
        if cached is not None and cached[0] is state and cached[1] == state._version:
            return cached[2]
        sc = _ops.heading_sc(state[..., 2].detach())
        self._sc_cache = (state, state._version, sc)
        return sc

    def get_all_agent_size(self) -> Tensor:
        return self.get_agent_size() if self.npc_count == 0 else torch.cat([self.get_agent_size(), self.get_npc_size()], dim=-2)

    def get_all_agent_present_mask(self) -> Tensor:
        return self.get_present_mask() if self.npc_count == 0 else torch.cat([self.get_present_mask(), self.get_npc_present_mask()], dim=-1)

    def get_all_agent_type(self) -> Tensor:
        return self.get_agent_type() if self.npc_count == 0 else torch.cat([self.get_agent_type(), self.get_npc_types()], dim=-1)

    def get_all_agents_absolute(self) -> Tensor:
        """Bx(A+Npc)x6: x, y, psi, length, width, present (simulator.py:730-738)."""
        return torch.cat([self.get_all_agent_state()[..., :3], self.get_all_agent_size(),
                          self.get_all_agent_present_mask().unsqueeze(-1).to(self.get_state().dtype)], dim=-1)

    def get_all_agents_relative(self, exclude_self: bool = True) -> Tensor:
        """BxAx(All[-1])x6: pose of every agent in the frame of each exposed agent (simulator.py:748-782); the diagonal is
        removed with a reshape trick instead of boolean indexing, so no device synchronisation is needed."""
        absolute = self.get_all_agents_absolute()
        A, total = self.agent_count, self.agent_count + self.npc_count
        xy, psi = absolute[..., :A, :2], absolute[..., :A, 2:3]
        rel_xy, rel_psi = relative(origin_xy=xy.unsqueeze(-2), origin_psi=psi.unsqueeze(-2), target_xy=absolute[..., :2].unsqueeze(-3),
                                   target_psi=absolute[..., 2:3].unsqueeze(-3))
        rel = torch.cat([rel_xy, rel_psi, absolute[..., 3:].unsqueeze(-3).expand(rel_xy.shape[:-1] + (3,))], dim=-1)
        if exclude_self:
            if A == 1:
                rel = rel[..., 1:, :]
            else:
                B = rel.shape[0]
                own, npc = rel[..., :A, :], rel[..., A:, :]
                # drop element (i, i) of an A x A block: flatten, skip every (A+1)-th entry
                own = own.reshape(B, A * A, 6)[:, 1:, :].reshape(B, A - 1, A + 1, 6)[:, :, :A, :].reshape(B, A, A - 1, 6)
                rel = torch.cat([own, npc], dim=-2)
                assert rel.shape[-2] == total - 1
        return rel

    # ---- what the exposed agents perceive (simulator.py:663-679, 740-746, 784-821)
    def get_noisy_state(self) -> Tensor:
        """BxAx(A+Npc)x4: the state of every agent as perceived by each exposed agent"""
        return self.observation_noise_model.get_noisy_state(self)

    def get_noisy_agent_size(self) -> Tensor:
        return self.observation_noise_model.get_noisy_agent_size(self)

    def get_noisy_present_mask(self) -> Tensor:
        return self.observation_noise_model.get_noisy_present_mask(self)

    def get_noisy_all_agents_absolute(self) -> Tensor:
        """BxAx(A+Npc)x6 [x, y, psi, length, width, present] in world coordinates, per observer"""
        return torch.cat([self.get_noisy_state()[..., :3], self.get_noisy_agent_size(), self.get_noisy_present_mask()[..., None]], dim=-1)

    def get_noisy_all_agents_relative(self, exclude_self: bool = True) -> Tensor:
        """BxAx(A+Npc [-1])x6 in the frame of each observer's OWN perceived pose; `exclude_self` drops the observer from its own list"""
        ab = self.get_noisy_all_agents_absolute()
        A, total = self.agent_count, self.agent_count + self.npc_count
        idx = torch.arange(A, device=ab.device)
        own = ab[:, idx, idx, :]
        rel_xy, rel_psi = relative(origin_xy=own[..., :2].unsqueeze(-2), origin_psi=own[..., 2:3].unsqueeze(-2), target_xy=ab[..., :2],
                                   target_psi=ab[..., 2:3])
        rel = torch.cat([rel_xy, rel_psi, ab[..., 3:]], dim=-1)
        if exclude_self:
            if A == 1:
                return rel[..., 1:, :]
            # drop column a of row a without boolean indexing (no device synchronisation): column j of the result is j + (j >= a)
            col = torch.arange(total - 1, device=ab.device)[None, :] + (torch.arange(total - 1, device=ab.device)[None, :] >= idx[:, None]).long()
            rel = torch.gather(rel, 2, col[None, :, :, None].expand(rel.shape[0], -1, -1, rel.shape[-1]))
        return rel

    def get_noisy_traffic_controls(self):
        return self.observation_noise_model.get_noisy_traffic_controls(self)

    def get_noisy_lane_features(self):
        return self.observation_noise_model.get_noisy_lane_features(self)

    def get_noisy_road_mesh(self):
        return self.observation_noise_model.get_noisy_road_mesh(self)

    def get_noisy_background_mesh(self):
        return self.observation_noise_model.get_noisy_background_mesh(self)

    def get_traffic_controls(self):
        return self.traffic_controls

    # ------------------------------------------------------------------------------------------------- dynamics
    def step(self, agent_action: Tensor) -> None:
        """One simulation step for BxAxAc actions (simulator.py:841-861)."""
        self.internal_time += 1
        assert_equal(len(agent_action.shape), 3)
        assert_equal(agent_action.shape[0], self.batch_size)
        assert_equal(agent_action.shape[-2], self.agent_count)
        self.npc_controller.advance_npcs(self)
        self._sc_cache_grad = None               # the shared [sin, cos] node of the state that is being left: do not pin its graph
        self.kinematic_model.step(agent_action)
        if self.traffic_controls is not None:                       # simulator.py:857-859
            for control in self.traffic_controls.values():
                control.step(self.internal_time)
        if self.waypoint_goals is not None:                         # simulator.py:860-861
            self.waypoint_goals.step(self.get_state(), self.internal_time, threshold=self.cfg.waypoint_removal_threshold)

    def set_state(self, agent_state: Tensor, mask: Optional[Tensor] = None) -> None:
        if mask is None:
            mask = torch.ones_like(agent_state[..., 0], dtype=torch.bool)
        assert_equal(len(agent_state.shape), 3)
        assert_equal(len(mask.shape), 2)
        assert_equal(agent_state.shape[0], self.batch_size)
        assert_equal(agent_state.shape[-2], self.agent_count)
        current = self.kinematic_model.get_state()
        k, full = agent_state.shape[-1], current.shape[-1]
        assert k <= full
        state = agent_state if k == full else torch.cat([agent_state, current[..., (k - full):]], dim=-1)
        self._sc_cache_grad = None
        self.kinematic_model.set_state(state.where(mask.unsqueeze(-1).expand_as(state), current))

    def update_present_mask(self, present_mask: Tensor) -> None:
        assert_equal(len(present_mask.shape), 2)
        assert_equal(present_mask.shape[0], self.batch_size)
        assert_equal(present_mask.shape[-1], self.agent_count)
        self.present_mask = present_mask

    def fit_action(self, future_state: Tensor, current_state: Optional[Tensor] = None) -> Tensor:
        return self.kinematic_model.fit_action(future_state=future_state, current_state=current_state)

    # ------------------------------------------------------------------------------------------------- infractions beside the rasteriser
    #: The GymEnv.step body (examples/gym_env.py:83-126 of the reference) is step -> render_egocentric -> compute_collision / compute_offroad /
    #: compute_wrong_way.  The rasteriser is bound by the HBM write stream and the metrics are compute-light and do not read the image, so they
    #: can run on a second HIP stream that waits only for what preceded the raster launch (an event, no host synchronisation) and is joined
    #: to the caller's stream before a result is handed out.  Same kernels, same inputs, same bits.
    #: The raster launch is PERSISTENT (its workgroups stay on their CUs until the last image is out), so metric kernels enqueued after it only
    #: get the slots they grab in the first microseconds (round 3: 7.51 ms per step beside, 7.30 behind).  Hence the metrics a loop asked for
    #: after its previous render are enqueued on the side stream right BEFORE the next raster launch, from inside `render` (`_mark_fork`):
    #: they take the CUs they need first, the surplus workgroups of the persistent launch start a few microseconds late and still find work,
    #: and `compute_*` hands the finished result out.  A metric that was not foreseen runs on the side stream after the launch, as in round 3,
    #: and is foreseen from then on; one that is no longer asked for is dropped after one step.
    overlap_infractions = False
    _side_streams: Dict[int, Any] = {}
    _side_priority = -1          # HIP stream priority of the side stream (-1: high)
    #: overlap_infractions = 'reserved' (round 4): the raster launch runs on a stream that is kept off `_reserved_per_xcd` CUs per XCD, the metrics on a
    #: stream confined to exactly those (`_ops.reserved_streams`, hipExtStreamCreateWithCUMask).  The write-bound launch loses nothing on 224 of 256
    #: CUs (tools/cu_mask_probe.hip) and the metrics -- served slowly beside the saturated write stream, but served -- finish well inside its 7 ms.
    _reserved_per_xcd = 4

    def _fork_sources(self):
        return [self.kinematic_model.get_state(), self.present_mask, self.agent_size, self.agent_type]

    def _side_stream(self, device):
        if self.overlap_infractions == 'reserved':
            return _ops.reserved_streams(device, Simulator._reserved_per_xcd)[1]          # confined to the CUs the raster launch is kept off
        idx = device.index if device.index is not None else torch.cuda.current_device()
        side = Simulator._side_streams.get(idx)
        if side is None:
            side = Simulator._side_streams[idx] = torch.cuda.Stream(device=device, priority=Simulator._side_priority)
        return side

    def raster_stream(self):
        """The stream that is kept off the reserved CUs (overlap_infractions = 'reserved'): a loop that runs under
        `with torch.cuda.stream(sim.raster_stream()):` renders on it without a detour, and its metrics run beside on the reserved CUs."""
        device = self.kinematic_model.get_state().device
        if not self._reserved_usable(device):
            return torch.cuda.current_stream(device)                # (warned once): the loop then simply runs on the caller's stream
        return _ops.reserved_streams(device, Simulator._reserved_per_xcd)[0]

    _reserved_warned = False

    def _reserved_usable(self, device) -> bool:
        """overlap_infractions = 'reserved' rests on what the bits of a CU mask stand for, which was OBSERVED on an MI355X in SPX mode
        (tools/cu_mask_probe.hip).  _ops.reserved_layout_ok launches a probe on both masked streams once per device and checks that the
        metric stream really runs on `_reserved_per_xcd` CUs of every XCD and the raster stream on all the others; on any other layout, or a
        device with fewer than 64 CUs (a CPX partition), the mode falls back to overlap_infractions = False with ONE warning -- it neither
        raises out of render() nor lets the "reserved" stream silently share the raster launch's CUs."""
        ok, why = _ops.reserved_layout_ok(device, Simulator._reserved_per_xcd) if device.type == 'cuda' else (False, 'not a GPU tensor')
        if not ok and not Simulator._reserved_warned:
            Simulator._reserved_warned = True
            import warnings
            warnings.warn(f"overlap_infractions='reserved' is not usable on {device} ({why}): the infraction metrics run behind the raster "
                          f"launch, as with overlap_infractions=False")
        return ok

    def _metric_fn(self, key):
        name = key[0]
        if name == 'collision':
            return lambda: self._compute_collision(None if key[1] is None else list(key[1]))
        return {'offroad': self._compute_offroad, 'wrong_way': self._compute_wrong_way}[name]

    def _mark_fork(self, write_bound: bool = True) -> None:
        """called by render() right before the raster launch: everything the metrics read has been enqueued by now.
        `write_bound`: the launch that follows is bound by the HBM write stream -- float32 images above 208 x 208.  (That is a statement about the
        BOUND, not about which kernel serves the launch: the fused persistent kernel starts above 160 x 160, raster.hip: split_serves, but a
        float32 launch between 160 and 208 pixels is still limited by instruction issue -- 192 x 192: 4.97 ms for 29 GB = 0.73 of the roof.)
        Only then does 'reserved' pay -- a compute-bound launch (uint8, low resolutions) would give an eighth of its CUs away for nothing, so
        the mode is skipped for it (the metrics run behind the launch)."""
        prev, self._fork = self._fork, None
        wanted, self._fork_used = getattr(self, '_fork_used', None) or [], []
        state = self.kinematic_model.get_state()
        if prev is not None:
            # a foreseen metric the loop did not ask for again was never joined: its kernels read state / present / [sin, cos] that the
            # caller's stream is about to overwrite or free.  Join them here -- the launch they ran beside is long over, the wait is free
            # (and a captured graph has no dangling fork).
            for _, done in prev[3].values():
                prev[2].wait_event(done)
        if not self.overlap_infractions or not state.is_cuda or self.npc_count > 0:
            return
        if self.overlap_infractions == 'reserved' and (not write_bound or not self._reserved_usable(state.device)):
            return
        self._heading_sc()                                        # the shared [sin, cos] exists before the fork
        srcs = self._fork_sources()
        stream = torch.cuda.current_stream(state.device)
        ev = torch.cuda.Event()
        ev.record(stream)
        ready = {}
        if wanted:
            # the metrics of the previous step, ahead of the raster launch on the side stream
            side = self._side_stream(state.device)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for key in wanted:
                    out = self._metric_fn(key)()
                    out.record_stream(stream)                        # allocated in the side stream's pool, consumed on the caller's stream
                    ready[key] = out
            done = torch.cuda.Event()
            done.record(side)
            ready = {k: (v, done) for k, v in ready.items()}
        self._fork = (ev, [(t, t._version) for t in srcs], stream, ready)

    def _beside_render(self, fn, key=None):
        """fn() -> Tensor: the result computed ahead of the raster launch when `key` was foreseen, else fn() on the side stream when a
        render of exactly this state is in flight on the current stream, else in place.
        With overlap_infractions on, asking for the same metric twice between two renders returns the SAME tensor object (the result is
        computed once per state): treat the returned tensors as read-only, or clone them before an in-place edit."""
        fork = self._fork
        if fork is None or not self.overlap_infractions:
            return fn()
        ev, stamp, main, ready = fork
        srcs = self._fork_sources()
        state = srcs[0]
        cached = getattr(self, '_sc_cache', None)
        if len(srcs) != len(stamp) or any(a is not b or a._version != v for a, (b, v) in zip(srcs, stamp)) or self.npc_count > 0 or \
                torch.cuda.current_stream(state.device) != main or cached is None or cached[0] is not state or cached[1] != state._version:
            return fn()
        if key is not None and key not in self._fork_used:
            self._fork_used.append(key)                              # foreseen at the next render
        if key is not None and key in ready:
            out, done = ready[key]
            main.wait_event(done)
            return out
        side = self._side_stream(state.device)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            out = fn()
        out.record_stream(main)                                  # allocated in the side stream's pool, consumed on the caller's stream
        done = torch.cuda.Event()
        done.record(side)
        main.wait_event(done)
        if key is not None:
            ready[key] = (out, done)                             # asked for again before the next render: the same tensor
        return out

    # ------------------------------------------------------------------------------------------------- device scene data
    def _scene(self):
        """Static maps (one per distinct road mesh in the batch), actor templates and packed actor keys; rebuilt after
        `to` / `extend` / `select_batch_elements` or when sizes / types tensors are replaced."""
        gen = self.birdview_mesh_generator
        # stamped on the IDENTITY (and version) of the long-lived tensors the cache is derived from -- not on the torch.cat temporaries
        # of get_all_agent_size() / get_all_agent_type(), whose addresses depend on the allocator.  The cache keeps the sources alive.
        sources = [self.agent_size, self.agent_type, gen.background_mesh.verts, gen.background_mesh.faces, gen.background_mesh.attrs]
        if self.npc_count > 0:
            sources += [self.get_npc_size(), self.get_npc_types()]
        for v in (self.traffic_controls or {}).values():
            sources += [v.pos, v.mask]
        extra = (self.batch_size, self.npc_count, tuple((self.traffic_controls or {}).keys()))
        c = self._scene_cache
        if c is not None and c['extra'] == extra and len(c['sources']) == len(sources) and \
                all(a is b and a._version == ver for a, b, ver in zip(sources, c['sources'], c['versions'])):
            return c
        sizes, types = self.get_all_agent_size(), self.get_all_agent_type()
        if not isinstance(self.renderer, HipRenderer):
            raise RuntimeError(f'{type(self.renderer).__name__} cannot take the fused scene path; use HipRenderer')
        dev = sizes.device
        bg = gen.background_mesh                                   # RGBMesh, batch B, (x, y, z) + colour per vertex
        names = list(self.agent_types)
        lv, cm = self.renderer.rendering_levels, self.renderer.color_map
        actor_levels = [float(lv[n]) for n in names] + [float(lv['direction']), float(lv['goal_waypoint'])]
        B = self.batch_size
        # traffic controls are drawn like actors without a direction triangle: one quad per stop line (mesh.py:1007-1035)
        controls = self.traffic_controls or {}
        ctrl_kinds = [k for k in controls if controls[k].pos.shape[1] > 0]
        for kind in ctrl_kinds:
            cats = [f'{kind}_{st}' for st in controls[kind].allowed_states] if kind == 'traffic_light' else [kind]
            actor_levels += [float(lv[c]) for c in cats]
        # one device map per DISTINCT mesh of the batch (a collated batch of 512 x Town01 + 512 x Town02 builds two), shared through the
        # process-wide content cache: `copy`, `select_batch_elements`, `extend`, `to` and `shard_simulator` end up here again and find the handles
        maps = [(self.renderer.scene_maps(bg, actor_levels, device=dev), None)]
        tmpl = actor_template(sizes.detach()).contiguous()          # B x N x 7 x 2 (a render that differentiates the sizes builds its own)
        keys, key_tables, wp_keys = [], [], []
        for smap, _ in maps:
            body = torch.tensor([(smap.rank_of(lv[n]) << 24) | int(_ops.quantise_colors(torch.tensor(cm[n], dtype=torch.float32) / 255.0)) for n in names],
                                dtype=torch.int64, device=dev)
            dkey = (smap.rank_of(lv['direction']) << 24) | int(_ops.quantise_colors(torch.tensor(cm['direction'], dtype=torch.float32) / 255.0))
            k = torch.stack([body[types.long()], torch.full_like(types.long(), dkey)], dim=-1)
            keys.append(k.to(torch.int32).contiguous())             # bit pattern of the uint32 key
            key_tables.append(sorted(set(body.tolist()) | {dkey}))   # distinct actor keys, known on the host (bit-plane kernel)
            wp_keys.append((smap.rank_of(lv['goal_waypoint']) << 24) | int(_ops.quantise_colors(torch.tensor(cm['goal_waypoint'], dtype=torch.float32) / 255.0)))
        ctrl = None
        if ctrl_kinds:
            q = lambda name: int(_ops.quantise_colors(torch.tensor(cm[name], dtype=torch.float32) / 255.0))
            st_q, tm_q, static_keys, light_tables = [], [], [], []
            for kind in ctrl_kinds:
                c = controls[kind]
                pos, m = c.pos.to(dev).to(sizes.dtype), c.mask.to(dev)[..., None]
                # padding elements sit at (-1000, -1000) with all four corners there (traffic_controls.py:31-33)
                xy = torch.where(m, pos[..., :2], torch.full_like(pos[..., :2], -1000.0))
                st_q.append(torch.cat([xy, torch.where(m, pos[..., 4:5], torch.zeros_like(pos[..., 4:5])), torch.zeros_like(pos[..., :1])], dim=-1))
                sx = torch.tensor([0.5, -0.5, -0.5, 0.5], dtype=pos.dtype, device=dev) * pos[..., 2:3]      # box2corners_th corner order
                sy = torch.tensor([0.5, 0.5, -0.5, -0.5], dtype=pos.dtype, device=dev) * pos[..., 3:4]
                quad = torch.stack([sx, sy], dim=-1) * m[..., None].to(pos.dtype)
                tm_q.append(torch.cat([quad, torch.zeros(quad.shape[:2] + (3, 2), dtype=pos.dtype, device=dev)], dim=-2))
                per_map = []
                for smap, _ in maps:
                    if kind == 'traffic_light':
                        per_map.append(torch.tensor([(smap.rank_of(lv[f'{kind}_{s_}']) << 24) | q(f'{kind}_{s_}') for s_ in c.allowed_states],
                                                    dtype=torch.int64, device=dev))
                    else:
                        per_map.append(torch.full((1,), (smap.rank_of(lv[kind]) << 24) | q(kind), dtype=torch.int64, device=dev))
                static_keys.append(per_map)
            ctrl = dict(kinds=ctrl_kinds, state=torch.cat(st_q, dim=1).contiguous(), tmpl=torch.cat(tm_q, dim=1).contiguous(), key_lut=static_keys)
            for i in range(len(maps)):
                key_tables[i] = sorted(set(key_tables[i]) | {int(v) for per_map in static_keys for v in per_map[i].tolist()})
        self._scene_cache = dict(sources=sources, versions=[t._version for t in sources], extra=extra, maps=maps, tmpl=tmpl, keys=keys, key_tables=key_tables, ctrl=ctrl, wp_keys=wp_keys)
        return self._scene_cache

    def _waypoint_triangles(self, waypoints: Tensor, rendering_mask: Optional[Tensor]):
        """B x Nc x M waypoints -> the world-space triangles generate() would add per camera (mesh.py:1120-1145): B x Nc x M*T x 3 x 2.
        The faces of a masked waypoint are zeroed there and so alias the first waypoint vertex of the camera -- a dot at the centre of
        waypoint 0, reproduced here by collapsing the triangle onto that point."""
        disc = self.birdview_mesh_generator.waypoint_mesh
        B, Nc, M = waypoints.shape[:3]
        corner = torch.gather(disc.verts[..., :2].unsqueeze(1).expand(-1, disc.faces.shape[1], -1, -1), 2,
                              disc.faces.long()[..., None].expand(-1, -1, -1, 2))                       # B x T x 3 x 2
        tri = corner[:, None, None].to(waypoints.dtype) + waypoints[..., None, None, :]                     # B x Nc x M x T x 3 x 2
        if rendering_mask is not None:
            first = (disc.verts[:, 0, :2][:, None].to(waypoints.dtype) + waypoints[:, :, 0])[:, :, None, None, None]       # B x Nc x 1 x 1 x 1 x 2
            tri = torch.where(rendering_mask.to(torch.bool)[..., None, None, None], tri, first.expand_as(tri))
        return tri.reshape(B, Nc, M * corner.shape[1], 3, 2).contiguous(), None

    def _control_keys(self, scene, i_map: int, sl) -> Tensor:
        """(b, Nq, 2) int32 keys of the control quads of the scenes in `sl`: (colour of the current state, 0 = no direction part)"""
        out = []
        for kind, per_map in zip(scene['ctrl']['kinds'], scene['ctrl']['key_lut']):
            c = self.traffic_controls[kind]
            lut = per_map[i_map]
            idx = c.state[sl].to(lut.device).long() if kind == 'traffic_light' else torch.zeros_like(c.state[sl].to(lut.device).long())
            out.append(lut[idx])
        body = torch.cat(out, dim=1)
        return torch.stack([body, torch.zeros_like(body)], dim=-1).to(torch.int32)

    # ------------------------------------------------------------------------------------------------- rendering
    def _noisy_scene_sources(self):
        """(mesh generator, traffic controls) of a noisy-perception frame (simulator.py:951-978)"""
        from torchdrivesim_amd.mesh import BaseMesh
        from torchdrivesim_amd.utils import rotate
        gen = self.birdview_mesh_generator.copy()
        gen.background_mesh = self.get_noisy_background_mesh()
        noisy_lf = self.get_noisy_lane_features()
        if noisy_lf is not None and noisy_lf.dense_lane_features is not None:
            markers, markers_mask = noisy_lf.dense_lane_features, noisy_lf.dense_lane_features_mask      # B x M x [x, y, psi, width], B x M
            if markers_mask is None:
                markers_mask = torch.ones_like(markers[..., 0], dtype=torch.bool)
            n_markers = markers.shape[-2]
            width = markers[..., 3]
            zero, one = torch.zeros_like(width), torch.ones_like(width)
            arrow = torch.stack([torch.stack([zero, -width / 2], dim=-1), torch.stack([zero, width / 2], dim=-1), torch.stack([one, zero], dim=-1)], dim=-2)
            verts = rotate(arrow, markers[..., None, 2:3]) + markers[..., None, :2]                    # one metre long, pointing along psi
            verts = torch.where(markers_mask[..., None, None], verts, torch.zeros_like(verts))
            faces = torch.tensor([[0, 1, 2]], dtype=torch.long, device=markers.device) + 3 * torch.arange(n_markers, device=markers.device)[:, None]
            dense = BirdviewMesh.set_properties(BaseMesh(verts=verts.flatten(-3, -2), faces=faces.expand_as(verts[..., 0])), category='stop_sign')
            gen.add_static_meshes([dense])
        controls = self.get_noisy_traffic_controls()
        if controls is not None:
            gen.initialize_traffic_controls_mesh(controls)
        return gen, controls

    def render(self, camera_xy: Tensor, camera_psi: Tensor, res: Optional[Resolution] = None, rendering_mask: Optional[Tensor] = None,
               fov: Optional[float] = None, waypoints: Optional[Tensor] = None, waypoints_rendering_mask: Optional[Tensor] = None,
               custom_agent_colors: Optional[Tensor] = None, noisy_perception: bool = False, _camera_sc: Optional[Tensor] = None,
               out: Optional[Tensor] = None, _ego: bool = False) -> Tensor:
        """Bird's-eye images for BxNx2 camera positions and BxNx1 headings -> BxNx3xHxW (simulator.py:920-992).
        `out` (not in the reference, which allocates per call, rendering/cv2.py:52): a caller-owned contiguous BxNx3xHxW tensor of the
        renderer's output dtype to render into; returned.  HipRenderer only, one static map per batch, not for differentiable calls."""
        if noisy_perception:
            # what the policy is shown instead of the truth (simulator.py:951-978): the observation model's background, its lane markers
            # drawn as triangles, its traffic controls.  The scene of the ordinary path is swapped for the duration of the call; its
            # static map is rebuilt from the noisy background (a map upload per call: this is the slow lane of the renderer).
            gen, controls = self._noisy_scene_sources()
            saved = (self.birdview_mesh_generator, self.traffic_controls, self._scene_cache)
            self.birdview_mesh_generator, self.traffic_controls, self._scene_cache = gen, controls, None
            try:
                return self.render(camera_xy, camera_psi, res=res, rendering_mask=rendering_mask, fov=fov, waypoints=waypoints,
                                   waypoints_rendering_mask=waypoints_rendering_mask, custom_agent_colors=custom_agent_colors,
                                   noisy_perception=False, _camera_sc=_camera_sc, out=out, _ego=_ego)
            finally:
                self.birdview_mesh_generator, self.traffic_controls, self._scene_cache = saved
        camera_sc = _camera_sc if _camera_sc is not None else torch.cat([torch.sin(camera_psi), torch.cos(camera_psi)], dim=-1)
        if camera_xy.dim() == 2:
            camera_xy, camera_sc = camera_xy.unsqueeze(1), camera_sc.unsqueeze(1)
        n_cam = camera_xy.shape[-2]
        present = self.get_all_agent_present_mask()
        mask = present.unsqueeze(-2).expand(present.shape[:-1] + (n_cam,) + present.shape[-1:])
        if rendering_mask is not None:
            mask = mask.logical_and(rendering_mask.to(torch.bool))
        if isinstance(self.renderer, HipRenderer):
            scene = self._scene()
            state = self.get_all_agent_state()
            # gradients through the rasteriser (K3 backward, build-defined) only when someone asks for them
            sizes = self.get_all_agent_size()
            size_grad = torch.is_grad_enabled() and sizes.requires_grad and self.renderer.out_dtype == torch.float32
            diff = torch.is_grad_enabled() and (state.requires_grad or camera_xy.requires_grad or camera_sc.requires_grad or size_grad) and \
                self.renderer.out_dtype == torch.float32
            if not diff:
                state, camera_xy, camera_sc = state.detach(), camera_xy.detach(), camera_sc.detach()
            # gradients with respect to the agents' length and width flow through the template vertices (actor_template is plain torch)
            tmpl_all = actor_template(sizes).contiguous() if size_grad else scene['tmpl']
            ctrl = scene['ctrl']
            if ctrl is not None:                                    # stop lines ride along as extra quads
                state = torch.cat([state, ctrl['state'].to(state.dtype)], dim=1)
                tmpl_all = torch.cat([tmpl_all, ctrl['tmpl'].to(tmpl_all.dtype)], dim=1)
                mask = torch.cat([mask, torch.ones(mask.shape[:-1] + (ctrl['state'].shape[1],), dtype=torch.bool, device=mask.device)], dim=-1)
            agent_sc = self._heading_sc() if ctrl is None else _ops.heading_sc(state[..., 2])
            wp_tri = wp_on = None
            if waypoints is not None and waypoints.shape[2] > 0:
                wp_tri, wp_on = self._waypoint_triangles(waypoints.to(state.dtype), waypoints_rendering_mask)
            if out is not None and len(scene['maps']) != 1:
                raise RuntimeError('`out=` needs a batch that is served by one launch')
            out_arg, out = ({} if out is None else dict(out=out)), []
            # the metrics run beside the launch; a differentiable render forks too when the launch stays on the caller's stream (the plain second
            # stream, or 'reserved' with the loop on sim.raster_stream()): the metric nodes are then autograd nodes of the side stream, and the
            # engine runs their backward there as well -- beside the rasteriser's backward
            if not diff or self.overlap_infractions is True or \
                    (self.overlap_infractions == 'reserved' and state.is_cuda and self._reserved_usable(state.device) and
                     torch.cuda.current_stream(state.device) == _ops.reserved_streams(state.device, Simulator._reserved_per_xcd)[0]):
                r = res if res is not None else getattr(self.renderer, 'res', None)
                self._mark_fork(write_bound=self.renderer.out_dtype == torch.float32 and (r is None or min(r.height, r.width) > 208))
            # render_egocentric with gradients: the cameras are the exposed agents themselves -- one autograd node takes state and headings and
            # folds the cameras' gradient into the agents' (no slice nodes for camera_xy / camera_sc in the graph)
            ego_n = n_cam if (_ego and diff and ctrl is None and len(scene['maps']) == 1 and n_cam <= state.shape[1]) else 0
            for i_map, ((smap, b), keys, ktab) in enumerate(zip(scene['maps'], scene['keys'], scene['key_tables'])):
                sl = slice(None) if b is None else slice(b, b + 1)
                cut = (lambda t: t) if b is None else (lambda t: t[sl])       # (a full slice would still be a node of the autograd graph)
                k = keys[sl]
                if custom_agent_colors is not None:
                    # generate() paints the four body vertices of agent a with custom_agent_colors[b, c, a] for camera c
                    # (mesh.py:1092-1099); the direction triangle keeps its colour.  Same rendering level, so only the colour
                    # bits of the body key change -- per camera.
                    rgb = _ops.quantise_colors(custom_agent_colors[sl].to(state.device)).to(torch.int32)          # (b,Nc,A,)
                    kc = k[:, None].expand(-1, n_cam, -1, -1).clone()
                    kc[..., 0] = (kc[..., 0] & ~0xFFFFFF) | rgb
                    k, ktab = kc.contiguous(), None
                if ctrl is not None:
                    kq = self._control_keys(scene, i_map, sl)
                    k = torch.cat([k, kq[:, None].expand(-1, n_cam, -1, -1) if k.dim() == 4 else kq], dim=-2).contiguous()
                extra = dict()
                if wp_tri is not None:
                    wk = scene['wp_keys'][i_map]
                    extra = dict(extra_tri=wp_tri[sl], extra_key=torch.full(wp_tri[sl].shape[:3], wk, dtype=torch.int32, device=state.device))
                    ktab = None if ktab is None else sorted(set(ktab) | {wk})
                launch = lambda: self.renderer.render_scene(smap, cut(state), cut(agent_sc), cut(tmpl_all), k, cut(mask).contiguous(),     # noqa: E731
                                                            cut(camera_xy), cut(camera_sc), res=res, fov=fov, key_table=ktab, differentiable=diff, **extra,
                                                            **out_arg, **(dict(ego_cameras=ego_n) if ego_n else {}))
                if self.overlap_infractions == 'reserved' and not diff and self._fork is not None and state.is_cuda and \
                        self._fork[2] != _ops.reserved_streams(state.device, Simulator._reserved_per_xcd)[0]:
                    # the launch goes to the stream that is kept off the reserved CUs: it waits for what the caller's stream has enqueued so far
                    # (the fork event) and the caller's stream waits for it -- beside it, on the reserved CUs, the foreseen metrics are running.
                    # (A loop that makes that stream its current one -- `with torch.cuda.stream(sim.raster_stream()):` -- saves these two waits.)
                    main = self._fork[2]
                    rs = _ops.reserved_streams(state.device, Simulator._reserved_per_xcd)[0]
                    # NOT the fork event: the per-launch inputs (per-camera colour keys, traffic-control keys, waypoint triangles and their
                    # keys) were built on the caller's stream AFTER it -- the launch waits for an event recorded here, behind all of them
                    built = torch.cuda.Event()
                    built.record(main)
                    rs.wait_event(built)
                    with torch.cuda.stream(rs):
                        img = launch()
                    img.record_stream(main)
                    done = torch.cuda.Event()
                    done.record(rs)
                    main.wait_event(done)
                    out.append(img)
                else:
                    out.append(launch())
            return out[0] if len(out) == 1 else torch.cat(out, dim=0)
        if out is not None:
            raise RuntimeError(f'`out=` is served by HipRenderer only, not by {type(self.renderer).__name__}')
        # any other BirdviewRenderer: the reference's generic dataflow (explicit per-camera mesh)
        rgb_mesh = self.birdview_mesh_generator.generate(n_cam, agent_state=self.get_all_agent_state()[:, None].expand(-1, n_cam, -1, -1),
                                                         present_mask=mask, custom_agent_colors=custom_agent_colors,
                                                         waypoints=waypoints, waypoints_rendering_mask=waypoints_rendering_mask,
                                                         traffic_lights=self.traffic_controls['traffic_light'].extend(n_cam, in_place=False)
                                                         if self.traffic_controls and 'traffic_light' in self.traffic_controls else None)
        img = self.renderer.render_frame(rgb_mesh, camera_xy, camera_sc, res=res, fov=fov)
        return img.reshape((self.batch_size, n_cam) + img.shape[1:])

    def render_egocentric(self, ego_rotate: bool = True, res: Optional[Resolution] = None, fov: Optional[float] = None,
                          visibility_matrix: Optional[Tensor] = None, custom_agent_colors: Optional[Tensor] = None,
                          n_subsequent_waypoints: int = 1, noisy_perception: bool = False, out: Optional[Tensor] = None) -> Tensor:
        """One camera per exposed agent -> BxAx3xHxW (simulator.py:994-1033).  `out`: see `render`."""
        state = self.get_state()
        camera_xy, camera_psi = state[..., :2], state[..., 2:3]
        if not ego_rotate:
            camera_psi = torch.ones_like(camera_psi) * (np.pi / 2)
        rendering_mask = visibility_matrix
        if self.cfg.single_agent_rendering:
            # the reference builds eye(2) here regardless of A; the intended eye(A) over all agents is used
            A, total = self.agent_count, self.agent_count + self.npc_count
            rendering_mask = torch.eye(A, total, dtype=torch.bool, device=state.device).unsqueeze(0).expand(self.batch_size, -1, -1)
        cam_sc = None
        if ego_rotate:
            cam_sc = self._heading_sc()                                     # the cameras ARE the exposed agents (differentiable when the state is)
            if cam_sc.shape[-2] != self.agent_count:
                cam_sc = cam_sc[..., :self.agent_count, :]
        waypoints = self.get_waypoints(count=n_subsequent_waypoints)                  # simulator.py:1013-1017
        waypoints_mask = self.get_waypoints_mask(count=n_subsequent_waypoints) if waypoints is not None else None
        return self.render(camera_xy, camera_psi, rendering_mask=rendering_mask, res=res, fov=fov, custom_agent_colors=custom_agent_colors,
                           waypoints=waypoints, waypoints_rendering_mask=waypoints_mask, noisy_perception=noisy_perception, _camera_sc=cam_sc,
                           out=out, _ego=ego_rotate)

    # ------------------------------------------------------------------------------------------------- infractions
    def compute_offroad(self) -> Tensor:
        """BxA off-road loss = thresholded squared corner-to-mesh distance x present (simulator.py:1035-1044).
        The whole road_mesh is the driving surface, lane markings included."""
        return self._beside_render(self._compute_offroad, ('offroad',))

    def _compute_offroad(self) -> Tensor:
        state = self.get_state()
        if self.agent_count == 0 or self.road_mesh.faces_count == 0:
            return torch.zeros_like(state[..., 0])
        from torchdrivesim_amd.infractions import _static_maps_for
        maps = _static_maps_for(self.road_mesh, state.device)      # geometry-only device map(s), cached on the mesh
        size, present = self.get_agent_size(), self.get_present_mask()
        sc_all = self._heading_sc()
        if sc_all.shape[-2] != self.agent_count:
            sc_all = sc_all[..., :self.agent_count, :]
        out = []
        for smap, b in maps:
            sl = slice(None) if b is None else slice(b, b + 1)
            cut = (lambda t: t) if b is None else (lambda t: t[sl])           # (a full slice would still be a node of the autograd graph)
            out.append(_ops.offroad(smap, cut(state), cut(size), threshold=self.cfg.offroad_threshold, present=cut(present),
                                    sc=None if sc_all is None else cut(sc_all)))
        return out[0] if len(out) == 1 else torch.cat(out, dim=0)

    def compute_wrong_way(self) -> Tensor:
        """Wrong-way metric per agent, -cos of the angle between the agent and the lane it is on where that angle exceeds
        `cfg.wrong_way_angle_threshold` (simulator.py:607-630 -> infractions.lanelet_orientation_loss), times the present mask.
        Zeros without a lanelet map, as the reference.  `lanelet_map`: a list of B `lanelet2.LaneletMap` or None."""
        return self._beside_render(self._compute_wrong_way, ('wrong_way',))

    def _compute_wrong_way(self) -> Tensor:
        state = self.get_state()
        if self.lanelet_map is None or all(m is None for m in self.lanelet_map):
            return torch.zeros(state.shape[0], state.shape[1], device=state.device)
        tol = self.cfg.lanelet_inclusion_tolerance
        if self._lane_set is None or self._lane_set[1] < tol or self._lane_set[0].device != state.device:
            self._lane_set = (lane_table_set(self.lanelet_map, state.device, tol), tol)
        assert self.cfg.wrong_way_angle_threshold >= np.pi / 2, 'direction_angle_threshold smaller than pi / 2 will produce false positives'
        return _ops.wrong_way(self._lane_set[0], state, self.recenter_offset, self.get_present_mask(), self.cfg.wrong_way_angle_threshold, tol)

    def compute_traffic_lights_violations(self) -> Tensor:
        """BxA: the agent is (mostly) past the stop line of a red light (simulator.py:1046-1062)."""
        state = self.get_state()
        controls = self.get_traffic_controls()
        if controls is not None and 'traffic_light' in controls:
            boxes = torch.cat([state[..., :2], self.get_agent_size()[..., :2], state[..., 2:3]], dim=-1)
            return controls['traffic_light'].compute_violation(boxes) * self.get_present_mask().to(state.dtype)
        return torch.zeros(state.shape[0], state.shape[1], dtype=torch.bool, device=state.device)

    def _all_boxes(self):
        return _ops.state_boxes(self.get_all_agent_state(), self.get_all_agent_size())

    def _collision_mask(self, agent_types: Optional[List[str]]):
        mask = self.get_all_agent_present_mask()
        if agent_types is not None:
            allowed = torch.tensor([self.agent_types.index(t) for t in agent_types if t in self.agent_types], device=mask.device)
            mask = mask.logical_and(torch.isin(self.get_all_agent_type(), allowed))
        return mask

    def compute_collision(self, agent_types: Optional[List[str]] = None) -> Tensor:
        """BxA collision metric of the exposed agents against ALL agents (simulator.py:1161-1194).  For `iou` / `discs`:
        collision_i = sum_j o_ij present_j - max_j o_ij present_j, self overlap assumed to be the max."""
        return self._beside_render(lambda: self._compute_collision(agent_types), ('collision', None if agent_types is None else tuple(agent_types)))

    def _compute_collision(self, agent_types: Optional[List[str]] = None) -> Tensor:
        metric = self.cfg.collision_metric
        A = self.agent_count
        if A == 0:
            return torch.zeros_like(self.get_state()[..., 0])
        if metric in (CollisionMetric.iou, CollisionMetric.discs):
            boxes = self._all_boxes()
            # a NaN heading is scrubbed inside the kernel, so the shared [sin, cos] of the raw headings serves the IoU metric
            sc = self._heading_sc() if metric == CollisionMetric.iou else None
            return _ops.collision(boxes, self._collision_mask(agent_types), n_exposed=A, metric=metric.value, sc=sc)
        if metric == CollisionMetric.nograd:
            assert agent_types is None, 'The argument `agent_types` is not supported by the selected collision metric.'
            # count of other present exposed agents whose rectangle shares area with the agent's (simulator.py:1111-1149 ->
            # infractions.py:352-375, where shapely answers `intersection(...).area != 0` on the host): one kernel, exact predicate in
            # float64 on the float32 corners of infractions.rectangle_vertices; NPCs are not counted, as in the reference (:1124)
            state, size, present = self.get_state().detach(), self.get_agent_size(), self.get_present_mask()
            boxes = torch.cat([state[..., :2], size, state[..., 2:3]], dim=-1)
            return _ops.overlap_count(boxes, present)
        raise ValueError('Unrecognized collision metric: ' + str(metric))
