"""
What each exposed agent perceives of the others (torchdrivesim/observation_noise.py): the noise-free base model and the "standard
sensing" model -- position noise that grows with distance and an occlusion test of every sight line against every other entity, which
is the O(A E^2) part and runs as a HIP kernel (tds_occlusion_mask_f32).  The map-related hooks of the reference (noisy lane features,
noisy background meshes from logs) depend on Lanelet2 data and are outside the scope of this framework.
"""
from dataclasses import dataclass

import torch
from torch import Tensor

from torchdrivesim_amd import _ops


@dataclass
class ObservationNoiseConfig:
    _type_: str = 'base'


@dataclass
class StandardSensingObservationNoiseConfig:
    _type_: str = 'standard_sensing'


def _per_ego(x: Tensor, n_ego: int, dim: int) -> Tensor:
    """insert an ego axis after the batch axis: (B, E, ...) -> (B, A, E, ...) view"""
    return x.unsqueeze(1).expand((x.shape[0], n_ego) + tuple(x.shape[1:]))


class ObservationNoise:
    """Every exposed agent sees the true state, size and presence of all agents (observation_noise.py:33-69)."""

    def __init__(self, cfg: ObservationNoiseConfig = None):
        self.cfg = cfg if cfg is not None else ObservationNoiseConfig()

    def get_noisy_state(self, simulator) -> Tensor:          # B x A x (A+Npc) x 4
        return _per_ego(simulator.get_all_agent_state(), simulator.agent_count, 1)

    def get_noisy_present_mask(self, simulator) -> Tensor:   # B x A x (A+Npc)
        return _per_ego(simulator.get_all_agent_present_mask(), simulator.agent_count, 1)

    def get_noisy_agent_size(self, simulator) -> Tensor:     # B x A x (A+Npc) x 2
        return _per_ego(simulator.get_all_agent_size(), simulator.agent_count, 1)

    def get_noisy_traffic_controls(self, simulator):
        return simulator.traffic_controls

    def get_noisy_road_mesh(self, simulator):
        return simulator.road_mesh

    def get_noisy_background_mesh(self, simulator):
        return simulator.birdview_mesh_generator.background_mesh

    def get_noisy_lane_features(self, simulator):
        return simulator.lane_features


class StandardSensingObservationNoise(ObservationNoise):
    """Gaussian position / heading / speed noise whose standard deviation steps up with the distance from the observer, and entities
    hidden behind others (observation_noise.py:72-132)."""
    #: (distance above which it applies, standard deviation), observation_noise.py:84-89
    DEVIATION_STEPS = ((0.5, 0.19), (25.0, 1.6), (50.0, 3.2), (100.0, 3.83))

    def __init__(self, cfg: StandardSensingObservationNoiseConfig = None):
        super().__init__(cfg if cfg is not None else StandardSensingObservationNoiseConfig())

    def deviation(self, simulator) -> Tensor:
        """B x A x (A+Npc) x 1 standard deviation of the noise each ego sees on each entity"""
        ego_xy = simulator.get_state()[..., :2]
        all_xy = simulator.get_all_agent_state()[..., :2]
        dist = torch.norm(ego_xy[..., None, :] - all_xy[:, None], dim=-1)
        dev = torch.zeros_like(dist)
        for thr, sd in self.DEVIATION_STEPS:
            dev = torch.maximum(dev, sd * (dist > thr).to(dist.dtype))
        return dev.unsqueeze(-1)

    def get_noisy_state(self, simulator) -> Tensor:
        base = super().get_noisy_state(simulator)
        return base + torch.randn_like(base) * self.deviation(simulator)

    def get_noisy_present_mask(self, simulator) -> Tensor:
        return _ops.occlusion_mask(simulator.get_all_agent_state(), simulator.get_all_agent_size(), simulator.get_all_agent_present_mask(),
                                   simulator.agent_count)


class MapObservationNoiseFromLog(ObservationNoise):
    """Map-side observations replayed from a log, one entry per simulation step, falling back to the simulator's own once the log is
    exhausted (observation_noise.py:135-178).  Accessors only: rendering the logged backgrounds (`noisy_perception=True`) would swap the
    static map every step and stays outside this framework."""

    def __init__(self, cfg=None, noisy_lane_features=None, noisy_background_mesh=None, noisy_traffic_controls=None, noisy_crosswalk_features=None):
        super().__init__(cfg)
        self.noisy_lane_features = noisy_lane_features
        self.noisy_background_mesh = noisy_background_mesh
        self.noisy_traffic_controls = noisy_traffic_controls
        self.noisy_crosswalk_features = noisy_crosswalk_features

    @staticmethod
    def _logged(log, simulator):
        return log[simulator.internal_time] if log is not None and simulator.internal_time < len(log) else None

    def get_noisy_lane_features(self, simulator):
        hit = self._logged(self.noisy_lane_features, simulator)
        return hit if hit is not None else simulator.lane_features

    def get_noisy_background_mesh(self, simulator):
        hit = self._logged(self.noisy_background_mesh, simulator)
        if hit is None:
            return simulator.birdview_mesh_generator.background_mesh
        from torchdrivesim_amd.mesh import set_colors_with_defaults
        gen = simulator.birdview_mesh_generator
        return set_colors_with_defaults(hit.clone(), color_map=gen.color_map, rendering_levels=gen.rendering_levels)

    def get_noisy_road_mesh(self, simulator):
        hit = self._logged(self.noisy_background_mesh, simulator)
        return hit if hit is not None else simulator.road_mesh

    def get_noisy_traffic_controls(self, simulator):
        hit = self._logged(self.noisy_traffic_controls, simulator)
        return hit if hit is not None else simulator.traffic_controls

    def get_noisy_crosswalk_features(self, simulator):
        return self._logged(self.noisy_crosswalk_features, simulator)
