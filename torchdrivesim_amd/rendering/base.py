"""
Renderer plugin API with the reference's names (torchdrivesim/rendering/base.py): `RendererConfig`, `Cameras`,
`BirdviewRenderer` (abstract `render_rgb_mesh`), `DummyRenderer`, default colour map and rendering levels.
"""
import abc
import logging
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from torchdrivesim_amd.mesh import RGBMesh
from torchdrivesim_amd.utils import Resolution

logger = logging.getLogger(__name__)


@dataclass
class RendererConfig:
    """Behaviour of the renderer; subclasses select the backend (rendering/base.py:23-34)."""
    backend: str = 'default'
    render_agent_direction: bool = True
    left_handed_coordinates: bool = False
    highlight_ego_vehicle: bool = False
    shift_mesh_by_camera_before_rendering: bool = True
    device: Optional[str] = None


@dataclass
class DummyRendererConfig(RendererConfig):
    backend: str = 'dummy'


class Cameras:
    """Orthographic bird's-eye cameras: position, heading as [sin, cos] and scale = 2 / fov (rendering/base.py:45-130).
    The projection helpers are torch ops for host-side use; the raster kernel applies the same arithmetic in HIP."""

    def __init__(self, xy: Tensor, sc: Tensor, scale: float):
        self.xy, self.sc, self.scale = xy, sc, scale

    def get_camera_center(self) -> Tensor:
        return self.xy

    def _rot(self, points: Tensor) -> Tensor:
        return torch.stack([self.sc.flip(dims=[-1]), self.sc * torch.tensor([-1, 1], device=points.device)], dim=-2)

    def transform_points_screen(self, points: Tensor, res: Resolution) -> Tensor:
        """world (Nc x P x 2) -> pixel coordinates; forward axis of the camera points to decreasing x (base.py:102-115)"""
        p = points - self.xy.unsqueeze(1)
        p = torch.matmul(self._rot(points).unsqueeze(1), p.unsqueeze(-1)).squeeze(-1)
        p = -p * self.scale
        p = p * min(res.height, res.width) / 2
        return p + torch.tensor([res.width, res.height], device=points.device) / 2

    def reverse_transform_points_screen(self, points: Tensor, res: Resolution) -> Tensor:
        p = points - torch.tensor([res.width, res.height], device=points.device) / 2
        p = p / (min(res.height, res.width) / 2)
        p = -p / self.scale
        p = torch.matmul(self._rot(points).unsqueeze(1).transpose(-1, -2), p.unsqueeze(-1)).squeeze(-1)
        return p + self.xy.unsqueeze(1)


class BirdviewRenderer(abc.ABC):
    """2-D bird's-eye renderer over a static background mesh and rectangular agents; square resolutions only
    (rendering/base.py:133-220)."""

    def __init__(self, cfg: RendererConfig, color_map: Optional[Dict[str, Tuple[int, int, int]]] = None,
                 rendering_levels: Optional[Dict[str, float]] = None, res: Resolution = Resolution(64, 64), fov: float = 35):
        self.cfg = cfg
        self.res = res
        self.scale = 2.0 / fov
        self.color_map = color_map if color_map is not None else get_default_color_map()
        self.rendering_levels = rendering_levels if rendering_levels is not None else get_default_rendering_levels()

    def copy(self):
        other = self.__class__(cfg=self.cfg, color_map=self.color_map.copy(), rendering_levels=self.rendering_levels.copy(), res=self.res)
        other.scale = self.scale
        return other

    def get_color(self, element_type: str) -> Tuple[int, int, int]:
        return self.color_map[element_type]

    def render_frame(self, rgb_mesh: RGBMesh, camera_xy: Tensor, camera_sc: Tensor, res: Optional[Resolution] = None,
                     fov: Optional[float] = None) -> Tensor:
        """rgb_mesh already expanded per camera; camera_xy / camera_sc BxNcx2 -> (B*Nc)x3xHxW float in [0,255].
        A RuntimeError from the backend is logged and yields a black image, as in the reference (base.py:190-201)."""
        scale = (2.0 / fov) if fov is not None else self.scale
        n_cam = camera_xy.shape[-2]
        camera_xy, camera_sc = camera_xy.reshape(-1, 2), camera_sc.reshape(-1, 2)
        cameras = self.construct_cameras(camera_xy, camera_sc, scale=scale)
        res = self.res if res is None else res
        try:
            image = self.render_rgb_mesh(rgb_mesh, res, cameras)
        except RuntimeError as e:
            logger.exception(e)
            image = torch.zeros((camera_xy.shape[0] * n_cam, res.height, res.width, 3), device=camera_xy.device)
        return image.reshape(-1, res.height, res.width, 3).permute(0, 3, 1, 2)

    @abc.abstractmethod
    def render_rgb_mesh(self, mesh: RGBMesh, res: Resolution, cameras: Cameras) -> Tensor:
        """-> (B*Nc)xHxWx3 float RGB in [0,255]"""

    def construct_cameras(self, xy: Tensor, sc: Tensor, scale: Optional[float] = None) -> Cameras:
        return Cameras(xy=xy, sc=sc, scale=self.scale if scale is None else scale)


class DummyRenderer(BirdviewRenderer):
    """Black images of the right size, for debugging and benchmarking (rendering/base.py:223-231)."""

    def render_rgb_mesh(self, mesh: RGBMesh, res: Resolution, cameras: Cameras) -> Tensor:
        n = cameras.get_camera_center().shape[0]
        return torch.zeros((n, res.height, res.width, 3), device=mesh.device, dtype=torch.float32)


def get_default_rendering_levels() -> Dict[str, float]:
    """category -> rendering level; lower renders on top (rendering/base.py:234-262, data table)"""
    return dict(direction=2, ego=3, vehicle=4, bicycle=5, pedestrian=6, map_boundary=7, goal_waypoint=8, ground_truth=9,
                prediction=10, traffic_light=11, traffic_light_green=11, traffic_light_yellow=11, traffic_light_red=11,
                stop_sign=11, yield_sign=11, left_lane=12, joint_lane=13, right_lane=14, road=15)


def get_default_color_map() -> Dict[str, Tuple[int, int, int]]:
    """category -> RGB in [0,255] (rendering/base.py:265-292, data table)"""
    return dict(background=(0, 0, 0), road=(155, 155, 155), corridor=(0, 155, 0), ego=(255, 0, 0), vehicle=(32, 74, 135),
                bicycle=(24, 104, 225), pedestrian=(173, 127, 168), ground_truth=(196, 188, 165), prediction=(255, 155, 0),
                left_lane=(80, 127, 86), right_lane=(128, 0, 128), joint_lane=(255, 255, 255), direction=(100, 255, 255),
                rear_lights=(255, 255, 0), map_boundary=(255, 255, 0), traffic_light_green=(81, 179, 100),
                traffic_light_yellow=(240, 189, 39), traffic_light_red=(224, 53, 49), yield_sign=(210, 125, 45),
                stop_sign=(72, 60, 50), goal_waypoint=(139, 64, 0))
