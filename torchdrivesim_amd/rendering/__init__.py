"""
Renderers.  The reference ships OpenCV / pytorch3d / nvdiffrast backends behind `renderer_from_config`
(torchdrivesim/rendering/__init__.py:18-50); here the one real backend is the MI355X rasteriser (`hip`), which has
the OpenCV backend's pixel semantics, plus the dummy renderer.  'default' selects `hip`; a `CV2RendererConfig` (the reference's
OpenCV configuration) selects it too.
"""
from torchdrivesim_amd.rendering.base import (RendererConfig, DummyRendererConfig, BirdviewRenderer, DummyRenderer, Cameras,
                                              get_default_color_map, get_default_rendering_levels)
from torchdrivesim_amd.rendering.hip import HipRendererConfig, CV2RendererConfig, HipRenderer, allocate_image_ring


def renderer_from_config(cfg: RendererConfig, *args, **kwargs) -> BirdviewRenderer:
    assert isinstance(cfg, RendererConfig)
    if cfg.backend == 'default':
        cfg = HipRendererConfig(left_handed_coordinates=cfg.left_handed_coordinates, render_agent_direction=cfg.render_agent_direction,
                                highlight_ego_vehicle=cfg.highlight_ego_vehicle)
    if isinstance(cfg, DummyRendererConfig):
        return DummyRenderer(cfg, *args, **kwargs)
    if isinstance(cfg, HipRendererConfig):
        return HipRenderer(cfg, *args, **kwargs)
    raise ValueError(f'Unrecognized renderer type: {type(cfg)} (backends available here: hip, dummy)')
