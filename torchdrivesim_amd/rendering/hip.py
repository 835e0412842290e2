"""
The MI355X rendering backend: `HipRenderer(BirdviewRenderer)` with the CV2 backend's pixel semantics
(torchdrivesim/rendering/cv2.py:27-70) computed by the K3 rasteriser (torchdrivesim_amd/csrc/raster.hip).

Two entry points:
  * `render_rgb_mesh(mesh, res, cameras)` -- the reference's abstract method for an arbitrary per-camera RGB mesh;
  * `render_scene(...)` -- the fused path `Simulator.render` takes: static map from a device-resident handle, actor
    mesh generated inside the kernel from agent state; never materialises per-camera meshes (mesh.py:1147-1156 does).
"""
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor

from torchdrivesim_amd import _ops
from torchdrivesim_amd.mesh import RGBMesh
from torchdrivesim_amd.rendering.base import BirdviewRenderer, Cameras, RendererConfig
from torchdrivesim_amd.utils import Resolution


@dataclass
class HipRendererConfig(RendererConfig):
    backend: str = 'hip'
    out_dtype: str = 'float32'      #: 'float32' (reference-faithful values 0..255) or 'uint8' (same values, 4x fewer bytes)


@dataclass
class CV2RendererConfig(HipRendererConfig):
    """The reference's name for the configuration of its OpenCV backend (rendering/cv2.py:12-15).  A configuration written for that
    backend selects the MI355X rasteriser here: same pixel semantics, no OpenCV involved."""
    backend: str = 'cv2'
    trim_mesh_before_rendering: bool = True      #: cv2.py:15; False keeps the faces that have no vertex in view (cv2.py:32-41)


def level_table(*level_sources) -> list:
    """distinct rendering levels in descending order (painter order: first = drawn first)"""
    vals = set()
    for src in level_sources:
        vals.update(float(v) for v in src)
    return sorted(vals, reverse=True)


class HipRenderer(BirdviewRenderer):
    def __init__(self, cfg: HipRendererConfig, *args, **kwargs):
        super().__init__(cfg, *args, **kwargs)
        self.cfg: HipRendererConfig = cfg

    @property
    def trim(self) -> bool:
        """cv2.py:32-41: with trim_mesh_before_rendering (the default) a face is drawn only if one of its vertices lies in the 1.05 x view"""
        return bool(getattr(self.cfg, 'trim_mesh_before_rendering', True))

    @property
    def out_dtype(self) -> torch.dtype:
        return {'float32': torch.float32, 'uint8': torch.uint8}[getattr(self.cfg, 'out_dtype', 'float32')]

    def render_rgb_mesh(self, mesh: RGBMesh, res: Resolution, cameras: Cameras) -> Tensor:
        if res.width != res.height:
            raise RuntimeError('only square resolutions are supported (as in the reference, rendering/base.py:136)')
        n = cameras.xy.shape[0]
        verts, attrs, faces = mesh.verts, mesh.attrs, mesh.faces
        if verts.shape[0] != n:
            raise RuntimeError(f'mesh batch {verts.shape[0]} does not match the number of cameras {n}')
        if verts.shape[-1] == 2:
            verts = torch.cat([verts, torch.zeros_like(verts[..., :1])], dim=-1)
        levels = level_table(torch.unique(verts[..., 2]).tolist()) if verts.shape[1] > 0 else [0.0]
        img = _ops.raster_mesh(verts, attrs, faces, cameras.xy, cameras.sc, levels, cameras.scale, res.height, out_dtype=self.out_dtype, trim=self.trim)
        return img.permute(0, 2, 3, 1)        # (n,H,W,3) view; render_frame permutes it back to CHW without a copy

    def make_static_map(self, rgb_mesh: RGBMesh, extra_levels=(), device=None) -> _ops.StaticMap:
        """Upload a single (batch 1) static RGB mesh with z levels to the device: colour and level of every face are
        those of its first vertex (cv2.py:44-46,58)."""
        verts = rgb_mesh.verts[0].detach().cpu()
        faces = rgb_mesh.faces[0].detach().cpu().long()
        attrs = rgb_mesh.attrs[0].detach().cpu()
        first = faces[:, 0] if faces.shape[0] else faces.new_zeros((0,))
        face_z = verts[first, 2].to(torch.float32).numpy() if faces.shape[0] else np.zeros(0, np.float32)
        face_rgb = _ops.quantise_colors(attrs[first]).numpy().astype(np.uint32) if faces.shape[0] else np.zeros(0, np.uint32)
        levels = level_table(face_z.tolist(), extra_levels)
        # grid cells of about two thirds of the field of view: the scan walks one contiguous entry range per grid row under the view, and
        # fewer, longer ranges beat tighter culling (measured at fov 35 m: 8 m cells 7.59 ms, 24 m 7.40, 48 m 7.64; 128 x 128: 5.53 / 5.25)
        fov = 2.0 / float(self.scale)
        return _ops.StaticMap(verts[:, :2], faces, face_z, face_rgb, levels, device=device or rgb_mesh.device,
                              cell_size=min(max(0.65 * fov, 8.0), 32.0))

    def render_scene(self, static_map: _ops.StaticMap, state: Tensor, agent_sc: Tensor, tmpl: Tensor, actor_key: Tensor, mask: Tensor,
                     camera_xy: Tensor, camera_sc: Tensor, res: Optional[Resolution] = None, fov: Optional[float] = None,
                     key_table=None, differentiable: bool = False, extra_tri: Optional[Tensor] = None,
                     extra_key: Optional[Tensor] = None, key_colors: Optional[Tensor] = None, color_keys=None,
                     out: Optional[Tensor] = None) -> Tensor:
        """-> B x Nc x 3 x H x W.  `out`: a caller-owned contiguous B x Nc x 3 x H x W tensor of the renderer's output dtype to render into
        (the C ABI takes caller buffers, include/tdship.h; the reference allocates per call, rendering/cv2.py:52) -- not for differentiable calls.  `differentiable`: attach the K3 backward (gradients w.r.t. state[..., :2], agent_sc, camera_xy,
        camera_sc; float32 output only).  `extra_tri` (B,Nc,K,3,2) / `extra_key` (B,Nc,K): per-camera world-space triangles.
        `key_colors` (K,3) / `color_keys` (K packed keys): colour-gradient handle, see _ops.raster_scene_diff."""
        res = self.res if res is None else res
        if res.width != res.height:
            raise RuntimeError('only square resolutions are supported')
        fov = fov if fov is not None else 2.0 / self.scale
        if differentiable:
            if self.out_dtype != torch.float32:
                raise RuntimeError('the differentiable path renders float32 images')
            if out is not None:
                raise RuntimeError('`out=` cannot be combined with a differentiable render (autograd owns the image)')
            return _ops.raster_scene_diff(static_map, state, agent_sc, tmpl, actor_key, mask, camera_xy, camera_sc, fov, res.height, key_table=key_table,
                                          extra_tri=extra_tri, extra_key=extra_key, key_colors=key_colors, color_keys=color_keys, trim=self.trim)
        return _ops.raster_scene(static_map, state, agent_sc, tmpl, actor_key, mask, camera_xy, camera_sc, fov, res.height, out_dtype=self.out_dtype,
                                 key_table=key_table, extra_tri=extra_tri, extra_key=extra_key, trim=self.trim, out=out)


def allocate_image_ring(render, shape, dtype=torch.float32, device='cuda', count: int = 2, candidates: int = 5, reps: int = 2, spread: float = 1.03,
                        repeat_fast: bool = True):
    """`count` caller-owned output buffers for `render(out=buffer)` (e.g. `lambda out: sim.render_egocentric(res=res, out=out)`), chosen as
    the fastest of up to `candidates` allocations of `shape`.

    Why choose: the rasteriser is bound by the write stream, and on MI355X what the write stream of its launch reaches depends on the
    ALLOCATION it writes to -- a 51.5 GB tensor is either "fast" (7.1 ms per launch) or "slow" (8.3 ms: every XCD's stores are 13 - 16 %
    slower), for as long as it lives, whichever kernel touches it first and wherever it lies; about every second allocation of a fresh
    process is slow (tools/slow_buffer_probe.py, tools/xcd_finish_times.py).  A loop that owns its observation ring can
    pay for that once, at start-up: allocate candidates one after the other, time a launch into each, stop as soon as the `count` fastest
    lie within `spread` of each other (the first two, when both are fast), keep those and free the rest.  The candidates are all held until
    the choice is made -- a slow allocation that is freed early would be handed out again -- so `candidates` x the buffer must fit the
    device (it is cut to what does).
    `repeat_fast`: when fewer than `count` candidates are fast, the returned list repeats the fast ones (a shorter ring) instead of taking a slow one.
    Returns (buffers, report) with report = dict(first_touch_ms=[...], launch_ms=[...], kept=[indices]) over the candidates tried.
    The candidates that are not kept go back to the driver (torch.cuda.empty_cache) so that a later allocation does not get them again."""
    device = torch.device(device)
    nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    free, _ = torch.cuda.mem_get_info(device)
    n_max = max(count, min(candidates, int((free - (4 << 30)) // max(nbytes, 1))))
    cands, first, best = [], [], []
    while len(cands) < n_max:
        buf = torch.empty(shape, dtype=dtype, device=device)
        cands.append(buf)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 2)]
        ev[0].record()
        render(buf)                                    # first touch: maps the memory
        ev[1].record()
        for r in range(reps):
            render(buf)
            ev[2 + r].record()
        torch.cuda.synchronize(device)
        first.append(ev[0].elapsed_time(ev[1]))
        best.append(min(ev[1 + r].elapsed_time(ev[2 + r]) for r in range(reps)))
        if len(cands) >= max(count, 2):
            top = sorted(best)[:count]
            if top[-1] <= spread * min(best) and (count > 1 or len(cands) > 1):
                break
    order = sorted(range(len(cands)), key=lambda i: best[i])
    fast = [i for i in order if i == order[0] or best[i] <= spread * best[order[0]]]
    if len(fast) >= count or not repeat_fast:
        kept = sorted(order[:count])
    else:
        # fewer fast allocations than buffers asked for, and the device holds no more candidates: a shorter ring used in turn (entries
        # repeat) rather than a slow buffer in it -- the renders of one stream are ordered anyway
        kept = [fast[i % len(fast)] for i in range(count)]
    out = [cands[i] for i in kept]
    del cands, buf
    torch.cuda.empty_cache()
    return out, dict(first_touch_ms=first, launch_ms=best, kept=kept)
