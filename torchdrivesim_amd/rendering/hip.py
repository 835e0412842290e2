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

    def scene_maps(self, rgb_mesh: RGBMesh, actor_levels=(), device=None):
        """The device map(s) of a BATCH of static meshes: ONE per distinct batch element (`_ops.group_rows`: the scenes of a collated batch
        that share a map hold identical padded rows, mesh.py:172-200), each taken from the process-wide content cache (`_ops.map_cache`) or
        built once.  Returns a StaticMap when the whole batch shares one mesh, else a StaticMapSet whose `scene_map` says which map scene b
        uses; all maps of a set carry the same level table (the actors' levels + every z of the batch)."""
        dev = torch.device(device or rgb_mesh.device)
        tensors = [t if t.is_cuda else t.to(dev) for t in (rgb_mesh.verts, rgb_mesh.faces, rgb_mesh.attrs)]
        scene_map, reps, hashes = _ops.group_rows(tensors)
        # the level table of every map: the actors' levels + every z of the batch's meshes.  (For a single mesh its own levels are in its
        # table anyway; listing them keeps the cache key of a sub-batch on ONE of the towns equal to the key the mixed batch built it under.)
        extra = [float(z) for z in actor_levels] + [float(z) for z in torch.unique(torch.cat([tensors[0][r, :, 2] for r in reps])).tolist()]
        extra_key = tuple(sorted(set(extra)))
        maps = []
        for r, h in zip(reps, hashes):
            rows = [t[r] for t in tensors]
            key = ('render', str(dev), h, tuple(tuple(x.shape) for x in rows), extra_key, float(self.scale))
            maps.append(_ops.map_cache.get(key, rows, lambda r=r: self.make_static_map(rgb_mesh[r:r + 1], extra, device=dev)))
        if len(maps) == 1:
            return maps[0]
        return _ops.StaticMapSet(maps, torch.from_numpy(scene_map))

    def render_scene(self, static_map: _ops.StaticMap, state: Tensor, agent_sc: Tensor, tmpl: Tensor, actor_key: Tensor, mask: Tensor,
                     camera_xy: Tensor, camera_sc: Tensor, res: Optional[Resolution] = None, fov: Optional[float] = None,
                     key_table=None, differentiable: bool = False, extra_tri: Optional[Tensor] = None,
                     extra_key: Optional[Tensor] = None, key_colors: Optional[Tensor] = None, color_keys=None,
                     out: Optional[Tensor] = None, ego_cameras: int = 0) -> Tensor:
        """-> B x Nc x 3 x H x W.  `out`: a caller-owned contiguous B x Nc x 3 x H x W tensor of the renderer's output dtype to render into
        (the C ABI takes caller buffers, include/tdship.h; the reference allocates per call, rendering/cv2.py:52) -- not for differentiable calls.  `differentiable`: attach the K3 backward (gradients w.r.t. state[..., :2], agent_sc, camera_xy,
        camera_sc; float32 output only).  `extra_tri` (B,Nc,K,3,2) / `extra_key` (B,Nc,K): per-camera world-space triangles.
        `key_colors` (K,3) / `color_keys` (K packed keys): colour-gradient handle, see _ops.raster_scene_diff.
        `ego_cameras` = Nc (differentiable calls): the cameras are the first Nc agents of `state` / `agent_sc` -- their gradient is folded into the agents'."""
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
                                          extra_tri=extra_tri, extra_key=extra_key, key_colors=key_colors, color_keys=color_keys, trim=self.trim,
                                          ego_cameras=ego_cameras)
        return _ops.raster_scene(static_map, state, agent_sc, tmpl, actor_key, mask, camera_xy, camera_sc, fov, res.height, out_dtype=self.out_dtype,
                                 key_table=key_table, extra_tri=extra_tri, extra_key=extra_key, trim=self.trim, out=out)


class _DeviceTimer:
    """what allocate_image_ring measures, on the device (tests inject their own timer with the same three methods)"""

    def __init__(self, render, device, reps):
        self.render, self.device, self.reps = render, device, reps

    def _ms(self, fn, reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r in range(reps):
            fn()
            ev[r + 1].record()
        torch.cuda.synchronize(self.device)
        return [ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]

    def first_touch(self, buf) -> float:
        return self._ms(lambda: self.render(buf), 1)[0]

    def launch(self, buf) -> float:
        return min(self._ms(lambda: self.render(buf), self.reps))

    def fill(self, buf) -> float:
        return min(self._ms(lambda: buf.fill_(0), 2))


def allocate_image_ring(render, shape, dtype=torch.float32, device='cuda', count: int = 2, candidates: int = 4, reps: int = 2, fast: float = 0.98,
                        spread: float = 1.015, allow_aliasing: bool = False, timer=None, alloc=None, short_ms: float = 3.0, agree: float = 1.04):
    """`count` DISTINCT caller-owned output buffers for `render(out=buffer)` (e.g. `lambda out: sim.render_egocentric(res=res, out=out)`), each
    checked to be one the write stream of the launch is served at full rate into.

    Why check: the rasteriser is bound by the write stream, and on MI355X what a write stream reaches depends on the PHYSICAL pages under the
    buffer -- a 51.5 GB hipMalloc is served at 1, 15/16 or 7/8 of the rate (7.03 / 7.45 / 8.05 ms per launch) for as long as it lives,
    about one in three at 7/8, while torch's fill_ takes 7.4 - 7.5 ms on all of them (DESIGN_HISTORY.md section 4, tools/alloc_probe.hip).  The
    buffers here come from `_ops.owned_image` -- the library's allocator that spreads the physical pages out (csrc/alloc.hip), which rarely
    produces a slower buffer (one of 100 in round 4, at 15/16 of the rate) -- and every candidate is still MEASURED, against an absolute yardstick of the same run:
        a candidate is fast  iff  its launch takes at most `fast` x the fill_ time (the fastest fill_ seen over the candidates)
                             and  at most `spread` x the fastest launch seen over the candidates
    (the second clause catches the in-between class -- 7.45 ms where 7.03 is possible -- that passes the first when fill_ itself is slow into
    spread-out pages: 7.6 - 7.9 ms instead of 7.45 -- and, at 1.5 %, the buffers that are merely a little slower: spread-out buffers differ by 1 - 2 % among
    themselves, for as long as they live; it cannot reject a candidate before a faster one has been seen).
    Candidates are allocated one after the other and all held until the choice is made (a rejected allocation that is freed would be handed
    out again); the search ends as soon as `count` fast ones exist, and never before unless `candidates` (cut to what the device holds)
    are exhausted.  When fewer than `count` are fast, the ring is filled up with the best of the others -- still distinct buffers: step i
    renders while the consumer of step i - 1 holds the other buffer, so an aliased ring would overwrite a live observation.
    `allow_aliasing=True` (a loop that consumes every image before the next render) repeats the fast buffer(s) instead; the report says so.
    A launch that takes more than 1.5 x the fill_ is not bound by the write stream (uint8 output, low resolutions): placement does not
    matter there and the first `count` candidates are taken.
    `timer`: an object with first_touch(buf) / launch(buf) / fill(buf) -> ms (tests); `alloc(shape, dtype, device)`: the allocator.
    Returns (buffers, report); report = dict(first_touch_ms, launch_ms, fill_ms (the yardstick), fast=[bool per candidate], kept=[indices],
    aliased=bool, write_bound=bool, yardstick='fill' | 'not applicable' (short launches: none of count + 1 agreeing candidates beats its
    fill_; the first `count` are kept) | 'not write-bound')."""
    device = torch.device(device)
    alloc = alloc if alloc is not None else _ops.owned_image
    timer = timer if timer is not None else _DeviceTimer(render, device, reps)
    nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
    n_max = max(count, candidates)
    if device.type == 'cuda':
        free, _ = torch.cuda.mem_get_info(device)
        # every candidate is held until the choice is made; building one needs twice its size for a moment (the spacers of csrc/alloc.hip)
        n_max = max(count, min(n_max, int((free - (4 << 30)) // max(nbytes, 1)) - 1))
    cands, first, best, fills = [], [], [], []
    write_bound, yardstick = True, 'fill'

    def is_fast(i):
        return best[i] <= fast * min(fills) and best[i] <= spread * min(best)

    while len(cands) < n_max:
        buf = alloc(shape, dtype, device)
        cands.append(buf)
        first.append(timer.first_touch(buf))            # maps the memory
        fills.append(timer.fill(buf))
        best.append(timer.launch(buf))                  # last: the buffer is handed over holding a rendered image
        if len(cands) == 1 and best[0] > 1.5 * fills[0]:
            write_bound = False
        if not write_bound and len(cands) >= count:
            break
        if write_bound and sum(is_fast(i) for i in range(len(cands))) >= count:
            break
        # The dead band between `fast` x fill_ and 1.5 x fill_: a SHORT write-bound launch (12.9 GB at B = 256: 1.85 ms, its ramp and tail weigh
        # 5 %) never beats its fill_, whatever the buffer -- but neither does a launch into slow pages at the headline size.  So only for
        # launches whose fill_ takes less than `short_ms` (3 ms: ramp and tail above the 2 % the yardstick resolves): once count + 1 candidates
        # have been seen, none passes and all lie within `agree` (4 %: the placement classes are 16 : 15 : 14, short launches scatter by 2 %) of
        # the fastest, the yardstick says nothing at this
        # size and probing on would only build more buffers to keep the first ones.
        if write_bound and min(fills) < short_ms and len(cands) >= count + 1 and not any(b <= fast * min(fills) for b in best) and \
                max(best) <= agree * min(best):
            yardstick = 'not applicable'
            break
    order = sorted(range(len(cands)), key=lambda i: best[i])
    fast_ones = [i for i in order if is_fast(i)] if write_bound else list(range(len(cands)))
    if yardstick == 'not applicable':
        fast_ones = []                                      # nothing to choose by: the first `count`, as for a launch that is not write-bound
        order = list(range(len(cands)))
    aliased = False
    if len(fast_ones) >= count:
        kept = sorted(fast_ones[:count])
    elif allow_aliasing and fast_ones:
        kept, aliased = [fast_ones[i % len(fast_ones)] for i in range(count)], True
    else:
        kept = sorted(order[:count])                       # the fast ones and the best of the rest: distinct buffers
    out = [cands[i] for i in kept]
    report = dict(first_touch_ms=first, launch_ms=best, fill_ms=min(fills), fast=[bool(is_fast(i)) for i in range(len(cands))] if write_bound else None,
                  kept=kept, aliased=aliased, write_bound=write_bound, yardstick=yardstick if write_bound else 'not write-bound')
    del cands, buf
    if device.type == 'cuda':
        torch.cuda.empty_cache()
    return out, report
