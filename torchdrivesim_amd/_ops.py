"""
Tensor-level entry points of the HIP kernels (device tensors in, device tensors out, launched on torch's current
stream through the C ABI of include/tdship.h).  The reference-shaped classes in kinematic.py, infractions.py,
rendering/ and simulator.py are thin layers over these.
"""
import atexit
import collections
import ctypes
import sys
import threading

import numpy as np
import torch

from torchdrivesim_amd import _native as nat

f32, u8, i32 = torch.float32, torch.uint8, torch.int32


def _c(t, dtype=f32):
    """dense tensor of the dtype the kernels read (no copy when already so)"""
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def heading_sc(psi):
    """[sin, cos] of headings with torch on the tensor's device -- exactly where the reference calls torch.sin/cos
    (simulator.py:940, utils.py:40-53, _iou_utils.py:290-291).  psi: (...,) -> (..., 2); differentiable."""
    if psi.requires_grad and torch.is_grad_enabled():
        return torch.stack([torch.sin(psi), torch.cos(psi)], dim=-1)
    sc = torch.empty(psi.shape + (2,), dtype=psi.dtype, device=psi.device)          # two launches instead of three: the same torch.sin / torch.cos, written in place
    torch.sin(psi, out=sc[..., 0])
    torch.cos(psi, out=sc[..., 1])
    return sc


_zero_consts = {}


def _zeros_const(shape, device):
    """a zero tensor that is never written (gradient columns that are identically zero are concatenated from it: one launch, no fill).
    Cached per (shape, device) once it exists on the device for every stream: created outside stream capture only (a captured
    torch.zeros is a graph node in the graph's private pool, not memory an eager backward may read later) and followed by ONE
    synchronisation of the creating stream, so that a first use from another stream is ordered behind the fill."""
    dev = torch.device(device)
    if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
        return torch.zeros(tuple(int(d) for d in shape), dtype=f32, device=dev)
    key = (tuple(int(d) for d in shape), str(dev))
    z = _zero_consts.get(key)
    if z is None:
        if len(_zero_consts) > 64:
            _zero_consts.clear()
        z = _zero_consts[key] = torch.zeros(key[0], dtype=f32, device=dev)
        if dev.type == 'cuda':
            torch.cuda.current_stream(dev).synchronize()
    return z


class _StateHeadingSC(torch.autograd.Function):
    """[sin psi, cos psi] of a (..., 4) state tensor as ONE autograd node.  The values are torch.sin / torch.cos of the psi column (what the
    reference computes, simulator.py:940); the backward is the chain rule in closed form, d/dpsi = g_sin cos - g_cos sin, written into the psi
    column of a zero state gradient -- three small launches instead of the eight of autograd's select / sin / cos / stack chain, and the
    consumers of one state (render, collision, off-road) can share the node (Simulator._heading_sc).  The node is shared, so it must be walkable
    more than once without retain_graph (autograd.grad of the collisions, then of the off-road loss: save_for_backward would free the values after
    the first walk) -- the values are kept on ctx, as a DETACHED ALIAS of the output: the output itself on ctx made the cycle
    output -> grad_fn -> ctx -> output, which only the cycle collector frees (GPU memory pressure does not wake it); the alias shares the
    storage and references nothing.  (An in-place edit of the returned tensor is therefore not caught by autograd's version check.)"""

    @staticmethod
    def forward(ctx, state):
        psi = state[..., 2]
        sc = torch.empty(state.shape[:-1] + (2,), dtype=state.dtype, device=state.device)
        torch.sin(psi, out=sc[..., 0])
        torch.cos(psi, out=sc[..., 1])
        ctx.sc = sc.detach()
        return sc

    @staticmethod
    def backward(ctx, g):
        sc = ctx.sc
        gpsi = torch.mul(g[..., 0], sc[..., 1]).addcmul_(g[..., 1], sc[..., 0], value=-1.0)
        z = _zeros_const(gpsi.shape + (2,), gpsi.device)
        return torch.cat([z, gpsi.unsqueeze(-1), z[..., :1]], dim=-1)          # (..., 4): zeros, zeros, d/dpsi, zeros


def state_heading_sc(state):
    """differentiable [sin psi, cos psi] of a (..., 4) state, one autograd node (see _StateHeadingSC)"""
    return _StateHeadingSC.apply(state)


class _Boxes(torch.autograd.Function):
    """[x, y, length, width, psi] boxes (simulator.py:1093) from a (..., 4) state and (..., 2) sizes as one autograd node: a cat forward, and a
    backward that builds both gradients with two launches instead of the zeros + copies + adds of three slice nodes."""

    @staticmethod
    def forward(ctx, state, size):
        ctx.need = (state.requires_grad, size.requires_grad)
        return torch.cat([state[..., :2], size, state[..., 2:3]], dim=-1)

    @staticmethod
    def backward(ctx, g):
        gs = gz = None
        if ctx.need[0]:
            gs = torch.cat([g[..., :2], g[..., 4:5], _zeros_const(g.shape[:-1] + (1,), g.device)], dim=-1)      # d/d(x, y, psi), 0 for v
        if ctx.need[1]:
            gz = g[..., 2:4].contiguous()
        return gs, gz


def state_boxes(state, size):
    return _Boxes.apply(state, size)


# ---------------------------------------------------------------------------------------------------------------
# K1 kinematics
# ---------------------------------------------------------------------------------------------------------------
class _BicycleStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, action, lr, dt, max_acc, max_steer, left_handed, no_reversing):
        state, action, lr = _c(state), _c(action), _c(lr)
        out = torch.empty_like(state)
        n = lr.numel()
        nat.call('tds_bicycle_step_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(action, f32, 'action'),
                 nat.dev_ptr(lr, f32, 'lr'), nat.dev_ptr(out, f32, 'out'), n, dt, max_acc, max_steer, int(left_handed),
                 int(no_reversing), nat.stream_ptr(state.device))
        ctx.save_for_backward(state, action, lr)
        ctx.cfg = (dt, max_acc, max_steer, int(left_handed), int(no_reversing))
        return out

    @staticmethod
    def backward(ctx, gout):
        state, action, lr = ctx.saved_tensors
        dt, max_acc, max_steer, lh, norev = ctx.cfg
        gout = _c(gout)
        gs, ga, gl = torch.empty_like(state), torch.empty_like(action), torch.empty_like(lr)
        nat.call('tds_bicycle_step_bwd_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(action, f32, 'action'),
                 nat.dev_ptr(lr, f32, 'lr'), nat.dev_ptr(gout, f32, 'grad_out'), nat.dev_ptr(gs, f32, 'gs'), nat.dev_ptr(ga, f32, 'ga'),
                 nat.dev_ptr(gl, f32, 'gl'), lr.numel(), dt, max_acc, max_steer, lh, norev, nat.stream_ptr(state.device))
        return gs, ga, gl, None, None, None, None, None


def bicycle_step(state, action, lr, dt=0.1, max_acc=5.0, max_steer=float(np.float32(np.pi / 2)), left_handed=False,
                 no_reversing=False):
    """state (...,4), action (...,2), lr (...) -> new state tensor (...,4) (kinematic.py:462-477 / :513-523)"""
    assert state.shape[-1] == 4 and action.shape[-1] == 2 and state.shape[:-1] == action.shape[:-1] == lr.shape
    return _BicycleStep.apply(state, action, lr, float(dt), float(max_acc), float(max_steer), bool(left_handed), bool(no_reversing))


class _SimpleStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, action, dt, norm, oriented):
        state, action = _c(state), _c(action)
        out = torch.empty_like(state)
        cn = (ctypes.c_float * 4)(*norm)
        nat.call('tds_simple_step_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(action, f32, 'action'),
                 nat.dev_ptr(out, f32, 'out'), state.numel() // 4, dt, cn, int(oriented), nat.stream_ptr(state.device))
        ctx.save_for_backward(state, action)
        ctx.cfg = (dt, tuple(norm), int(oriented))
        return out

    @staticmethod
    def backward(ctx, gout):
        state, action = ctx.saved_tensors
        dt, norm, oriented = ctx.cfg
        gout = _c(gout)
        gs, ga = torch.empty_like(state), torch.empty_like(action)
        cn = (ctypes.c_float * 4)(*norm)
        nat.call('tds_simple_step_bwd_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(action, f32, 'action'),
                 nat.dev_ptr(gout, f32, 'grad_out'), nat.dev_ptr(gs, f32, 'gs'), nat.dev_ptr(ga, f32, 'ga'), state.numel() // 4, dt, cn,
                 oriented, nat.stream_ptr(state.device))
        return gs, ga, None, None, None


def simple_step(state, action, dt=0.1, norm=(20.0, 20.0, 10 * np.pi, 5.0), oriented=False):
    assert state.shape[-1] == 4 and action.shape == state.shape
    return _SimpleStep.apply(state, action, float(dt), tuple(float(x) for x in norm), bool(oriented))


class _UnicycleStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, action, dt, max_acc, max_w):
        state, action = _c(state), _c(action)
        out = torch.empty_like(state)
        nat.call('tds_unicycle_step_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(action, f32, 'action'),
                 nat.dev_ptr(out, f32, 'out'), state.numel() // 4, dt, max_acc, max_w, nat.stream_ptr(state.device))
        ctx.save_for_backward(state, action)
        ctx.cfg = (dt, max_acc, max_w)
        return out

    @staticmethod
    def backward(ctx, gout):
        state, action = ctx.saved_tensors
        dt, max_acc, max_w = ctx.cfg
        gout = _c(gout)
        gs, ga = torch.empty_like(state), torch.empty_like(action)
        nat.call('tds_unicycle_step_bwd_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(action, f32, 'action'),
                 nat.dev_ptr(gout, f32, 'grad_out'), nat.dev_ptr(gs, f32, 'gs'), nat.dev_ptr(ga, f32, 'ga'), state.numel() // 4, dt,
                 max_acc, max_w, nat.stream_ptr(state.device))
        return gs, ga, None, None, None


def unicycle_step(state, action, dt=0.1, max_acc=5.0, max_yaw_rate=1.0):
    assert state.shape[-1] == 4 and action.shape[-1] == 2
    return _UnicycleStep.apply(state, action, float(dt), float(max_acc), float(max_yaw_rate))


# ---------------------------------------------------------------------------------------------------------------
# K2a collisions
# ---------------------------------------------------------------------------------------------------------------
_METRICS = {'iou': nat.METRIC_IOU, 'discs': nat.METRIC_DISCS}


def metric_sc(boxes, metric):
    """[sin, cos] of the heading the metric uses: psi (iou, _iou_utils.py:290-291) or
    psi + pi/2 * (width > length) (discs, infractions.py:404).  boxes (...,5)."""
    psi = boxes[..., 4]
    if metric == 'discs':
        psi = psi + (np.pi / 2) * (boxes[..., 3] > boxes[..., 2])
    return heading_sc(psi)


def collision_forward(boxes, sc, present, n_exposed, metric, want_overlap=False, want_partner=False):
    boxes, sc = _c(boxes), _c(sc)
    present = _c(present, u8) if present.dtype != torch.bool else present.contiguous().view(u8)
    B, N = boxes.shape[:2]
    A = N if n_exposed is None else int(n_exposed)
    out = torch.empty((B, A), dtype=f32, device=boxes.device)
    overlap = torch.empty((B, A), dtype=torch.int64, device=boxes.device) if want_overlap else None
    partner = torch.empty((B, A), dtype=i32, device=boxes.device) if want_partner else None
    nat.call('tds_collision_f32', boxes.device, nat.dev_ptr(boxes, f32, 'boxes'), nat.dev_ptr(sc, f32, 'sc'), nat.dev_ptr(present, u8, 'present'),
             nat.dev_ptr(out, f32, 'out'), nat.dev_ptr(overlap, torch.int64, 'overlap'), nat.dev_ptr(partner, i32, 'partner'), B, A, N,
             _METRICS[metric], nat.stream_ptr(boxes.device))
    return out, overlap, partner


def overlap_count(boxes, present, sc=None):
    """The `nograd` collision metric (simulator.py:1111-1149, infractions.py:352-375): boxes (B,A,5), present (B,A) -> (B,A) float64,
    the number of other present agents whose rectangle shares area with the agent's.  NaNs are scrubbed to 0 first."""
    boxes = torch.nan_to_num(_c(boxes.detach()), nan=0.0)
    sc = heading_sc(boxes[..., 4]) if sc is None else _c(sc)
    present = _c(present, u8) if present.dtype != torch.bool else present.contiguous().view(u8)
    B, A = boxes.shape[:2]
    out = torch.empty((B, A), dtype=torch.float64, device=boxes.device)
    nat.call('tds_overlap_count_f32', boxes.device, nat.dev_ptr(boxes, f32, 'boxes'), nat.dev_ptr(sc, f32, 'sc'), nat.dev_ptr(present, u8, 'present'),
             nat.dev_ptr(out, torch.float64, 'out'), B, A, nat.stream_ptr(boxes.device))
    return out


class _Collision(torch.autograd.Function):
    @staticmethod
    def forward(ctx, boxes, sc, present, n_exposed, metric):
        boxes, sc = _c(boxes), _c(sc)
        out, _, _ = collision_forward(boxes, sc, present, n_exposed, metric)
        ctx.save_for_backward(boxes, sc, present)
        ctx.cfg = (n_exposed, metric)
        return out

    @staticmethod
    def backward(ctx, gout):
        boxes, sc, present = ctx.saved_tensors
        n_exposed, metric = ctx.cfg
        B, N = boxes.shape[:2]
        A = N if n_exposed is None else int(n_exposed)
        gout = _c(gout)
        pres = _c(present, u8) if present.dtype != torch.bool else present.contiguous().view(u8)
        gb, gsc = torch.empty_like(boxes), torch.empty_like(sc)
        nat.call('tds_collision_bwd_f32', boxes.device, nat.dev_ptr(boxes, f32, 'boxes'), nat.dev_ptr(sc, f32, 'sc'),
                 nat.dev_ptr(pres, u8, 'present'), nat.dev_ptr(gout, f32, 'grad_out'), nat.dev_ptr(gb, f32, 'gb'), nat.dev_ptr(gsc, f32, 'gsc'),
                 B, A, N, _METRICS[metric], nat.stream_ptr(boxes.device))
        return gb, gsc, None, None, None


def collision(boxes, present, n_exposed=None, metric='iou', sc=None):
    """Simulator.compute_collision fused over a scene: boxes (B,N,5) [x,y,len,wid,psi], present (B,N) -> (B,A)."""
    if sc is None:
        sc = metric_sc(torch.nan_to_num(boxes, nan=0.0), metric)
    return _Collision.apply(boxes, sc, present, n_exposed, metric)


def pairwise_overlap(box1, box2, metric='iou', sc1=None, sc2=None, num_discs=5):
    """iou_differentiable / collision_detection_with_discs (infractions.py:307,503), element-wise; boxes (...,5).
    Forward only (the fused scene entry point carries the gradient)."""
    shape = box1.shape[:-1]
    if sc1 is None:
        sc1 = metric_sc(box1, metric)
    if sc2 is None:
        sc2 = metric_sc(box2, metric)
    b1, b2, s1, s2 = _c(box1).reshape(-1, 5), _c(box2).reshape(-1, 5), _c(sc1).reshape(-1, 2), _c(sc2).reshape(-1, 2)
    out = torch.empty(b1.shape[0], dtype=f32, device=b1.device)
    if metric == 'discs' and num_discs != 5:
        nat.call('tds_pairwise_discs_f32', b1.device, nat.dev_ptr(b1, f32, 'box1'), nat.dev_ptr(s1, f32, 'sc1'), nat.dev_ptr(b2, f32, 'box2'),
                 nat.dev_ptr(s2, f32, 'sc2'), nat.dev_ptr(out, f32, 'out'), b1.shape[0], int(num_discs), nat.stream_ptr(b1.device))
        return out.reshape(shape)
    nat.call('tds_pairwise_overlap_f32', b1.device, nat.dev_ptr(b1, f32, 'box1'), nat.dev_ptr(s1, f32, 'sc1'), nat.dev_ptr(b2, f32, 'box2'),
             nat.dev_ptr(s2, f32, 'sc2'), nat.dev_ptr(out, f32, 'out'), b1.shape[0], _METRICS[metric], nat.stream_ptr(b1.device))
    return out.reshape(shape)


def box2corners(box, sc=None):
    """_iou_utils.box2corners_th: (...,5) -> (...,4,2)"""
    if sc is None:
        sc = heading_sc(box[..., 4])
    b, s = _c(box).reshape(-1, 5), _c(sc).reshape(-1, 2)
    out = torch.empty((b.shape[0], 4, 2), dtype=f32, device=b.device)
    nat.call('tds_box2corners_f32', b.device, nat.dev_ptr(b, f32, 'box'), nat.dev_ptr(s, f32, 'sc'), nat.dev_ptr(out, f32, 'out'), b.shape[0],
             nat.stream_ptr(b.device))
    return out.reshape(box.shape[:-1] + (4, 2))


# ---------------------------------------------------------------------------------------------------------------
# static map handle, K2b offroad
# ---------------------------------------------------------------------------------------------------------------
def occlusion_mask(state, size, present, n_exposed):
    """StandardSensingObservationNoise.get_noisy_present_mask (observation_noise.py:89-132): state (B,E,4), size (B,E,2),
    present (B,E) -> (B,A,E) bool, True = present and in line of sight of ego a"""
    B, E = present.shape
    state, size = _c(state.detach()), _c(size.detach())
    p8 = present.contiguous().view(u8) if present.dtype == torch.bool else _c(present, u8)
    out = torch.empty((B, int(n_exposed), E), dtype=u8, device=state.device)
    nat.call('tds_occlusion_mask_f32', state.device, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(size, f32, 'size'), nat.dev_ptr(p8, u8, 'present'),
             nat.dev_ptr(out, u8, 'out'), B, int(n_exposed), E, nat.stream_ptr(state.device))
    return out.view(torch.bool)


def quantise_colors(attrs):
    """cv2.py:50: floor(attr * (1 - 1e-3) * 256) as uint8, packed 0x00RRGGBB.  attrs (...,3) float32 cpu tensor -> int64"""
    q = (attrs.to(f32) * (1.0 - 1e-3) * 256).floor().to(torch.uint8).to(torch.int64)
    return (q[..., 0] << 16) | (q[..., 1] << 8) | q[..., 2]


class StaticMap:
    """Device-resident static mesh + uniform grid (tds_map_t).  Built once per map and per device from HOST arrays."""

    def __init__(self, verts, faces, face_z=None, face_rgb=None, levels=None, device='cuda', cell_size=0.0):
        verts = np.ascontiguousarray(torch.as_tensor(verts).detach().cpu().numpy(), dtype=np.float32).reshape(-1, 2)
        faces = np.ascontiguousarray(torch.as_tensor(faces).detach().cpu().numpy(), dtype=np.int32).reshape(-1, 3)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('StaticMap lives on an MI355X; there is no CPU implementation')
        self.n_verts, self.n_faces = verts.shape[0], faces.shape[0]
        self.levels = None if levels is None else [float(x) for x in levels]
        vp = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        fz = None if face_z is None else np.ascontiguousarray(np.asarray(face_z), dtype=np.float32)
        fc = None if face_rgb is None else np.ascontiguousarray(np.asarray(face_rgb), dtype=np.uint32)
        lv = None if levels is None else np.ascontiguousarray(np.asarray(levels), dtype=np.float32)
        handle = ctypes.c_void_p()
        global map_creations
        map_creations += 1
        nat.call('tds_map_create', self.device, vp(verts), vp(faces), vp(fz), vp(fc), verts.shape[0], faces.shape[0], vp(lv),
                 0 if lv is None else len(lv), float(cell_size), ctypes.byref(handle))
        self._h = handle

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError('StaticMap was destroyed')
        return self._h

    def face_keys(self):
        """distinct face keys of the map (None: more than 64) -- with the actors' keys they decide which rasteriser serves a launch"""
        if not hasattr(self, '_face_keys'):
            buf, n = (ctypes.c_uint32 * 64)(), ctypes.c_int(0)
            nat.call('tds_map_keys', self.device, self.handle, ctypes.cast(buf, ctypes.c_void_p), 64, ctypes.byref(n))
            self._face_keys = None if n.value < 0 else [int(buf[i]) for i in range(n.value)]
        return self._face_keys

    def info(self):
        buf = (ctypes.c_int64 * 10)()
        nat.call('tds_map_info_ex', self.device, self.handle, buf, 10)
        return dict(V=buf[0], F=buf[1], nx=buf[2], ny=buf[3], entries=buf[4], bytes=buf[5], n_levels=buf[6], near_candidates=buf[7],
                    render_entries=buf[8], pairs=buf[9])

    def rank_of(self, level):
        """1-based painter rank of a rendering level (larger = drawn later = on top)"""
        return self.levels.index(float(level)) + 1

    def _destroy(self):
        if getattr(self, '_h', None) is not None:
            h, self._h = self._h, None
            nat.call('tds_map_destroy', self.device, h)

    def close(self):
        """destroy the device map now -- unless the process-wide cache handed it out (`shared`): other simulators may hold it, and it is
        destroyed when the last holder drops it"""
        if not getattr(self, 'shared', False):
            self._destroy()

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass


#: how many tds_map_create calls this process has made (tests count them: one per DISTINCT mesh, none for a batch operation)
map_creations = 0

_ROW_SEEDS = (0x9e3779b97f4a7c15, 0xc2b2ae3d27d4eb4f)


def _rows_view(t):
    """(tensor to read, number of rows, bytes per row, bytes between rows) for the row kernels: whole 4-byte words, dense inside a row;
    a batch that is ONE row expanded (stride 0) is one row."""
    if t.dim() == 0:
        raise RuntimeError('group_rows needs tensors with a leading batch axis')
    if t.dtype in (torch.bool, torch.uint8, torch.int8, torch.int16, torch.float16, torch.bfloat16):
        t = t.to(torch.int32)
    n = t.shape[0]
    if n > 1 and t.stride(0) == 0:
        t, n = t[:1], 1
    if t.numel() and not t[0].is_contiguous():
        t = t.contiguous()
    row_bytes = (t[0].numel() if n else 0) * t.element_size()
    stride = t.stride(0) * t.element_size() if n > 1 else row_bytes
    return t, n, row_bytes, stride


def row_hashes(tensors):
    """(B, 2 * len(tensors)) int64 device tensor: two 64-bit content hashes of every batch row of every tensor (tds_rows_hash_u64).  A batch
    that is one row expanded is hashed once and the hash repeated."""
    B = tensors[0].shape[0]
    dev = tensors[0].device
    cols = []
    for t in tensors:
        if t.shape[0] != B:
            raise RuntimeError('row_hashes: the tensors do not share the batch axis')
        v, n, row_bytes, stride = _rows_view(t.detach())
        for seed in _ROW_SEEDS:
            h = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
            if n:
                nat.call('tds_rows_hash_u64', dev, ctypes.c_void_p(v.data_ptr()), n, row_bytes, stride, ctypes.c_uint64(seed), ctypes.c_void_p(h.data_ptr()),
                         nat.stream_ptr(dev))
            cols.append(h[:n].expand(B) if n != B else h)
    return torch.stack(cols, dim=1) if cols else torch.zeros(B, 0, dtype=torch.int64, device=dev)


def group_rows(tensors):
    """Which batch elements of the (device) tensors are identical in ALL of them -- the scenes of a collated batch that share a mesh
    (mesh.py:172-200: padded rows of the same map are identical byte for byte).  Returns (scene_map, reps, hashes):
        scene_map  (B,) numpy int32, group of every batch element, groups numbered by first occurrence;
        reps       list, the first batch element of every group;
        hashes     list of tuples, the content hashes of every group's rows (keys of the process-wide map cache).
    Grouping is by hash on the device, then CONFIRMED exactly (tds_rows_equal_u8 against the group's representative); a row that shares a
    hash but not the bytes becomes a group of its own.  One small device -> host copy (B hashes); the meshes stay on the device."""
    B = tensors[0].shape[0]
    if B == 0:
        return np.zeros(0, np.int32), [], []
    dev = tensors[0].device
    h = row_hashes(tensors).cpu().numpy()                                         # B x 2T
    index, reps, scene_map = {}, [], np.empty(B, np.int32)
    for b in range(B):
        key = h[b].tobytes()
        g = index.get(key)
        if g is None:
            g = index[key] = len(reps)
            reps.append(b)
        scene_map[b] = g
    if len(reps) < B:
        rep = torch.from_numpy(np.asarray(reps, np.int32)[scene_map]).to(dev)
        equal = torch.ones(B, dtype=u8, device=dev)
        for t in tensors:
            v, n, row_bytes, stride = _rows_view(t.detach())
            if n == B:
                nat.call('tds_rows_equal_u8', dev, ctypes.c_void_p(v.data_ptr()), n, row_bytes, stride, nat.dev_ptr(rep, i32, 'rep'), nat.dev_ptr(equal, u8, 'equal'),
                         nat.stream_ptr(dev))
        bad = np.nonzero(equal.cpu().numpy() == 0)[0]
        for b in bad:                                                            # a hash collision: the row gets a map of its own
            scene_map[b] = len(reps)
            reps.append(int(b))
    hashes = [tuple(int(x) for x in h[r]) for r in reps]
    return scene_map, reps, hashes


class MapCache:
    """Process-wide cache of device maps by CONTENT: (device, what the map was built for, content hashes of the mesh rows, parameters) ->
    StaticMap, with the rows it was built from kept beside it (device clones, about a megabyte per map) so that a hit is confirmed byte for
    byte.  Maps are immutable, so simulators, their copies, sub-batches (`select_batch_elements`), extensions and shards share handles
    instead of running tds_map_create again (the reference carries B private mesh copies through these operations, simulator.py:444-511).
    Least recently used entries are dropped beyond `capacity` maps (TDS_MAP_CACHE, default 64); a dropped map lives on while a simulator
    holds it."""

    def __init__(self, capacity=None):
        import os
        self.capacity = int(os.environ.get('TDS_MAP_CACHE', 64)) if capacity is None else capacity
        self._d = collections.OrderedDict()
        self._lock = threading.Lock()
        self.hits = self.misses = 0

    def get(self, key, rows, build):
        """the cached map of `key` if its stored rows equal `rows` (list of device tensors), else build() -> StaticMap, stored"""
        key = (key, id(nat.lib()))             # (the testing build is a separate image of the library: its handles stay with it)
        with self._lock:
            ent = self._d.get(key)
            if ent is not None:
                smap, kept = ent
                if smap._h is not None and len(kept) == len(rows) and all(a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a, b)) for a, b in zip(kept, rows)):
                    self._d.move_to_end(key)
                    self.hits += 1
                    return smap
            self.misses += 1
            smap = build()
            smap.shared = True
            self._d[key] = (smap, [r.detach().clone() for r in rows])
            self._d.move_to_end(key)
            while len(self._d) > max(self.capacity, 0):
                self._d.popitem(last=False)
            return smap

    def clear(self):
        with self._lock:
            self._d.clear()


map_cache = MapCache()
atexit.register(map_cache.clear)


class StaticMapSet:
    """Several StaticMaps of one device (created with the same level table) + which of them every scene of a batch uses: lets one
    launch serve a batch whose scenes have different meshes (tds_mapset_t).  `scene_map`: (B,) int32 device tensor of indices."""

    def __init__(self, maps, scene_map):
        assert len(maps) > 0
        self.maps = list(maps)                       # keeps the maps alive
        self.device = self.maps[0].device
        self.levels = self.maps[0].levels
        self.n_faces = sum(m.n_faces for m in self.maps)
        self.scene_map = _c(torch.as_tensor(scene_map).to(self.device), i32)
        arr = (ctypes.c_void_p * len(self.maps))(*[m.handle for m in self.maps])
        handle = ctypes.c_void_p()
        nat.call('tds_mapset_create', self.device, arr, len(self.maps), ctypes.byref(handle))
        self._h = handle

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError('StaticMapSet was destroyed')
        return self._h

    def rank_of(self, level):
        return self.maps[0].rank_of(level)

    def face_keys(self):
        """distinct face keys over the maps of the set (None: more than 64)"""
        if not hasattr(self, '_face_keys'):
            buf, n = (ctypes.c_uint32 * 64)(), ctypes.c_int(0)
            nat.call('tds_mapset_keys', self.device, self.handle, ctypes.cast(buf, ctypes.c_void_p), 64, ctypes.byref(n))
            self._face_keys = None if n.value < 0 else [int(buf[i]) for i in range(n.value)]
        return self._face_keys

    def select(self, idx):
        """the same maps for a sub-batch / re-ordered batch"""
        return StaticMapSet(self.maps, self.scene_map[idx])

    def close(self):
        if getattr(self, '_h', None) is not None:
            h, self._h = self._h, None
            nat.call('tds_mapset_destroy', self.device, h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _agents_per_scene(smap, state):
    n_scenes = smap.scene_map.shape[0]
    n = state.numel() // 4
    if state.dim() < 2 or state.shape[0] != n_scenes or n % n_scenes:
        raise RuntimeError(f'a StaticMapSet for {n_scenes} scenes needs agent tensors of shape ({n_scenes}, A, ...), got {tuple(state.shape)}')
    return n // n_scenes


def offroad_forward(smap, state, lenwid, sc, present, threshold):
    state, lenwid, sc = _c(state), _c(lenwid), _c(sc)
    n = state.numel() // 4
    out = torch.empty(state.shape[:-1], dtype=f32, device=state.device)
    pres = None
    if present is not None:
        pres = _c(present, u8) if present.dtype != torch.bool else present.contiguous().view(u8)
    if isinstance(smap, StaticMapSet):
        nat.call('tds_offroad_multi_f32', state.device, smap.handle, nat.dev_ptr(smap.scene_map, i32, 'scene_map'), _agents_per_scene(smap, state),
                 nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(lenwid, f32, 'lenwid'), nat.dev_ptr(sc, f32, 'sc'), nat.dev_ptr(pres, u8, 'present'),
                 nat.dev_ptr(out, f32, 'out'), n, float(threshold), nat.stream_ptr(state.device))
        return out
    nat.call('tds_offroad_f32', state.device, smap.handle, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(lenwid, f32, 'lenwid'),
             nat.dev_ptr(sc, f32, 'sc'), nat.dev_ptr(pres, u8, 'present'), nat.dev_ptr(out, f32, 'out'), n, float(threshold),
             nat.stream_ptr(state.device))
    return out


class _Offroad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, smap, state, lenwid, sc, present, threshold):
        state, lenwid, sc = _c(state), _c(lenwid), _c(sc)
        out = offroad_forward(smap, state, lenwid, sc, present, threshold)
        ctx.save_for_backward(state, lenwid, sc)
        ctx.cfg = (smap, present, threshold)
        return out

    @staticmethod
    def backward(ctx, gout):
        state, lenwid, sc = ctx.saved_tensors
        smap, present, threshold = ctx.cfg
        gout = _c(gout)
        pres = None
        if present is not None:
            pres = _c(present, u8) if present.dtype != torch.bool else present.contiguous().view(u8)
        gs, gl, gsc = torch.empty_like(state), torch.empty_like(lenwid), torch.empty_like(sc)
        if isinstance(smap, StaticMapSet):
            nat.call('tds_offroad_multi_bwd_f32', state.device, smap.handle, nat.dev_ptr(smap.scene_map, i32, 'scene_map'), _agents_per_scene(smap, state),
                     nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(lenwid, f32, 'lenwid'), nat.dev_ptr(sc, f32, 'sc'), nat.dev_ptr(pres, u8, 'present'),
                     nat.dev_ptr(gout, f32, 'grad_out'), nat.dev_ptr(gs, f32, 'gs'), nat.dev_ptr(gl, f32, 'gl'), nat.dev_ptr(gsc, f32, 'gsc'),
                     state.numel() // 4, float(threshold), nat.stream_ptr(state.device))
            return None, gs, gl, gsc, None, None
        nat.call('tds_offroad_bwd_f32', state.device, smap.handle, nat.dev_ptr(state, f32, 'state'), nat.dev_ptr(lenwid, f32, 'lenwid'),
                 nat.dev_ptr(sc, f32, 'sc'), nat.dev_ptr(pres, u8, 'present'), nat.dev_ptr(gout, f32, 'grad_out'), nat.dev_ptr(gs, f32, 'gs'),
                 nat.dev_ptr(gl, f32, 'gl'), nat.dev_ptr(gsc, f32, 'gsc'), state.numel() // 4, float(threshold), nat.stream_ptr(state.device))
        return None, gs, gl, gsc, None, None


def offroad(smap, state, lenwid, threshold=0.5, present=None, sc=None):
    """offroad_infraction_loss (infractions.py:176-229, pure path) [* present]: state (...,4), lenwid (...,2) -> (...)"""
    if sc is None:
        sc = heading_sc(state[..., 2])
    return _Offroad.apply(smap, state, lenwid, sc, present, float(threshold))


# ---------------------------------------------------------------------------------------------------------------
# K3 rasteriser
# ---------------------------------------------------------------------------------------------------------------
#: scratch for the binned fast path of K3, one per (device, stream, cameras, resolution): two renders of one shape on different streams
#: (or threads) of a device never share it.  Never carries state between calls; the least recently used entries are dropped.
_workspaces = collections.OrderedDict()
_WORKSPACES_MAX = 8
use_workspace = True
use_bitplanes = True      # test hook: False forces the packed-key kernels
use_index_slices = True   # test hook: False makes the raster backward read the forward image (the path of the packed-key kernels)


def _raster_workspace(dev, n_img, res, out_mode=None, n_keys=-1):
    """n_keys: distinct keys of the launch (map + actors) when the caller knows them (-1: not known) -- with at most 15 the bit-plane kernels
    run, which above 160 x 160 (float32) / 216 x 216 (uint8) need nothing but their 64 bytes of work queues (128 bytes instead of 2.1 GB at
    B x A = 65 536 cameras of 256 x 256)"""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    n_keys = int(n_keys) if (out_mode is not None and 0 <= n_keys <= 15) else -1
    key = (idx, torch.cuda.current_stream(dev).cuda_stream, threading.get_ident(), n_img, res, out_mode if n_keys >= 0 else None, n_keys >= 0)
    ws = _workspaces.get(key)
    if ws is None:
        n = ctypes.c_int64(0)
        if n_keys >= 0:
            nat.call('tds_raster_scene_workspace_bytes_for', dev, n_img, res, out_mode, n_keys, ctypes.byref(n))
        else:
            nat.call('tds_raster_scene_workspace_bytes', dev, n_img, res, ctypes.byref(n))
        ws = torch.empty(max(int(n.value), 0), dtype=torch.uint8, device=dev) if n.value > 0 else False
        _workspaces[key] = ws
        while len(_workspaces) > _WORKSPACES_MAX:
            _workspaces.popitem(last=False)            # the tensor is freed stream-ordered by the caching allocator
    else:
        _workspaces.move_to_end(key)
    return ws if ws is not False else None


# ---- where the images live --------------------------------------------------------------------------------------------------------
#: The raster launch is bound by the HBM write stream, and what that stream reaches depends on the PHYSICAL pages under the image: about one
#: large hipMalloc in three is served at 7/8 of the rate for as long as it lives, a buffer whose pages are spread out hardly ever is (one of 100 probed in round 4, in the 15/16 class; csrc/alloc.hip,
#: DESIGN_HISTORY.md section 4).  So images of SPREAD_MIN bytes and more are not taken from torch's default pool but from a torch memory pool whose
#: blocks the library builds (tds_torch_alloc / tds_torch_free behind torch.cuda.memory.CUDAPluggableAllocator): the reference-shaped call
#: `render_egocentric()` -- a fresh tensor per call, rendering/cv2.py:52 -- gets such a block, cached and stream-ordered by torch's
#: allocator like any other.  False: plain torch.empty (tests, tools/alloc experiments).
use_image_pool = True
SPREAD_MIN = 256 << 20
_image_pools = {}            # device index -> torch.cuda.MemPool
_pool_allocators = []        # the pluggable allocators behind them: they must outlive every pool, so they live as long as the process


def image_pool(device):
    """The torch memory pool of `device` whose blocks have spread-out physical pages (one per device, created on first use)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _image_pools.get(idx)
    if pool is None:
        nat.lib()                                         # fails loudly when the library is missing
        if not _pool_allocators:
            _pool_allocators.append(torch.cuda.memory.CUDAPluggableAllocator(nat.LIB_PATH, 'tds_torch_alloc', 'tds_torch_free'))
            atexit.register(release_image_pool)           # the pools go before the interpreter takes the allocator apart
        pool = _image_pools[idx] = torch.cuda.MemPool(_pool_allocators[0].allocator())
    return pool


def release_image_pool(device=None) -> None:
    """Drop the image pool(s): their cached blocks go back to the driver once the tensors in them are gone."""
    def _index(d):
        d = torch.device(d)
        return d.index if d.index is not None else torch.cuda.current_device()
    for idx in list(_image_pools) if device is None else [_index(device)]:
        pool = _image_pools.pop(idx, None)
        del pool                                          # the pool dies here, while its allocator is alive


def empty_image(shape, dtype, device) -> torch.Tensor:
    """torch.empty for a rendered image: from the image pool when it is large enough for the placement to matter"""
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * torch.empty((), dtype=dtype).element_size()
    device = torch.device(device)
    if not use_image_pool or nbytes < SPREAD_MIN or device.type != 'cuda' or torch.cuda.is_current_stream_capturing():
        return torch.empty(shape, dtype=dtype, device=device)
    with torch.cuda.use_mem_pool(image_pool(device), device=device):
        return torch.empty(shape, dtype=dtype, device=device)


class _OwnedBuffer:
    """a tds_buffer (csrc/alloc.hip) exposed through __cuda_array_interface__: torch.as_tensor(...) keeps it alive, the buffer goes back
    to the driver when the last tensor over it is gone"""

    def __init__(self, shape, dtype, device, dense=False):
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        n = 1
        for d in shape:
            n *= int(d)
        self.handle = ctypes.c_void_p()
        nat.check(nat.lib().tds_buffer_create(max(n * torch.empty((), dtype=dtype).element_size(), 1), idx, nat.BUFFER_DENSE if dense else 0,
                                              ctypes.byref(self.handle)), 'tds_buffer_create')
        self._lib = nat.lib()
        typestr = {torch.float32: '<f4', torch.uint8: '|u1', torch.int32: '<i4'}[dtype]
        self.__cuda_array_interface__ = dict(shape=tuple(int(d) for d in shape), typestr=typestr, data=(int(self._lib.tds_buffer_ptr(self.handle)), False),
                                             version=2, strides=None)

    def __del__(self, _finalizing=sys.is_finalizing):      # (bound at definition: module globals are gone when the interpreter shuts down)
        h, self.handle = getattr(self, 'handle', None), None
        if h and not _finalizing():                        # at interpreter shutdown the driver takes the memory back itself

            try:
                torch.cuda.synchronize(self.device)            # nothing in flight may still write to pages that are about to be unmapped
            except Exception:                                  # noqa: BLE001 -- interpreter shutdown
                pass
            self._lib.tds_buffer_destroy(h)


def owned_image(shape, dtype=torch.float32, device="cuda", dense=False) -> torch.Tensor:
    """A caller-owned image buffer with spread-out physical pages that does NOT go through torch's caching allocator (long-lived output
    rings: rendering.allocate_image_ring); freed when the tensor and all its views are gone."""
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    return torch.as_tensor(_OwnedBuffer(shape, dtype, device, dense=dense), device=device)


# ---- streams confined to a part of the CUs ---------------------------------------------------------------------------------------------
_reserved_streams = {}       # (device index, CUs per XCD) -> (raster stream, metric stream, handles kept alive)


def check_reserved_layout(raster_places, metric_places, per_xcd: int, cus: int):
    """Is what the two masked streams REALLY run on the layout `overlap_infractions = 'reserved'` relies on?  `*_places`: the distinct
    XCC_ID << 16 | SE/SH/CU words that tds_stream_places reported for the raster stream and for the metric stream.  Wanted: the metric
    stream on exactly `per_xcd` CUs of EVERY XCD, the raster stream on all the others, no CU in both, at least 64 CUs in all (a CPX partition
    of 32 CUs has nothing to give away).  -> (ok, reason).  Pure host logic: tests/test_host_logic.py feeds it good, shifted and small layouts."""
    raster, metric = set(int(p) for p in raster_places), set(int(p) for p in metric_places)
    if cus < 64:
        return False, f'the device reports {cus} CUs: too few to keep {per_xcd} per XCD free of the raster launch'
    if not raster or not metric:
        return False, 'a masked stream ran nowhere'
    if raster & metric:
        return False, f'{len(raster & metric)} CUs serve both streams: the mask bits do not mean what they mean on an MI355X in SPX mode'
    xcds = sorted({p >> 16 for p in raster | metric})
    per = {x: sum(1 for p in metric if p >> 16 == x) for x in xcds}
    if any(n != per_xcd for n in per.values()):
        return False, f'the metric stream has {per} CUs per XCD instead of {per_xcd} on each'
    if len(raster) + len(metric) != cus:
        return False, f'the two streams cover {len(raster) + len(metric)} of {cus} CUs'
    return True, 'ok'


def stream_places(stream, n: int = 8192):
    """the distinct places (XCC_ID << 16 | SE/SH/CU) the kernels of a torch stream run on (tds_stream_places; synchronises)"""
    dev = stream.device
    out = torch.full((n,), -1, dtype=i32, device=dev)
    stream.wait_stream(torch.cuda.current_stream(dev))
    nat.call('tds_stream_places', dev, ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(out.data_ptr()), n)      # (with `dev` current)
    stream.synchronize()
    return sorted(set(out.cpu().tolist()))


def reserved_streams(device, per_xcd: int = 4):
    """(raster_stream, metric_stream) of `device`: two torch.cuda.ExternalStream over tds_stream_create -- the first may use every CU but
    `per_xcd` per XCD, the second ONLY those (mask bit i is CU i / 8 of XCD i % 8 on MI355X, tools/cu_mask_probe.hip).  The persistent raster
    launch holds every CU it may use until its last image is out; kept off 32 of the 256 it loses nothing (it is bound by the write stream)
    and the metric kernels have somewhere to run beside it (Simulator.overlap_infractions = 'reserved').  Created once per device; the
    layout is VERIFIED on the device when the streams are made (reserved_layout_ok)."""
    ent = _reserved_entry(device, per_xcd)
    if ent[0] is None:
        raise RuntimeError(f"no CU-masked streams on this device: {ent[3][1]}")
    return ent[0], ent[1]


def reserved_layout_ok(device, per_xcd: int = 4):
    """(ok, reason): can `overlap_infractions = 'reserved'` be used on this device?  The mask layout is an observation of one part in one
    partition mode; here a probe kernel on each of the two streams reports where it really ran (tds_stream_places) and check_reserved_layout
    judges it.  Never raises: a device too small for the masks answers (False, why)."""
    return _reserved_entry(device, per_xcd)[3]


def _reserved_entry(device, per_xcd):
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _reserved_streams.get((idx, per_xcd))
    if ent is not None and ent[0] is not None and ent[3][1].startswith('not verified') and not torch.cuda.is_current_stream_capturing():
        # the streams were made under stream capture, where no probe can run: judged now, at the first call outside capture
        try:
            with torch.cuda.device(idx):
                cus = ctypes.c_int(0)
                nat.check(nat.lib().tds_device_cu_count(idx, ctypes.byref(cus)), 'tds_device_cu_count')
                verdict = check_reserved_layout(stream_places(ent[0]), stream_places(ent[1]), per_xcd, cus.value)
        except Exception as exc:                                     # noqa: BLE001 -- 'never raises': the mode falls back with one warning
            verdict = (False, f'the probe of the CU-masked streams failed: {exc}')
        ent = _reserved_streams[(idx, per_xcd)] = (ent[0], ent[1], ent[2], verdict)
    if ent is None:
        try:
            with torch.cuda.device(idx):
                ent = _make_reserved_entry(idx, per_xcd)
        except Exception as exc:                                     # noqa: BLE001 -- reserved_layout_ok / Simulator._reserved_usable promise not to raise
            ent = (None, None, [], (False, f'creating or probing the CU-masked streams failed: {exc}'))
        _reserved_streams[(idx, per_xcd)] = ent
    return ent


def _make_reserved_entry(idx, per_xcd):
    """(raster stream, metric stream, handles, (ok, reason)) of device `idx` (current); (None, None, [], (False, why)) on a device too small"""
    L = nat.lib()
    cus = ctypes.c_int(0)
    nat.check(L.tds_device_cu_count(idx, ctypes.byref(cus)), 'tds_device_cu_count')
    n_words = (cus.value + 31) // 32
    reserved = [0] * n_words
    for bit in range(min(8 * per_xcd, cus.value)):
        reserved[bit // 32] |= 1 << (bit % 32)
    every = [0] * n_words
    for bit in range(cus.value):
        every[bit // 32] |= 1 << (bit % 32)
    rest = [e & ~r for e, r in zip(every, reserved)]
    if cus.value < 64 or not any(rest) or not any(reserved):
        return None, None, [], (False, f'the device reports {cus.value} CUs: too few to keep {per_xcd} per XCD free of the raster launch')
    handles = []
    for mask in (rest, reserved):
        arr = (ctypes.c_uint32 * n_words)(*mask)
        h = ctypes.c_void_p()
        nat.check(L.tds_stream_create(idx, ctypes.cast(arr, ctypes.c_void_p), n_words, ctypes.byref(h)), 'tds_stream_create')
        handles.append(h)
    rs = torch.cuda.ExternalStream(handles[0].value, device=torch.device('cuda', idx))
    ms = torch.cuda.ExternalStream(handles[1].value, device=torch.device('cuda', idx))
    if torch.cuda.is_current_stream_capturing():
        verdict = (True, 'not verified: created under stream capture')          # judged at the first call outside capture (_reserved_entry)
    else:
        verdict = check_reserved_layout(stream_places(rs), stream_places(ms), per_xcd, cus.value)
    return rs, ms, handles, verdict


#: set to a list to have raster_scene append (start, end) torch.cuda.Event pairs recorded around every kernel launch
raster_events = None
#: the same for the launches of the raster backward kernel
raster_bwd_events = None
#: calls of raster_scene so far in this process (bench.py reports which of them its timed region was, so that a kernel trace of the run can
#: be cut to exactly those launches: tools/make_profiles.py)
raster_calls = 0


def raster_scene(smap, state, agent_sc, tmpl, actor_key, mask, cam_xy, cam_sc, fov, res, out_dtype=torch.float32, out=None, key_table=None,
                 extra_tri=None, extra_key=None, index_slices=False, trim=True):
    """Fused Simulator.render: state (B,N,4), agent_sc (B,N,2), tmpl (B,N,7,2), actor_key (B,N,2) int32 bit patterns -- or (B,Nc,N,2)
    when every camera sees its own colours (custom_agent_colors) --, mask (B,Nc,N) bool/uint8, cam_xy / cam_sc (B,Nc,2)
    -> (B,Nc,3,res,res) float32 [0,255] or uint8.  extra_tri (B,Nc,K,3,2) world-space triangles with keys extra_key (B,Nc,K) int32
    (0 = none) are drawn per camera (waypoint discs); their keys belong into `key_table` too.
    index_slices=True: returns (image, slices, keys) -- the per-pixel key index as bit-slices (int32 tensor, layout in include/tdship.h)
    and the ascending key table of the launch, what the backward pass reads instead of the image; (image, None, None) when the call
    cannot be served by the bit-plane kernel."""
    B, Nc = cam_xy.shape[:2]
    N = state.shape[1]
    dev = cam_xy.device
    cam_xy, cam_sc = _c(cam_xy), _c(cam_sc)
    if N > 0:
        state, agent_sc, tmpl = _c(state), _c(agent_sc), _c(tmpl)
        actor_key = _c(actor_key, i32)
        assert tuple(actor_key.shape) in ((B, N, 2), (B, Nc, N, 2)), 'actor_key must be (B,N,2) or (B,Nc,N,2)'
        mask = mask.contiguous().view(u8) if mask.dtype == torch.bool else _c(mask, u8)
    assert out_dtype in (torch.float32, torch.uint8)
    if out is None:
        out = empty_image((B, Nc, 3, res, res), out_dtype, dev)
    else:
        # a real error, not an assert (python -O strips those), on anything the kernel's 16-byte non-temporal stores cannot take
        if tuple(out.shape) != (B, Nc, 3, res, res) or out.dtype != out_dtype or not out.is_contiguous():
            raise RuntimeError(f'`out` must be a contiguous {out_dtype} tensor of shape {(B, Nc, 3, res, res)}, got {out.dtype} {tuple(out.shape)}'
                               f'{"" if out.is_contiguous() else " (not contiguous)"}')
        if out.device != dev:
            raise RuntimeError(f'`out` is on {out.device}, the cameras on {dev}')
        if out.data_ptr() % 16 != 0:
            raise RuntimeError('`out` must be 16-byte aligned (a view with an odd storage offset is not)')
    p = lambda t, d, nme: nat.dev_ptr(t, d, nme) if N > 0 else None
    mode = nat.OUT_F32 if out_dtype == torch.float32 else nat.OUT_U8
    # distinct actor keys (host side): enables the bit-plane kernel.  Callers that know them (Simulator) pass `key_table`;
    # otherwise they are read back from the device tensor (a synchronisation -- fine for tests and one-off calls)
    K = 0
    if extra_tri is not None and extra_tri.shape[2] > 0:
        K = extra_tri.shape[2]
        assert tuple(extra_tri.shape) == (B, Nc, K, 3, 2) and tuple(extra_key.shape) == (B, Nc, K), 'extra_tri must be (B,Nc,K,3,2), extra_key (B,Nc,K)'
        extra_tri, extra_key = _c(extra_tri), _c(extra_key, i32)
    if key_table is None and (N > 0 or K > 0) and use_bitplanes:
        parts = ([actor_key.flatten()] if N > 0 else []) + ([extra_key.flatten()] if K > 0 else [])
        key_table = [v for v in torch.unique(torch.cat(parts)).cpu().tolist() if v != 0]
    kt = None
    if key_table is not None and use_bitplanes:
        vals = [int(v) & 0xffffffff for v in key_table]
        kt = (ctypes.c_uint32 * max(len(vals), 1))(*vals)
    n_keys = -1
    if kt is not None:
        n_keys = len(set(vals) | set(smap.face_keys())) if smap.face_keys() is not None else -1
    ws = _raster_workspace(dev, B * Nc, int(res), mode, n_keys) if use_workspace else None
    ev = None
    if raster_events is not None:          # bench.py: HIP events on the launch stream, right around the kernel
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record(torch.cuda.current_stream(dev))
    multi = isinstance(smap, StaticMapSet)
    if multi and smap.scene_map.shape[0] != B:
        raise RuntimeError(f'the StaticMapSet is for {smap.scene_map.shape[0]} scenes, the cameras for {B}')
    aux = slices = None
    if index_slices and kt is not None and out_dtype == torch.float32 and res % 4 == 0 and B * Nc > 0:
        nbytes = ctypes.c_int64(0)
        nat.call('tds_raster_index_slices_bytes', dev, B * Nc, int(res), ctypes.byref(nbytes))
        slices = torch.empty(nbytes.value // 4, dtype=i32, device=dev)
        aux = nat.RasterAux(index_slices=slices.data_ptr(), index_slices_bytes=nbytes.value)
    if not trim:
        aux = aux if aux is not None else nat.RasterAux()
        aux.flags = nat.RASTER_NO_TRIM
    head = ('tds_raster_scene_multi', dev, smap.handle, nat.dev_ptr(smap.scene_map, i32, 'scene_map')) if multi else ('tds_raster_scene', dev, smap.handle)
    def launch(aux):
        nat.call(*head, p(state, f32, 'state'), p(agent_sc, f32, 'agent_sc'), p(tmpl, f32, 'tmpl'),
             p(actor_key, i32, 'actor_key'), p(mask, u8, 'mask'), nat.dev_ptr(cam_xy, f32, 'cam_xy'), nat.dev_ptr(cam_sc, f32, 'cam_sc'),
             B, Nc, N, float(2.0 / fov), int(res), mode, nat.dev_ptr(out, out_dtype, 'out'),
             None if ws is None else ctypes.c_void_p(ws.data_ptr()), 0 if ws is None else ws.numel(),
             None if kt is None else ctypes.cast(kt, ctypes.c_void_p), 0 if kt is None else len(key_table),
             1 if (N > 0 and actor_key.dim() == 4) else 0,
             nat.dev_ptr(extra_tri, f32, 'extra_tri') if K > 0 else None, nat.dev_ptr(extra_key, i32, 'extra_key') if K > 0 else None, K,
             None if aux is None else ctypes.cast(ctypes.pointer(aux), ctypes.c_void_p), nat.stream_ptr(dev))

    global raster_calls
    raster_calls += 1
    try:
        launch(aux)
    except nat.TdsError as e:
        # only a documented capacity (more than 15 keys, planes too large for the bit-plane kernel: TDS_ELIMIT) sends the call to the kernels
        # that produce no slices; a bad argument (TDS_EINVAL: wrong byte count, wrong dtype ...) or a HIP failure is the caller's to see
        if aux is None or slices is None or e.code != nat.E_LIMIT:
            raise
        slices = None
        aux = None if trim else nat.RasterAux(flags=nat.RASTER_NO_TRIM)
        launch(aux)
    if ev is not None:
        ev[1].record(torch.cuda.current_stream(dev))
        raster_events.append(ev)
    if index_slices:
        if slices is None or aux is None or aux.n_keys == 0:
            return out, None, None
        return out, slices, [int(aux.keys[i]) for i in range(aux.n_keys)]
    return out


class _RasterScene(torch.autograd.Function):
    """Differentiable wrapper of the fused scene rasteriser.  Forward: tds_raster_scene (CV2 pixel semantics).  Backward: the build-defined
    edge-sampling gradient with respect to actor position / heading / template vertices (sizes) and camera position / heading (DESIGN.md "K3 backward"); the
    reference's CV2 backend has none (rendering/cv2.py:27-70).  When the bit-plane kernel serves the forward it also leaves the per-pixel key
    index as bit-slices (3 % of the image), and the backward (tds_raster_scene_bwd_idx_f32) reads those and the incoming gradient next
    to colour boundaries only; otherwise the forward image is kept and tds_raster_scene_bwd_f32 streams image and gradient in full."""

    @staticmethod
    def forward(ctx, state, agent_sc, cam_xy, cam_sc, key_colors, smap, tmpl, actor_key, mask, fov, res, key_table, extra_tri, extra_key, color_keys, trim,
                ego=0):
        # ego = Nc > 0: the cameras ARE the first Nc agents (render_egocentric): positions and headings are taken from `state` / `agent_sc` here,
        # and the backward folds the cameras' gradient into the agents' -- no slice nodes in the graph, no separate gradient tensors
        ctx.ego = int(ego)
        if ctx.ego:
            cam_xy, cam_sc = state[:, :ctx.ego, :2].contiguous(), agent_sc[:, :ctx.ego].contiguous()
        out, slices, keys = raster_scene(smap, state, agent_sc, tmpl, actor_key, mask, cam_xy, cam_sc, fov, res, key_table=key_table,
                                         extra_tri=extra_tri, extra_key=extra_key, index_slices=True, trim=trim) if use_index_slices else \
            (raster_scene(smap, state, agent_sc, tmpl, actor_key, mask, cam_xy, cam_sc, fov, res, key_table=key_table, extra_tri=extra_tri,
                          extra_key=extra_key, trim=trim), None, None)
        if slices is not None:
            ctx.save_for_backward(state, agent_sc, cam_xy, cam_sc, tmpl, mask, slices)
        else:
            ctx.save_for_backward(state, agent_sc, cam_xy, cam_sc, tmpl, mask, out)
        ctx.fov, ctx.res, ctx.keys = fov, res, keys
        ctx.color_keys = None
        if key_colors is not None:
            if keys is None:
                raise RuntimeError('colour gradients need the key-index slices of the bit-plane kernel (at most 15 distinct keys, float32 output, '
                                   'resolution a multiple of 4, no per-camera colours)')
            assert len(color_keys) == key_colors.shape[0] and key_colors.shape[-1] == 3, 'key_colors is (len(color_keys), 3)'
            ctx.color_keys = [int(k) & 0xffffffff for k in color_keys]
        return out

    @staticmethod
    def backward(ctx, gout):
        state, agent_sc, cam_xy, cam_sc, tmpl, mask, kept = ctx.saved_tensors
        B, Nc = cam_xy.shape[:2]
        N = state.shape[1]
        dev = cam_xy.device
        # a gradient that is one image broadcast over the cameras (image.sum(), a fixed linear read-out ...) stays one image
        gstride = 3 * int(ctx.res) * int(ctx.res)
        if ctx.keys is not None and gout.dim() == 5 and gout.stride(0) == 0 and gout.stride(1) == 0 and B * Nc > 1:
            gout, gstride = _c(gout[0, 0]), 0
        else:
            gout = _c(gout)
        g_agent = torch.empty((B, Nc, max(N, 1), 4), dtype=f32, device=dev)
        g_cam = torch.empty((B, Nc, 4), dtype=f32, device=dev)
        m8 = mask.contiguous().view(u8) if mask.dtype == torch.bool else _c(mask, u8)
        p = lambda t, d, nme: nat.dev_ptr(_c(t, d), d, nme) if N > 0 else None
        ev = None
        if raster_bwd_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(dev))
        poses = (p(state, f32, 'state'), p(agent_sc, f32, 'agent_sc'), p(tmpl, f32, 'tmpl'), None if N == 0 else nat.dev_ptr(m8, u8, 'mask'),
                 nat.dev_ptr(_c(cam_xy), f32, 'cam_xy'), nat.dev_ptr(_c(cam_sc), f32, 'cam_sc'))
        g_tmpl = torch.empty((B, Nc, N, 7, 2), dtype=f32, device=dev) if (N > 0 and ctx.needs_input_grad[6]) else None      # template vertices: actor sizes
        tail = (B, Nc, N, float(2.0 / ctx.fov), int(ctx.res), nat.dev_ptr(g_agent, f32, 'grad_agent'), nat.dev_ptr(g_cam, f32, 'grad_cam'),
                nat.dev_ptr(g_tmpl, f32, 'grad_tmpl'), nat.stream_ptr(dev))
        g_color = None
        if ctx.keys is not None:
            kt = (ctypes.c_uint32 * 16)(*ctx.keys)
            if ctx.color_keys is not None and ctx.needs_input_grad[4]:
                g_color = torch.empty((B, Nc, 16, 4), dtype=f32, device=dev)
            nat.call('tds_raster_scene_bwd_idx_f32', dev, *poses, nat.dev_ptr(kept, i32, 'index_slices'), ctypes.cast(kt, ctypes.c_void_p), len(ctx.keys),
                     nat.dev_ptr(gout, f32, 'grad_out'), gstride, *tail[:-2], nat.dev_ptr(g_color, f32, 'grad_color'), tail[-2], tail[-1])
        else:
            nat.call('tds_raster_scene_bwd_f32', dev, *poses, nat.dev_ptr(kept, f32, 'image'), nat.dev_ptr(gout, f32, 'grad_out'), *tail)
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(dev))
            raster_bwd_events.append(ev)
        ga = g_agent.sum(dim=1) if N > 0 else None                   # over cameras: (B, N, 4)
        g_state = g_sc = None
        if N > 0:
            if ctx.ego:
                ga[:, :ctx.ego] += g_cam                               # the camera of agent a is agent a
            g_state = torch.cat([ga[..., :2], _zeros_const(ga.shape[:-1] + (2,), dev)], dim=-1)
            g_sc = ga[..., 2:].contiguous()
        g_key_colors = None
        if g_color is not None:
            # rows of the launch's key table (index i + 1 <-> keys[i]) -> the caller's rows; a key that was not part of the launch shows nowhere
            per_key = g_color.sum(dim=(0, 1))[:, :3]                  # over cameras: (16, 3)
            pos = {k: i + 1 for i, k in enumerate(ctx.keys)}
            rows = torch.tensor([pos.get(k, -1) for k in ctx.color_keys], device=dev)
            g_key_colors = torch.where((rows >= 0)[:, None], per_key[rows.clamp(min=0)], torch.zeros((), device=dev))
        return (g_state, g_sc, None if ctx.ego else g_cam[..., :2].contiguous(), None if ctx.ego else g_cam[..., 2:].contiguous(), g_key_colors,
                None, None if g_tmpl is None else g_tmpl.sum(dim=1), None, None, None, None, None, None, None, None, None, None)


def raster_scene_diff(smap, state, agent_sc, tmpl, actor_key, mask, cam_xy, cam_sc, fov, res, key_table=None, extra_tri=None, extra_key=None,
                      key_colors=None, color_keys=None, trim=True, ego_cameras=0):
    """raster_scene with a backward pass (float32 output only; the per-camera triangles get no gradient).
    key_colors (K,3) float tensor + color_keys (K packed keys): a handle for COLOUR gradients -- row r stands for the colour the image shows
    where key color_keys[r] wins; the forward takes its pixels from the keys' own RGB bits (the caller keeps the two consistent), the
    backward returns d loss / d key_colors[r] = the sum of the incoming gradient over those pixels, all cameras (exact)."""
    if ego_cameras:
        # `ego_cameras` = Nc: the cameras are the first Nc agents of `state` / `agent_sc` (cam_xy / cam_sc are ignored)
        cam_xy = cam_sc = None
    return _RasterScene.apply(state, agent_sc, cam_xy, cam_sc, key_colors, smap, tmpl, actor_key, mask, float(fov), int(res), key_table, extra_tri,
                              extra_key, color_keys, bool(trim), int(ego_cameras))


def raster_mesh(verts, attrs, faces, cam_xy, cam_sc, levels, scale, res, out_dtype=torch.float32, trim=True):
    """Generic render_rgb_mesh: verts (n,V,3), attrs (n,V,3), faces (n,F,3) -> (n,3,res,res)"""
    n = cam_xy.shape[0]
    dev = cam_xy.device
    verts, attrs, cam_xy, cam_sc = _c(verts), _c(attrs), _c(cam_xy), _c(cam_sc)
    faces = _c(faces, i32)
    out = torch.empty((n, 3, res, res), dtype=out_dtype, device=dev)
    lv = (ctypes.c_float * max(len(levels), 1))(*[float(x) for x in levels])
    nat.call('tds_raster_mesh', dev, nat.dev_ptr(verts, f32, 'verts'), nat.dev_ptr(attrs, f32, 'attrs'), nat.dev_ptr(faces, i32, 'faces'),
             n, verts.shape[1], faces.shape[1], nat.dev_ptr(cam_xy, f32, 'cam_xy'), nat.dev_ptr(cam_sc, f32, 'cam_sc'),
             ctypes.cast(lv, ctypes.c_void_p), len(levels), float(scale), int(res),
             nat.OUT_F32 if out_dtype == torch.float32 else nat.OUT_U8, nat.dev_ptr(out, out_dtype, 'out'), 0 if trim else nat.RASTER_NO_TRIM,
             nat.stream_ptr(dev))
    return out


# ---------------------------------------------------------------------------------------------------------------
# lane tables and the wrong-way query (csrc/lanes.hip)
# ---------------------------------------------------------------------------------------------------------------
class LaneTableHandle:
    """Device-resident lane table of one map (tds_lanes_t), built from the HOST arrays of `lanelet2.lane_table`."""

    def __init__(self, table, device='cuda', max_tolerance=1.0, cell_size=0.0):
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('lane tables live on an MI355X; there is no CPU implementation')
        self.max_tolerance = float(max_tolerance)
        self.n_lanelets = int(len(table.flags))
        poly = np.ascontiguousarray(table.poly_xy, np.float64)
        cl = np.ascontiguousarray(table.cl_xyz, np.float64)
        ps, cs = np.ascontiguousarray(table.poly_start, np.int32), np.ascontiguousarray(table.cl_start, np.int32)
        fl = np.ascontiguousarray(table.flags, np.int32)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        handle = ctypes.c_void_p()
        nat.call('tds_lanes_create', self.device, vp(poly), vp(ps), vp(cl), vp(cs), vp(fl), self.n_lanelets, float(cell_size),
                 self.max_tolerance, ctypes.byref(handle))
        self._h = handle

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError('LaneTableHandle was destroyed')
        return self._h

    def info(self):
        buf = (ctypes.c_int64 * 4)()
        nat.call('tds_lanes_info', self.device, self.handle, buf)
        return dict(lanelets=buf[0], nx=buf[1], ny=buf[2], bytes=buf[3])

    def close(self):
        if getattr(self, '_h', None) is not None:
            h, self._h = self._h, None
            nat.call('tds_lanes_destroy', self.device, h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LaneTableSet:
    """The lane tables of a batch (tds_laneset_t) + which of them every scene uses (`scene_map`, (B,) int32 on the device, -1 = the
    scene has no lane map; None = every scene uses table 0)."""

    def __init__(self, tables, scene_map=None):
        assert len(tables) > 0
        self.tables = list(tables)                   # keeps the tables alive
        self.device = self.tables[0].device
        self.scene_map = None if scene_map is None else _c(torch.as_tensor(scene_map).to(self.device), i32)
        arr = (ctypes.c_void_p * len(self.tables))(*[t.handle for t in self.tables])
        handle = ctypes.c_void_p()
        nat.call('tds_laneset_create', self.device, arr, len(self.tables), ctypes.byref(handle))
        self._h = handle

    @property
    def handle(self):
        if self._h is None:
            raise RuntimeError('LaneTableSet was destroyed')
        return self._h

    def close(self):
        if getattr(self, '_h', None) is not None:
            h, self._h = self._h, None
            nat.call('tds_laneset_destroy', self.device, h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def wrong_way(lane_set, state, recenter_offset, present, direction_angle_threshold, lanelet_dist_tolerance):
    """lanelet_orientation_loss (infractions.py:232-304) [* present]: state (B,A,4) -> (B,A) float32.  No gradient, as in the
    reference (the lane directions come from host floats there)."""
    state = _c(state.detach())
    if state.dim() != 3 or state.shape[-1] != 4:
        raise RuntimeError(f'wrong_way: state must be (B,A,4), got {tuple(state.shape)}')
    B, A = state.shape[:2]
    out = torch.empty((B, A), dtype=f32, device=state.device)
    if B * A == 0:
        return out
    if lane_set.scene_map is not None and lane_set.scene_map.shape[0] != B:
        raise RuntimeError(f'wrong_way: the lane-table set was made for {lane_set.scene_map.shape[0]} scenes, the state has {B}')
    off = None if recenter_offset is None else _c(recenter_offset.detach())
    pres = None
    if present is not None:
        pres = present.contiguous().view(u8) if present.dtype == torch.bool else _c(present, u8)
    nat.call('tds_wrong_way_f32', state.device, lane_set.handle, None if lane_set.scene_map is None else nat.dev_ptr(lane_set.scene_map, i32, 'scene_map'),
             A, nat.dev_ptr(state, f32, 'state'), None if off is None else nat.dev_ptr(off, f32, 'recenter_offset'),
             None if pres is None else nat.dev_ptr(pres, u8, 'present'), nat.dev_ptr(out, f32, 'out'), B * A,
             float(direction_angle_threshold), float(lanelet_dist_tolerance), nat.stream_ptr(state.device))
    return out


def lanelet_directions(tables, scene_map, points, lanelet_dist_tolerance, max_dirs=16):
    """find_lanelet_directions for points (n,2) float64 on the device: (dirs (n,max_dirs) f64, dists (n,max_dirs) f64, count (n) i32,
    status (n) u8: bit 0 find_direction failed, bit 1 excluded tag)"""
    f64 = torch.float64
    lane_set = tables if isinstance(tables, LaneTableSet) else LaneTableSet(tables, scene_map)
    points = _c(points, f64)
    n = points.shape[0]
    dev = points.device
    dirs = torch.zeros((n, max_dirs), dtype=f64, device=dev)
    dists = torch.zeros((n, max_dirs), dtype=f64, device=dev)
    count = torch.zeros((n,), dtype=i32, device=dev)
    status = torch.zeros((n,), dtype=u8, device=dev)
    pps = 1 if lane_set.scene_map is None else max(1, n // max(1, lane_set.scene_map.shape[0]))
    nat.call('tds_lanelet_directions_f64', dev, lane_set.handle, None if lane_set.scene_map is None else nat.dev_ptr(lane_set.scene_map, i32, 'scene_map'),
             pps, nat.dev_ptr(points, f64, 'points'), nat.dev_ptr(dirs, f64, 'dirs'), nat.dev_ptr(dists, f64, 'dists'),
             nat.dev_ptr(count, i32, 'count'), nat.dev_ptr(status, u8, 'status'), max_dirs, n, float(lanelet_dist_tolerance),
             nat.stream_ptr(dev))
    return dirs, dists, count, status
