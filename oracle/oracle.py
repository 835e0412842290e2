"""
ctypes front-end of the CPU oracle (oracle/tds_oracle.c).  TEST INFRASTRUCTURE ONLY -- see the header
of tds_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
All arrays are numpy, float32 / int32 / uint8, C-contiguous.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libtds_oracle.so')
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i32 = ctypes.POINTER(ctypes.c_int32)
c_i64 = ctypes.POINTER(ctypes.c_int64)
c_u8 = ctypes.POINTER(ctypes.c_uint8)
c_i8 = ctypes.POINTER(ctypes.c_int8)


def build(force=False):
    src = os.path.join(_HERE, 'tds_oracle.c')
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s'] + (['-B'] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_render_rgb_mesh_one.restype = ctypes.c_int64
        if 'OMP_NUM_THREADS' not in os.environ:
            _lib.orc_set_num_threads(ctypes.c_int(usable_cpus()))
    return _lib


def usable_cpus():
    """The CPUs this process may use: its affinity mask cut to the CPU quota of its cgroup.  OpenMP's default is one thread per logical CPU,
    and a box that shows 256 of them but grants 16 CPUs' worth of time runs 256 threads four times slower than 32."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q, p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()), int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _fl(x):
    return ctypes.c_float(float(x))


def _i64(x):
    return ctypes.c_int64(int(x))


MAX_ACC = 5.0
MAX_STEER = float(np.float32(np.pi / 2))

# rendering/base.py:234-292 (data tables restated)
DEFAULT_LEVELS = dict(direction=2, ego=3, vehicle=4, bicycle=5, pedestrian=6, map_boundary=7, goal_waypoint=8,
                      ground_truth=9, prediction=10, traffic_light=11, traffic_light_green=11, traffic_light_yellow=11,
                      traffic_light_red=11, stop_sign=11, yield_sign=11, left_lane=12, joint_lane=13, right_lane=14, road=15)
DEFAULT_COLORS = dict(background=(0, 0, 0), road=(155, 155, 155), corridor=(0, 155, 0), ego=(255, 0, 0), vehicle=(32, 74, 135),
                      bicycle=(24, 104, 225), pedestrian=(173, 127, 168), ground_truth=(196, 188, 165), prediction=(255, 155, 0),
                      left_lane=(80, 127, 86), right_lane=(128, 0, 128), joint_lane=(255, 255, 255), direction=(100, 255, 255),
                      rear_lights=(255, 255, 0), map_boundary=(255, 255, 0), traffic_light_green=(81, 179, 100),
                      traffic_light_yellow=(240, 189, 39), traffic_light_red=(224, 53, 49), yield_sign=(210, 125, 45),
                      stop_sign=(72, 60, 50), goal_waypoint=(139, 64, 0))


def bicycle_step(state, action, lr, dt=0.1, max_acc=MAX_ACC, max_steer=MAX_STEER, left_handed=False, no_reversing=False):
    state, action, lr = _f(state), _f(action), _f(lr)
    out = np.empty_like(state)
    fn = lib().orc_bicycle_norev_step if no_reversing else lib().orc_bicycle_step
    fn(_p(state, c_f), _p(action, c_f), _p(lr, c_f), _p(out, c_f), _i64(lr.size), _fl(dt), _fl(max_acc), _fl(max_steer),
       ctypes.c_int(int(left_handed)))
    return out


def simple_step(state, action, dt=0.1, max_dx=20.0, max_dpsi=10 * np.pi, max_dv=5.0, oriented=False):
    state, action = _f(state), _f(action)
    out = np.empty_like(state)
    norm = np.array([max_dx, max_dx, max_dpsi, max_dv], dtype=np.float32)
    lib().orc_simple_step(_p(state, c_f), _p(action, c_f), _p(out, c_f), _i64(state.size // 4), _fl(dt), _p(norm, c_f),
                          ctypes.c_int(int(oriented)))
    return out


def bicycle_fit_action(future, current, dt=0.1, max_acc=MAX_ACC, max_steer=MAX_STEER, left_handed=False):
    future, current = _f(future), _f(current)
    out = np.empty(future.shape[:-1] + (2,), dtype=np.float32)
    lib().orc_bicycle_fit_action(_p(future, c_f), _p(current, c_f), _p(out, c_f), _i64(future.size // 4), _fl(dt), _fl(max_acc),
                                 _fl(max_steer), ctypes.c_int(int(left_handed)))
    return out


def bicycle_by_displacement_step(state, action, lr, dt=None, model_dt=0.1, max_dx=20.0, oriented=False):
    """BicycleByDisplacement / BicycleByOrientedDisplacement.step (kinematic.py:526-587): the displacement action becomes a target
    position, KinematicBicycle.fit_action (made with the model's own dt, :556) turns it into a bicycle action, KinematicBicycle.step
    applies it with `dt`.  float32 throughout, the reference's operation order."""
    state, action, lr = _f(state), _f(action), _f(lr)
    dt = model_dt if dt is None else dt
    xy = action * np.float32(max_dx)
    if oriented:                                   # utils.rotate: [[c, -s], [s, c]] @ v, products rounded separately
        c, s_ = np.cos(state[..., 2]), np.sin(state[..., 2])
        xy = np.stack([c * xy[..., 0] + (-s_) * xy[..., 1], s_ * xy[..., 0] + c * xy[..., 1]], -1).astype(np.float32)
    target = state.copy()
    target[..., 0] = state[..., 0] + xy[..., 0] * np.float32(dt)
    target[..., 1] = state[..., 1] + xy[..., 1] * np.float32(dt)
    return bicycle_step(state, bicycle_fit_action(target, state, dt=model_dt), lr, dt=dt)


def box2corners(box5, sc=None):
    box5, sc = _f(box5), _f(sc)
    out = np.empty(box5.shape[:-1] + (4, 2), dtype=np.float32)
    lib().orc_box2corners(_p(box5, c_f), _p(sc, c_f), _p(out, c_f), _i64(box5.size // 5))
    return out


def iou_pairs(box1, box2, sc1=None, sc2=None, debug=False):
    box1, box2, sc1, sc2 = _f(box1), _f(box2), _f(sc1), _f(sc2)
    n = box1.size // 5
    out = np.empty(box1.shape[:-1], dtype=np.float32)
    idx = np.empty((n, 9), dtype=np.int32) if debug else None
    nv = np.empty(n, dtype=np.int8) if debug else None
    area = np.empty(n, dtype=np.float32) if debug else None
    lib().orc_iou_pairs(_p(box1, c_f), _p(sc1, c_f), _p(box2, c_f), _p(sc2, c_f), _p(out, c_f), _i64(n), _p(idx, c_i32), _p(nv, c_i8),
                        _p(area, c_f))
    return (out, idx, nv, area) if debug else out


def discs_pairs(box1, box2, sc1=None, sc2=None, num_discs=5):
    box1, box2, sc1, sc2 = _f(box1), _f(box2), _f(sc1), _f(sc2)
    out = np.empty(box1.shape[:-1], dtype=np.float32)
    if num_discs != 5:
        lib().orc_discs_pairs_n(_p(box1, c_f), _p(sc1, c_f), _p(box2, c_f), _p(sc2, c_f), _p(out, c_f), _i64(box1.size // 5), ctypes.c_int(num_discs))
        return out
    lib().orc_discs_pairs(_p(box1, c_f), _p(sc1, c_f), _p(box2, c_f), _p(sc2, c_f), _p(out, c_f), _i64(box1.size // 5))
    return out


def collision(boxes, present, n_exposed=None, metric='iou', sc=None):
    """boxes B x N x 5 [x,y,len,wid,psi], present B x N; returns B x A (A = n_exposed or N)."""
    boxes, sc = _f(boxes), _f(sc)
    present = np.ascontiguousarray(present, dtype=np.uint8)
    B, N = boxes.shape[:2]
    A = N if n_exposed is None else n_exposed
    out = np.zeros((B, A), dtype=np.float32)
    lib().orc_collision(_p(boxes, c_f), _p(sc, c_f), _p(present, c_u8), _p(out, c_f), _i64(B), _i64(A), _i64(N),
                        ctypes.c_int({'iou': 0, 'discs': 1}[metric]))
    return out


def offroad(state, lenwid, verts, faces, threshold=0.5, present=None, sc=None):
    """state B x A x 4, lenwid B x A x 2, verts (Bm x) V x 2, faces (Bm x) F x 3."""
    state, lenwid, sc, verts = _f(state), _f(lenwid), _f(sc), _f(verts)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    if verts.ndim == 2:
        verts, faces = verts[None], faces[None]
    B, A = state.shape[:2]
    present = None if present is None else np.ascontiguousarray(present, dtype=np.uint8)
    out = np.zeros((B, A), dtype=np.float32)
    lib().orc_offroad(_p(state, c_f), _p(lenwid, c_f), _p(sc, c_f), _p(present, c_u8), _p(verts, c_f), _p(faces, c_i32), _i64(B), _i64(A),
                      _i64(verts.shape[1]), _i64(faces.shape[1]), _i64(verts.shape[0]), _fl(threshold), _p(out, c_f))
    return out


def actor_template(lenwid):
    lenwid = _f(lenwid)
    out = np.empty(lenwid.shape[:-1] + (7, 2), dtype=np.float32)
    lib().orc_actor_template(_p(lenwid, c_f), _p(out, c_f), _i64(lenwid.size // 2))
    return out


def fill_convex_poly(img, pts, color):
    """img: H x W x 3 float32 (OpenCV layout img[y, x]), modified in place."""
    assert img.dtype == np.float32 and img.flags.c_contiguous
    pts = np.ascontiguousarray(pts, dtype=np.int32)
    col = np.asarray(color, dtype=np.float32)
    lib().orc_fill_convex_poly(_p(img, c_f), ctypes.c_int(img.shape[1]), ctypes.c_int(img.shape[0]), _p(pts, c_i32),
                               ctypes.c_int(pts.shape[0]), _p(col, c_f))
    return img


def render_rgb_mesh(verts, attrs, faces, cam_xy, cam_sc, scale, res, record=False):
    """Generic CV2Renderer.render_rgb_mesh restatement: verts Nimg x V x 3, attrs Nimg x V x 3, faces Nimg x F x 3.
    Returns Nimg x H x W x 3 (transposed as the reference returns it) [, call lists]."""
    verts, attrs, cam_xy, cam_sc = _f(verts), _f(attrs), _f(cam_xy), _f(cam_sc)
    faces = np.ascontiguousarray(faces, dtype=np.int32)
    n, V, F = verts.shape[0], verts.shape[1], faces.shape[1]
    W = H = int(res)
    out = np.zeros((n, H, W, 3), dtype=np.float32)
    tris = np.zeros((n, F, 6), dtype=np.int32) if record else None
    cols = np.zeros((n, F, 3), dtype=np.uint8) if record else None
    counts = []
    for i in range(n):
        k = lib().orc_render_rgb_mesh_one(_p(verts[i], c_f), _p(attrs[i], c_f), _p(faces[i], c_i32), _i64(V), _i64(F),
                                          _fl(cam_xy[i, 0]), _fl(cam_xy[i, 1]), _fl(cam_sc[i, 0]), _fl(cam_sc[i, 1]), _fl(scale),
                                          ctypes.c_int(W), ctypes.c_int(H), _p(out[i], c_f),
                                          _p(tris[i], c_i32) if record else None, _p(cols[i], c_u8) if record else None, _i64(F))
        counts.append(k)
    return (out, tris, cols, counts) if record else out


def static_mesh_arrays(verts, faces, vert_category, categories, colors=None, levels=None):
    """BirdviewMesh.fill_attr (mesh.py:663-683): per-vertex z and colour in [0,1]."""
    colors = DEFAULT_COLORS if colors is None else colors
    levels = DEFAULT_LEVELS if levels is None else levels
    verts = _f(verts)
    zs = np.array([levels[c] for c in categories], dtype=np.float32)
    cols = (np.array([colors[c] for c in categories], dtype=np.float32).reshape(-1, 3) / np.float32(255.0)).astype(np.float32)
    vc = np.asarray(vert_category).astype(np.int64)
    sverts = np.concatenate([verts[:, :2], zs[vc][:, None]], -1).astype(np.float32) if len(verts) else np.zeros((0, 3), np.float32)
    sattrs = cols[vc].astype(np.float32) if len(verts) else np.zeros((0, 3), np.float32)
    return sverts, sattrs, np.ascontiguousarray(faces, dtype=np.int32)


def render_scenes(state, size, mask, cam_xy, cam_sc, sverts, sattrs, sfaces, fov, res, agent_sc=None,
                  actor_levels=None, actor_colors=None, record=False, images=True, out=None):
    """Simulator.render restatement (reference dataflow).  state B x N x 4, size B x N x 2,
    mask B x Nc x N (present & rendering mask), cam_xy / cam_sc B x Nc x 2.  Returns B x Nc x 3 x H x W."""
    state, size, cam_xy, cam_sc = _f(state), _f(size), _f(cam_xy), _f(cam_sc)
    B, N = state.shape[:2]
    Nc = cam_xy.shape[1]
    if agent_sc is None:
        agent_sc = np.stack([np.sin(state[..., 2]), np.cos(state[..., 2])], -1)
    agent_sc = _f(agent_sc)
    tmpl = actor_template(size)
    if actor_levels is None:
        actor_levels = np.broadcast_to(np.array([DEFAULT_LEVELS['vehicle'], DEFAULT_LEVELS['direction']], np.float32), (B, N, 2))
    if actor_colors is None:
        c = np.array([DEFAULT_COLORS['vehicle'], DEFAULT_COLORS['direction']], np.float32) / np.float32(255.0)
        actor_colors = np.broadcast_to(c, (B, N, 2, 3))
    actor_levels, actor_colors = _f(actor_levels), _f(actor_colors)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    sverts, sattrs = _f(sverts), _f(sattrs)
    sfaces = np.ascontiguousarray(sfaces, dtype=np.int32)
    W = H = int(res)
    if images and out is not None:          # a caller's buffer, reused from call to call (every element is written)
        assert out.shape == (B, Nc, 3, H, W) and out.dtype == np.float32 and out.flags.c_contiguous
    else:
        out = np.zeros((B, Nc, 3, H, W), dtype=np.float32) if images else None
    cap = sfaces.shape[0] + 3 * N
    tris = np.zeros((B * Nc, cap, 6), dtype=np.int32) if record else None
    cols = np.zeros((B * Nc, cap, 3), dtype=np.uint8) if record else None
    cnt = np.zeros(B * Nc, dtype=np.int64)
    lib().orc_render_scenes(_p(state, c_f), _p(agent_sc, c_f), _p(tmpl, c_f), _p(actor_levels, c_f), _p(actor_colors, c_f),
                            _p(mask, c_u8), _p(cam_xy, c_f), _p(cam_sc, c_f), _p(sverts, c_f), _p(sattrs, c_f), _p(sfaces, c_i32),
                            _i64(sverts.shape[0]), _i64(sfaces.shape[0]), _i64(B), _i64(Nc), _i64(N), _fl(2.0 / fov),
                            ctypes.c_int(W), ctypes.c_int(H), _p(out, c_f), _p(tris, c_i32), _p(cols, c_u8), _i64(cap),
                            _p(cnt, c_i64))
    return (out, tris, cols, cnt) if record else out


def occlusion_mask(state, size, present, n_exposed):
    """StandardSensingObservationNoise.get_noisy_present_mask: state (B,E,4), size (B,E,2), present (B,E) -> (B,A,E) bool"""
    state, size = _f(state), _f(size)
    present = np.ascontiguousarray(present, dtype=np.uint8)
    B, E = present.shape
    out = np.zeros((B, int(n_exposed), E), dtype=np.uint8)
    lib().orc_occlusion_mask(_p(state, c_f), _p(size, c_f), _p(present, c_u8), _p(out, c_u8), _i64(B), _i64(n_exposed), _i64(E))
    return out.astype(bool)


def set_trim_mesh(on: bool):
    """CV2RendererConfig.trim_mesh_before_rendering (cv2.py:15) for every render call that follows"""
    lib().orc_set_trim_mesh(ctypes.c_int(1 if on else 0))


def set_num_threads(n):
    lib().orc_set_num_threads(ctypes.c_int(int(n)))


def num_threads():
    return int(lib().orc_num_threads())
