"""K3 backward (build-defined edge-sampling gradient, DESIGN.md "K3 backward"): the HIP kernel against a numpy restatement of the
same definition, against finite differences of the rendered image under a smooth weighting, and through the Simulator."""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, actor_keys, dev, make_map

pytestmark = pytest.mark.gpu

SIDE_IN, SIDE_OUT = 0.25, 1.25
MAP_VC, MAP_CATS = np.array([0] * 4 + [1] * 4 + [2] * 4, np.int64), ['road', 'left_lane', 'right_lane']


@pytest.fixture(scope='module')
def ops():
    from torchdrivesim_amd import _ops
    return _ops


def numpy_backward(state, sc, tmpl, mask, cam_xy, cam_sc, image, gout, fov, res):
    """the definition in csrc/raster_bwd.hip, one sample loop per edge, float64"""
    B, Nc = cam_xy.shape[:2]
    N = state.shape[1]
    k, half = (2.0 / fov) * res * 0.5, res * 0.5
    g_agent, g_cam, g_tmpl = np.zeros((B, Nc, N, 4)), np.zeros((B, Nc, 4)), np.zeros((B, Nc, N, 7, 2))
    view_r = 1.05 * 1.41421356 / (2.0 / fov)
    for b in range(B):
        for c in range(Nc):
            cs, cc = cam_sc[b, c]
            cx, cy = cam_xy[b, c]
            I, G = image[b, c].astype(np.float64), gout[b, c].astype(np.float64)
            for j in range(N):
                if not mask[b, c, j]:
                    continue
                x, y = state[b, j, :2]
                sj, cj = sc[b, j]
                t = tmpl[b, j].astype(np.float64)
                if (x - cx) ** 2 + (y - cy) ** 2 > (view_r + np.hypot(*t[0]) + fov) ** 2:
                    continue
                for e in range(7):
                    ia, ib = (e, (e + 1) % 4) if e < 4 else (e, 4 if e == 6 else e + 1)
                    ctr = t[:4].mean(0) if e < 4 else t[4:7].mean(0)
                    rel = lambda q: np.array([cj * q[0] - sj * q[1] + x - cx, sj * q[0] + cj * q[1] + y - cy])
                    pix = lambda v: np.array([-k * (cc * v[0] + cs * v[1]) + half, -k * (-cs * v[0] + cc * v[1]) + half])
                    va, vb = rel(t[ia]), rel(t[ib])
                    pa, pb, pc = np.floor(pix(va)) + 0.5, np.floor(pix(vb)) + 0.5, pix(rel(ctr))      # the forward draws truncated vertices
                    d = pb - pa
                    ln = np.hypot(*d)
                    if ln <= 1e-6:
                        continue
                    n = np.array([d[1], -d[0]]) / ln
                    if n @ (0.5 * (pa + pb) - pc) < 0:
                        n = -n
                    ns = max(1, int(np.ceil(np.float32(ln))))
                    dl = ln / ns
                    A0 = A1 = 0.0
                    for si in range(ns):
                        u = (si + 0.5) / ns
                        p = pa + u * d
                        pi_, po = np.floor(p - SIDE_IN * n), np.floor(p + SIDE_OUT * n)
                        if (pi_ < 0).any() or (po < 0).any() or (pi_ >= res).any() or (po >= res).any():
                            continue
                        xi, yi, xo, yo = int(pi_[0]), int(pi_[1]), int(po[0]), int(po[1])
                        D = (0.5 * (G[:, xi, yi] + G[:, xo, yo]) * (I[:, xi, yi] - I[:, xo, yo])).sum()
                        A0 += D * dl * (1 - u)
                        A1 += D * dl * u
                    nM = lambda qx, qy: -k * (n[0] * (cc * qx + cs * qy) + n[1] * (-cs * qx + cc * qy))
                    T = A0 * t[ia] + A1 * t[ib]
                    g = np.array([(A0 + A1) * nM(1, 0), (A0 + A1) * nM(0, 1), nM(-T[1], T[0]), nM(T[0], T[1])])
                    g_agent[b, c, j] += g
                    q = np.array([nM(cj, sj), nM(-sj, cj)])            # a sample at u moves with (1 - u) dt_a + u dt_b, rotated into the world
                    g_tmpl[b, c, j, ia] += A0 * q
                    g_tmpl[b, c, j, ib] += A1 * q
            # camera: every colour boundary of the image moves rigidly (neighbouring pixel pairs)
            ii, jj = np.meshgrid(np.arange(res, dtype=np.float64), np.arange(res, dtype=np.float64), indexing='ij')
            Dx = (0.5 * (G[:, :-1] + G[:, 1:]) * (I[:, 1:] - I[:, :-1])).sum(0)           # pairs (i, j) - (i + 1, j)
            Dy = (0.5 * (G[:, :, :-1] + G[:, :, 1:]) * (I[:, :, 1:] - I[:, :, :-1])).sum(0)
            dxx, dyx = ii[1:] - half, jj[1:] + 0.5 - half
            dxy, dyy = ii[:, 1:] + 0.5 - half, jj[:, 1:] - half
            Sx, Sy = Dx.sum(), Dy.sum()
            Cc = (Dx * (cc * dxx - cs * dyx)).sum() + (Dy * (cs * dxy + cc * dyy)).sum()
            Cs = (Dx * (cs * dxx + cc * dyx)).sum() - (Dy * (cc * dxy - cs * dyy)).sum()
            g_cam[b, c] = [-k * (cc * Sx - cs * Sy), -k * (cs * Sx + cc * Sy), -Cs, -Cc]
    return g_agent, g_cam, g_tmpl


def scene(gen, B=2, N=5, Nc=3, big=False):
    # a ground quad with two lane stripes on it: the static map has colour boundaries of its own
    verts = np.array([[-60, -60], [60, -60], [60, 60], [-60, 60], [-60, -7], [60, -5], [60, -1], [-60, -3], [-9, -60], [-2, -60], [4, 60], [-3, 60]], np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7], [8, 9, 10], [8, 10, 11]], np.int32)
    state = np.concatenate([gen.uniform(-8, 8, (B, N, 2)), gen.uniform(-np.pi, np.pi, (B, N, 1)), gen.uniform(0, 5, (B, N, 1))], -1).astype(np.float32)
    size = (np.array([12.0, 3.0]) if big else np.array([4.5, 2.0])) * gen.uniform(0.9, 1.1, (B, N, 2))
    cam_xy = gen.uniform(-3, 3, (B, Nc, 2)).astype(np.float32)
    cam_psi = gen.uniform(-np.pi, np.pi, (B, Nc, 1)).astype(np.float32)
    mask = gen.uniform(size=(B, Nc, N)) < 0.9
    return verts, faces, state, size.astype(np.float32), cam_xy, cam_psi, mask


def smooth_weight(res, gen, n=6):
    """a smooth function over the image, 3 channels"""
    yy, xx = np.meshgrid(np.arange(res), np.arange(res))
    w = np.zeros((3, res, res))
    for ch in range(3):
        for _ in range(n):
            cx, cy, s, a = gen.uniform(0, res), gen.uniform(0, res), gen.uniform(res / 6, res / 2), gen.uniform(-1, 1)
            w[ch] += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return w.astype(np.float32) / 255.0


def render(ops, smap, oracle, state_t, size, mask, cam_xy_t, cam_sc_t, fov, res, diff):
    B, N = state_t.shape[:2]
    tmpl = dev(oracle.actor_template(size))
    sc = ops.heading_sc(state_t[..., 2])
    f = ops.raster_scene_diff if diff else ops.raster_scene
    return f(smap, state_t, sc, tmpl, actor_keys(smap, B, N), dev(mask), cam_xy_t, cam_sc_t, fov, res)


@pytest.mark.parametrize('res', [96, 64, 90, 256, 320])      # float4 rows: several bands per wave / one group per lane / scalar path (90) / groups
def test_kernel_matches_numpy_definition(ops, oracle, res):   # of a row in two waves (320)
    gen = np.random.default_rng(5)
    verts, faces, state, size, cam_xy, cam_psi, mask = scene(gen)
    smap = make_map(ops, verts, faces, MAP_VC, MAP_CATS)
    fov = 35.0
    st = dev(state).requires_grad_(True)
    cxy = dev(cam_xy).requires_grad_(True)
    cpsi = dev(cam_psi)
    csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1).requires_grad_(True)
    sc = ops.heading_sc(st[..., 2]).detach().requires_grad_(True)
    tmpl = dev(oracle.actor_template(size)).requires_grad_(True)
    B, N = state.shape[:2]
    img = ops.raster_scene_diff(smap, st, sc, tmpl, actor_keys(smap, B, N), dev(mask), cxy, csc, fov, res)
    gout = torch.from_numpy(smooth_weight(res, gen)).to(DEV).expand_as(img).contiguous() * torch.linspace(0.5, 1.5, img.shape[0] * img.shape[1],
                                                                                                        device=DEV).view(img.shape[0], img.shape[1], 1, 1, 1)
    img.backward(gout)
    ga, gc, gt = numpy_backward(state, sc.detach().cpu().numpy(), tmpl.detach().cpu().numpy(), mask, cam_xy, csc.detach().cpu().numpy(), img.detach().cpu().numpy(),
                            gout.cpu().numpy(), fov, res)
    ga = ga.sum(1)
    scale = np.abs(ga).max() + 1e-9
    assert scale > 1e-3, 'degenerate test scene'
    np.testing.assert_allclose(st.grad[..., :2].cpu().numpy(), ga[..., :2], atol=2e-4 * scale, rtol=2e-3)
    assert (st.grad[..., 2:] == 0).all()
    np.testing.assert_allclose(sc.grad.cpu().numpy(), ga[..., 2:], atol=2e-4 * max(np.abs(ga[..., 2:]).max(), 1e-9), rtol=2e-3)
    np.testing.assert_allclose(cxy.grad.cpu().numpy(), gc[..., :2], atol=2e-4 * max(np.abs(gc[..., :2]).max(), 1e-9), rtol=2e-3)
    np.testing.assert_allclose(csc.grad.cpu().numpy(), gc[..., 2:], atol=2e-4 * max(np.abs(gc[..., 2:]).max(), 1e-9), rtol=2e-3)
    gt = gt.sum(1)                                                # template vertices (the actors' sizes), over cameras
    assert np.abs(gt).max() > 1e-3
    np.testing.assert_allclose(tmpl.grad.cpu().numpy(), gt, atol=2e-4 * np.abs(gt).max(), rtol=2e-3)


@pytest.mark.parametrize('res', [64, 256, 288, 512])      # one word column .. two strips (288) .. more than 64 row quads per column (512)
def test_index_slice_backward_equals_the_image_backward(ops, oracle, res):
    """The two backward kernels compute the same sums: tds_raster_scene_bwd_idx_f32 from the forward's key-index slices (reading the
    incoming gradient next to colour boundaries only) and tds_raster_scene_bwd_f32 from the forward image."""
    gen = np.random.default_rng(17)
    verts, faces, state, size, cam_xy, cam_psi, mask = scene(gen, B=2, N=6, Nc=4)
    smap = make_map(ops, verts, faces, MAP_VC, MAP_CATS)
    tmpl = dev(oracle.actor_template(size))
    B, N = state.shape[:2]
    cpsi = dev(cam_psi)
    gout = None
    grads = []
    for use in (True, False):
        ops.use_index_slices = use
        try:
            st, cxy = dev(state).requires_grad_(True), dev(cam_xy).requires_grad_(True)
            csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1).requires_grad_(True)
            sc = ops.heading_sc(st[..., 2]).detach().requires_grad_(True)
            tm = tmpl.clone().requires_grad_(True)
            img = ops.raster_scene_diff(smap, st, sc, tm, actor_keys(smap, B, N), dev(mask), cxy, csc, 35.0, res)
            if gout is None:
                gout = torch.randn(img.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
            img.backward(gout)
            grads.append([t.grad.double().cpu().numpy() for t in (st, sc, cxy, csc, tm)])
        finally:
            ops.use_index_slices = True
    for a, b in zip(*grads):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-4 * max(np.abs(b).max(), 1e-9))
    assert np.abs(grads[0][2]).max() > 0 and np.abs(grads[0][0]).max() > 0 and np.abs(grads[0][4]).max() > 0
    # a gradient that is ONE image broadcast over all cameras (stride 0) is read as such: same result as its dense copy
    one = torch.randn(img.shape[2:], device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    res2 = []
    for g in (one.expand_as(img), one.expand_as(img).contiguous()):
        st, cxy = dev(state).requires_grad_(True), dev(cam_xy).requires_grad_(True)
        csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1)
        out = ops.raster_scene_diff(smap, st, ops.heading_sc(st[..., 2]), tmpl, actor_keys(smap, B, N), dev(mask), cxy, csc, 35.0, res)
        out.backward(g)
        res2.append((st.grad.clone(), cxy.grad.clone()))
    assert torch.equal(res2[0][0], res2[1][0]) and torch.equal(res2[0][1], res2[1][1])


def test_colour_gradient_is_the_per_key_sum_of_the_incoming_gradient(ops, oracle):
    """R6: dL/dcolour of a key = sum of the incoming gradient over the pixels where that key wins, exact (integer-valued gradients make every
    float32 sum exact whatever its order)"""
    gen = np.random.default_rng(23)
    verts, faces, state, size, cam_xy, cam_psi, mask = scene(gen, B=2, N=6, Nc=3)
    smap = make_map(ops, verts, faces, MAP_VC, MAP_CATS)
    tmpl = dev(oracle.actor_template(size))
    B, N, res = state.shape[0], state.shape[1], 128
    akeys = actor_keys(smap, B, N)
    cpsi = dev(cam_psi)
    csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1)
    st = dev(state)
    sc = ops.heading_sc(st[..., 2])
    img, slices, keys = ops.raster_scene(smap, st, sc, tmpl, akeys, dev(mask), dev(cam_xy), csc, 35.0, res, index_slices=True)
    assert slices is not None and keys == sorted(keys) and 3 <= len(keys) <= 15
    # the slices decode to the image: pixel = colour of the key whose 1-based position they hold
    sl = slices.view(B * 3, (res + 31) // 32, res // 4, 4, 4).cpu().numpy().astype(np.uint32)
    idx = np.zeros((B * 3, res, res), np.int64)
    xs = np.arange(res)
    for bit in range(4 if len(keys) > 7 else (3 if len(keys) > 3 else 2)):
        words = sl[:, :, :, bit, :].reshape(B * 3, (res + 31) // 32, res)                   # [img][x word][y]
        idx |= (((words[:, xs // 32, :] >> (xs % 32)[None, :, None].astype(np.uint32)) & 1).astype(np.int64)) << bit      # [img][x][y]
    table = np.array([[0, 0, 0]] + [[(k >> 16) & 255, (k >> 8) & 255, k & 255] for k in keys], np.float32)
    np.testing.assert_array_equal(table[idx].transpose(0, 3, 1, 2).reshape(img.shape), img.cpu().numpy())
    # gradient with respect to the colours of three of the keys (and of one key that is not in the scene)
    color_keys = [keys[0], keys[-1], keys[1], 0x05123456]
    kc = torch.tensor([[(k >> 16) & 255, (k >> 8) & 255, k & 255] for k in color_keys], dtype=torch.float32, device=DEV, requires_grad=True)
    st2 = dev(state).requires_grad_(True)
    out = ops.raster_scene_diff(smap, st2, ops.heading_sc(st2[..., 2]), tmpl, akeys, dev(mask), dev(cam_xy), csc, 35.0, res, key_colors=kc, color_keys=color_keys)
    gout = torch.randint(-3, 4, out.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).float()
    out.backward(gout)
    g = gout.cpu().numpy().reshape(B * 3, 3, res, res)
    expect = np.stack([np.stack([g[:, ch][idx == (keys.index(k) + 1 if k in keys else -1)].sum() for ch in range(3)]) for k in color_keys])
    np.testing.assert_array_equal(kc.grad.cpu().numpy(), expect.astype(np.float32))
    assert np.abs(expect[:3]).sum() > 0 and not expect[3].any()
    with pytest.raises(RuntimeError, match='colour gradients need'):
        ops.use_index_slices = False
        try:
            ops.raster_scene_diff(smap, st2, ops.heading_sc(st2[..., 2]), tmpl, akeys, dev(mask), dev(cam_xy), csc, 35.0, res, key_colors=kc, color_keys=color_keys)
        finally:
            ops.use_index_slices = True


def fine_weight(res, seed, n=3):
    """a smooth function of the normalised image coordinates, sampled at `res` (so that it can be evaluated at any resolution)"""
    gen = np.random.default_rng(seed)
    yy, xx = np.meshgrid((np.arange(res) + 0.5) / res, (np.arange(res) + 0.5) / res)
    w = np.zeros((3, res, res))
    for ch in range(3):
        for _ in range(n):
            cx, cy, s, a = gen.uniform(0.2, 0.8), gen.uniform(0.2, 0.8), gen.uniform(1 / 6, 1 / 3), gen.uniform(-1, 1)
            w[ch] += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return torch.from_numpy((w / 255.0).astype(np.float32)).to(DEV)


@pytest.mark.parametrize('param', ['x', 'y', 'psi', 'cam_x', 'cam_y', 'cam_psi', 'size'])
def test_gradient_follows_finite_differences(ops, oracle, param):
    """L = sum f I with a smooth f.  The image is piecewise constant, so the reference is a central difference of the hard rasterisation
    rendered at 4x the resolution (same f, weights / 16) over a step of a few pixels.  A single pose compares a once-per-pixel boundary
    sample against an OpenCV-style raster (2 - 15 % apart, the phase of the edges against the pixel grid); both sides are therefore
    AVERAGED over 12 sub-pixel shifts of the whole scene -- the gradient of the shift-averaged loss -- and must agree within 10 % of the scale.
    'size': the agents' length and width (through the template vertices), judged together against their common scale -- the two are sums of
    the same edge integrals (a length moves the short edges, a width the long ones), and the absolute deviation is the same for both
    (about 2 units here, 4 % of the largest size gradient; on its own the small length gradient of this scene would be 12 % off)."""
    gen = np.random.default_rng(11)
    verts, faces, state, size, cam_xy, cam_psi, mask = scene(gen, B=1, N=3, Nc=1, big=True)
    mask[:] = True
    state[0, :, :2] = [[1, 2], [8, -5], [-6, 7]]                  # all actors entirely in view (12 m long, view +-17.5 m)
    state[0, :, 2] = [0.3, 0.7, 2.1]
    cam_xy[:] = 0.25
    cam_psi[:] = 0.9
    smap = make_map(ops, verts, faces, MAP_VC, MAP_CATS)
    fov, res, ss = 35.0, 256, 4

    from torchdrivesim_amd.mesh import actor_template

    def loss_of(r, state_np, cam_xy_np, cam_psi_np, diff=False, size_np=None):
        f = fine_weight(r, 3) / ((r // res) ** 2)
        st = dev(state_np).requires_grad_(diff)
        cxy = dev(cam_xy_np).requires_grad_(diff)
        cpsi = dev(cam_psi_np).requires_grad_(diff)
        csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1)
        sz = dev(size if size_np is None else size_np).requires_grad_(diff)
        # sizes reach the image through the template vertices (the product's actor_template; bit-equal to the reference's, G4)
        fn = ops.raster_scene_diff if diff else ops.raster_scene
        img = fn(smap, st, ops.heading_sc(st[..., 2]), actor_template(sz).contiguous(), actor_keys(smap, 1, st.shape[1]), dev(mask), cxy, csc, fov, r)
        return (img.double() * f.double()).sum(), st, cxy, cpsi, sz

    cols = ('length', 'width') if param == 'size' else (param,)
    h = 0.05 if 'psi' in param else (0.6 if param == 'size' else 0.3)          # size: an outline edge moves by 0.3 m either way (half the change)
    n = 1 if param.startswith('cam') else state.shape[1]
    shifts = np.random.default_rng(2).uniform(-0.5, 0.5, (12, 2)) * (fov / res)          # within one pixel
    g_sum, fd_sum = np.zeros((len(cols), n)), np.zeros((len(cols), n))
    for sh in shifts:
        # actors and camera parameters under test move against the static map and the image grid by a fraction of a pixel
        st0, cx0 = state.copy(), cam_xy.copy()
        st0[0, :, :2] += sh
        if param.startswith('cam'):
            cx0[0, :, :] += sh[::-1] * 0.5
        L, st, cxy, cpsi, sz = loss_of(res, st0, cx0, cam_psi, True)
        L.backward()
        grads = dict(x=st.grad[0, :, 0], y=st.grad[0, :, 1], psi=st.grad[0, :, 2], cam_x=cxy.grad[0, :, 0], cam_y=cxy.grad[0, :, 1],
                     cam_psi=cpsi.grad[0, :, 0], length=sz.grad[0, :, 0], width=sz.grad[0, :, 1])
        for ci, name in enumerate(cols):
            g_sum[ci] += grads[name].cpu().numpy().astype(np.float64)
            for i in range(n):
                vals = []
                for sgn in (+1, -1):
                    s2, c2, p2, z2 = st0.copy(), cx0.copy(), cam_psi.copy(), size.copy()
                    tgt, col = dict(x=(s2, 0), y=(s2, 1), psi=(s2, 2), cam_x=(c2, 0), cam_y=(c2, 1), cam_psi=(p2, 0), length=(z2, 0), width=(z2, 1))[name]
                    tgt[0, i, col] += sgn * h
                    vals.append(loss_of(res * ss, s2, c2, p2, size_np=z2)[0].item())
                fd_sum[ci, i] += (vals[0] - vals[1]) / (2 * h)
    g, fd = g_sum / len(shifts), fd_sum / len(shifts)
    # (a camera move shifts the static map as well: its lane stripes contribute to the camera gradient)
    scale = max(np.abs(fd).max(), np.abs(g).max())
    assert scale > 1.0, 'degenerate test'
    assert np.abs(g - fd).max() <= 0.10 * scale, (param, g, fd)


def test_gradients_reach_simulator_state(ops):
    """render_egocentric is differentiable when the state requires grad (and only then)"""
    from test_gpu_simulator import make_sim, town_mesh
    from torchdrivesim_amd.utils import Resolution
    gen = np.random.default_rng(3)
    B, A = 2, 6
    road, t = town_mesh(B)
    rv = t['verts'][t['vert_category'] == [str(c) for c in t['categories']].index('road')]
    anchor = rv[gen.integers(0, len(rv), (B, 1))]
    state = np.concatenate([anchor + gen.uniform(-10, 10, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 5, (B, A, 1))], -1).astype(np.float32)
    size = np.tile(np.array([4.5, 2.0], np.float32), (B, A, 1))
    sim = make_sim(state, size, np.ones((B, A), bool), road)
    with torch.no_grad():
        assert not sim.render_egocentric(res=Resolution(64, 64), fov=35.0).requires_grad
    st = sim.get_state().detach().clone().requires_grad_(True)
    sim.kinematic_model.set_state(st)
    img = sim.render_egocentric(res=Resolution(128, 128), fov=35.0)
    assert img.requires_grad and img.shape == (B, A, 3, 128, 128)
    w = torch.linspace(0, 1, 128, device=img.device).view(1, 1, 1, 128, 1)
    (img * w).sum().backward()
    g = st.grad
    assert g is not None and torch.isfinite(g).all()
    assert (g[..., 3] == 0).all()                     # speed never enters the image
    assert g[..., :3].abs().sum() > 0


def test_gradients_reach_the_agent_sizes(ops):
    """render_egocentric differentiates the agents' length and width (through the template vertices) when agent_size requires grad"""
    from test_gpu_simulator import make_sim, town_mesh
    from torchdrivesim_amd.utils import Resolution
    gen = np.random.default_rng(4)
    B, A = 2, 5
    road, t = town_mesh(B)
    rv = t['verts'][t['vert_category'] == [str(c) for c in t['categories']].index('road')]
    anchor = rv[gen.integers(0, len(rv), (B, 1))]
    state = np.concatenate([anchor + gen.uniform(-8, 8, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), gen.uniform(0, 5, (B, A, 1))], -1).astype(np.float32)
    size = np.tile(np.array([4.5, 2.0], np.float32), (B, A, 1))
    sim = make_sim(state, size, np.ones((B, A), bool), road)
    plain = sim.render_egocentric(res=Resolution(128, 128), fov=35.0)
    assert not plain.requires_grad
    sz = sim.get_agent_size().detach().clone().requires_grad_(True)
    sim.set_agent_size(sz) if hasattr(sim, 'set_agent_size') else setattr(sim, 'agent_size', sz)
    img = sim.render_egocentric(res=Resolution(128, 128), fov=35.0)
    assert img.requires_grad and torch.equal(img.detach(), plain)
    img.sum().backward()                                          # a larger vehicle covers more (darker or brighter) pixels: a gradient wherever it is visible
    assert sz.grad is not None and torch.isfinite(sz.grad).all() and sz.grad.abs().sum() > 0

