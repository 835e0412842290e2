"""CPU check of the DEFINITION of the K3 backward (csrc/raster_bwd.hip, restated in numpy in test_gpu_raster_backward.numpy_backward):
the edge-sampling gradient evaluated on the oracle's OpenCV-style image follows central differences of the oracle's own renders at
4x the resolution.  (The HIP kernel is compared with the same numpy restatement in the GPU tests.)"""
import numpy as np

from test_gpu_raster_backward import numpy_backward


def gauss(res, seed=3):
    yy, xx = np.meshgrid((np.arange(res) + 0.5) / res, (np.arange(res) + 0.5) / res)
    w, g = np.zeros((3, res, res)), np.random.default_rng(seed)
    for ch in range(3):
        for _ in range(3):
            cx, cy, s, a = g.uniform(0.2, 0.8), g.uniform(0.2, 0.8), g.uniform(1 / 6, 1 / 3), g.uniform(-1, 1)
            w[ch] += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return (w / 255.0).astype(np.float32)


def test_edge_sampling_gradient_follows_finite_differences_of_the_oracle(oracle):
    fov, res, ss = 35.0, 128, 4
    size = np.array([[[12.0, 3.0]]], np.float32)
    mask = np.ones((1, 1, 1), bool)
    tmpl = oracle.actor_template(size)
    empty = (np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))

    def image(r, s, c, p):
        csc = np.concatenate([np.sin(p), np.cos(p)], -1).astype(np.float32)
        asc = np.stack([np.sin(s[..., 2]), np.cos(s[..., 2])], -1).astype(np.float32)
        return oracle.render_scenes(s, size, mask, c, csc, *empty, fov, r, agent_sc=asc).reshape(3, r, r), csc, asc

    f, f2 = gauss(res), gauss(res * ss) / (ss * ss)
    for psi_a, psi_c, pos in ((0.3, 0.9, (1.0, 2.0)), (0.7, 2.0, (1.0, 2.0))):
        state = np.array([[[pos[0], pos[1], psi_a, 0.0]]], np.float32)
        cam_xy, cam_psi = np.zeros((1, 1, 2), np.float32), np.full((1, 1, 1), psi_c, np.float32)
        img, csc, asc = image(res, state, cam_xy, cam_psi)
        ga, gc, _ = numpy_backward(state, asc, tmpl, mask, cam_xy, csc, img.reshape(1, 1, 3, res, res), f[None, None], fov, res)
        (sj, cj), (sc_, cc_) = asc[0, 0], csc[0, 0]
        g = np.array([ga[0, 0, 0, 0], ga[0, 0, 0, 1], ga[0, 0, 0, 2] * cj - ga[0, 0, 0, 3] * sj, gc[0, 0, 2] * cc_ - gc[0, 0, 3] * sc_])
        fd = []
        for which, h in (('x', 0.3), ('y', 0.3), ('psi', 0.05), ('cpsi', 0.05)):
            vals = []
            for sgn in (1, -1):
                s2, p2 = state.copy(), cam_psi.copy()
                if which == 'cpsi':
                    p2[0, 0, 0] += sgn * h
                else:
                    s2[0, 0, 'xy'.index(which) if which in 'xy' else 2] += sgn * h
                vals.append((image(res * ss, s2, cam_xy, p2)[0].astype(np.float64) * f2).sum())
            fd.append((vals[0] - vals[1]) / (2 * h))
        fd = np.array(fd)
        # translations set the scale; a heading acts with a lever arm of a few metres (half the actor / the distance to the camera)
        st = max(np.abs(fd[:2]).max(), np.abs(g[:2]).max(), 1.0)
        assert np.abs(g[:2] - fd[:2]).max() <= 0.3 * st, (psi_a, psi_c, g, fd)
        assert abs(g[2] - fd[2]) <= 0.3 * max(abs(g[2]), abs(fd[2]), 6.0 * st), (psi_a, psi_c, g, fd)
        assert abs(g[3] - fd[3]) <= 0.3 * max(abs(g[3]), abs(fd[3]), 8.0 * st), (psi_a, psi_c, g, fd)
