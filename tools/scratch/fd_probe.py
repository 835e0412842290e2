import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_parity import DEV, actor_keys, dev, make_map
from test_gpu_raster_backward import scene, render
from torchdrivesim_amd import _ops as ops
from oracle import oracle

def weight(res, seed, n=4):
    gen = np.random.default_rng(seed)
    yy, xx = np.meshgrid((np.arange(res) + 0.5) / res, (np.arange(res) + 0.5) / res)
    w = np.zeros((3, res, res))
    for ch in range(3):
        for _ in range(n):
            cx, cy, s, a = gen.uniform(0, 1), gen.uniform(0, 1), gen.uniform(1 / 8, 1 / 3), gen.uniform(-1, 1)
            w[ch] += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
    return torch.from_numpy((w / 255.0).astype(np.float32)).to(DEV)

gen = np.random.default_rng(11)
verts, faces, state, size, cam_xy, cam_psi, mask = scene(gen, B=1, N=3, Nc=1, big=True)
mask[:] = True
state[0, :, :2] = [[0, 0], [9, 6], [-7, -9]]
cam_xy[:] = 0.5
smap = make_map(ops, verts, faces, np.zeros(4, np.int64), ['road'])
fov = 35.0
def loss_of(res, s, c, p, diff=False, ss=1):
    f = weight(res, 7) / (ss * ss)
    st = dev(s).requires_grad_(diff); cxy = dev(c).requires_grad_(diff); cpsi = dev(p).requires_grad_(diff)
    csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1)
    img = render(ops, smap, oracle, st, size, mask, cxy, csc, fov, res, diff)
    return (img * f).sum(), st, cxy, cpsi
L, st, cxy, cpsi = loss_of(256, state, cam_xy, cam_psi, True)
L.backward()
grads = dict(x=st.grad[0, :, 0], y=st.grad[0, :, 1], psi=st.grad[0, :, 2], cam_x=cxy.grad[0, :, 0], cam_y=cxy.grad[0, :, 1], cam_psi=cpsi.grad[0, :, 0])
for ss in (1, 4):
  for param in grads:
    for h in ((0.15, 0.3, 0.6) if 'psi' not in param else (0.01, 0.02, 0.04)):
        n = 1 if param.startswith('cam') else 3
        fd = np.zeros(n)
        for i in range(n):
            vals = []
            for sgn in (1, -1):
                s2, c2, p2 = state.copy(), cam_xy.copy(), cam_psi.copy()
                if param == 'x': s2[0, i, 0] += sgn * h
                if param == 'y': s2[0, i, 1] += sgn * h
                if param == 'psi': s2[0, i, 2] += sgn * h
                if param == 'cam_x': c2[0, i, 0] += sgn * h
                if param == 'cam_y': c2[0, i, 1] += sgn * h
                if param == 'cam_psi': p2[0, i, 0] += sgn * h
                vals.append(loss_of(256 * ss, s2, c2, p2, ss=ss)[0].item())
            fd[i] = (vals[0] - vals[1]) / (2 * h)
        print(ss, param, h, 'grad', np.round(grads[param].cpu().numpy(), 2), 'fd', np.round(fd, 2))
