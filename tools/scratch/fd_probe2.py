import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_parity import DEV, actor_keys, dev, make_map
from test_gpu_raster_backward import render
from torchdrivesim_amd import _ops as ops
from oracle import oracle
verts = np.array([[-60, -60], [60, -60], [60, 60], [-60, 60]], np.float32); faces = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
smap = make_map(ops, verts, faces, np.zeros(4, np.int64), ['road'])
fov = 35.0
size = np.array([[[12.0, 3.0]]], np.float32)
mask = np.ones((1, 1, 1), bool)
def loss_of(res, s, c, p, f, diff=False):
    st = dev(s).requires_grad_(diff); cxy = dev(c).requires_grad_(diff); cpsi = dev(p).requires_grad_(diff)
    csc = torch.cat([torch.sin(cpsi), torch.cos(cpsi)], -1)
    img = render(ops, smap, oracle, st, size, mask, cxy, csc, fov, res, diff)
    return (img * f).sum(), st, cxy, cpsi, img
for psi_a, psi_c in ((0.0, 0.0), (0.7, 0.0), (0.0, 1.1), (0.7, 2.0)):
    state = np.array([[[1.0, 2.0, psi_a, 0.0]]], np.float32); cam_xy = np.zeros((1, 1, 2), np.float32); cam_psi = np.full((1, 1, 1), psi_c, np.float32)
    for name, fcoarse in (('ramp_i', lambda res: torch.arange(res, device=DEV).view(1, res, 1).expand(3, res, res) / res / 255.0),
                          ('ramp_j', lambda res: torch.arange(res, device=DEV).view(1, 1, res).expand(3, res, res) / res / 255.0)):
        res = 256
        L, st, cxy, cpsi, img = loss_of(res, state, cam_xy, cam_psi, fcoarse(res), True)
        L.backward()
        g = [st.grad[0, 0, 0].item(), st.grad[0, 0, 1].item(), st.grad[0, 0, 2].item(), cxy.grad[0, 0, 0].item(), cxy.grad[0, 0, 1].item(), cpsi.grad[0, 0, 0].item()]
        fd = []
        ss = 4
        for which, h in (('x', 0.3), ('y', 0.3), ('psi', 0.02), ('cx', 0.3), ('cy', 0.3), ('cpsi', 0.02)):
            vals = []
            for sgn in (1, -1):
                s2, c2, p2 = state.copy(), cam_xy.copy(), cam_psi.copy()
                if which == 'x': s2[0, 0, 0] += sgn * h
                if which == 'y': s2[0, 0, 1] += sgn * h
                if which == 'psi': s2[0, 0, 2] += sgn * h
                if which == 'cx': c2[0, 0, 0] += sgn * h
                if which == 'cy': c2[0, 0, 1] += sgn * h
                if which == 'cpsi': p2[0, 0, 0] += sgn * h
                vals.append(loss_of(res * ss, s2, c2, p2, fcoarse(res * ss) / (ss * ss))[0].item())
            fd.append((vals[0] - vals[1]) / (2 * h))
        print(psi_a, psi_c, name, 'grad', np.round(g, 2), 'fd', np.round(fd, 2))
