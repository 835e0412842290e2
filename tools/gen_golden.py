#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the build container, where the reference
checkout is mounted at /root/reference; nothing from the reference is copied -- the
outputs are numeric arrays (inputs + the reference's own outputs) written to
tests/golden/*.npz.  The GPU box never runs this script.

Three import-only stand-ins are injected so that the reference's hot path imports here:
  * omegaconf         (rendering/__init__.py:7 -- type conversion only)
  * shapely.geometry  (infractions.py:7 -- only used by the `nograd` metric)
  * cv2               a *recording* fake: fillConvexPoly() appends (points, color) to a log and
                      returns the image untouched.  This pins everything the reference does up
                      to the OpenCV boundary (trim, z-order, projection, truncation, colour
                      quantisation) -- SURVEY.md section 8c, fixture G5.

Usage:  python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import json
import math
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get('TDS_REFERENCE', '/root/reference')


# --------------------------------------------------------------------------------------
# stand-ins
# --------------------------------------------------------------------------------------
def install_stubs():
    om = types.ModuleType('omegaconf')

    class DictConfig:  # never instantiated
        pass

    class OmegaConf:
        @staticmethod
        def to_container(*a, **k):
            raise NotImplementedError

    class SCMode:
        INSTANTIATE = 0

    om.DictConfig, om.OmegaConf, om.SCMode = DictConfig, OmegaConf, SCMode
    sys.modules['omegaconf'] = om

    sh = types.ModuleType('shapely')
    shg = types.ModuleType('shapely.geometry')

    class Polygon:
        def __init__(self, *a, **k):
            raise NotImplementedError('shapely stand-in')

    shg.Polygon = Polygon
    sh.geometry = shg
    sys.modules['shapely'] = sh
    sys.modules['shapely.geometry'] = shg

    cv2 = types.ModuleType('cv2')
    cv2.LINE_AA = 16
    cv2.calls = []

    def fillConvexPoly(img, points, color, shift=0, lineType=8):
        assert shift == 0 and lineType == 16
        assert img.dtype == np.float32 and img.ndim == 3 and img.shape[2] == 3
        assert points.dtype == np.int32 and points.shape == (3, 2)
        cv2.calls.append((points.copy(), list(color)))
        return img

    cv2.fillConvexPoly = fillConvexPoly
    sys.modules['cv2'] = cv2
    return cv2


def seeded(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


def npy(t):
    return t.detach().cpu().numpy()


# --------------------------------------------------------------------------------------
# G1 kinematics
# --------------------------------------------------------------------------------------
def gen_kinematic(out):
    from torchdrivesim.kinematic import (KinematicBicycle, BicycleNoReversing, SimpleKinematicModel,
                                         OrientedKinematicModel)
    g = seeded(101)
    B, A = 3, 7
    state = torch.stack([
        (torch.rand(B, A, generator=g) - 0.5) * 600,
        (torch.rand(B, A, generator=g) - 0.5) * 600,
        (torch.rand(B, A, generator=g) - 0.5) * 4 * math.pi,
        (torch.rand(B, A, generator=g) - 0.3) * 20,
    ], dim=-1)
    action = (torch.rand(B, A, 2, generator=g) - 0.5) * 2
    action4 = (torch.rand(B, A, 4, generator=g) - 0.5) * 2
    lr = 1.0 + torch.rand(B, A, generator=g)
    d = dict(state=npy(state), action=npy(action), action4=npy(action4), lr=npy(lr))

    def run_bicycle(cls, **kw):
        m = cls(**kw)
        m.set_params(lr=lr.clone())
        m.set_state(state.clone())
        m.step(action.clone())
        return m

    d['out_bicycle'] = npy(run_bicycle(KinematicBicycle).get_state())
    d['out_bicycle_lh'] = npy(run_bicycle(KinematicBicycle, left_handed=True).get_state())
    d['out_bicycle_dt'] = None
    m = KinematicBicycle()
    m.set_params(lr=lr.clone())
    m.set_state(state.clone())
    m.step(action.clone(), dt=0.25)
    d['out_bicycle_dt'] = npy(m.get_state())
    d['out_norev'] = npy(run_bicycle(BicycleNoReversing).get_state())
    # two chained steps (state tensor is replaced, never mutated)
    m = run_bicycle(KinematicBicycle)
    m.step(action.flip(0).clone())
    d['out_bicycle_2steps'] = npy(m.get_state())

    for name, cls in (('simple', SimpleKinematicModel), ('oriented', OrientedKinematicModel)):
        m = cls()
        m.set_state(state.clone())
        m.step(action4.clone())
        d[f'out_{name}'] = npy(m.get_state())

    # fit_action
    future = state + torch.stack([
        (torch.rand(B, A, generator=g) - 0.5) * 3,
        (torch.rand(B, A, generator=g) - 0.5) * 3,
        (torch.rand(B, A, generator=g) - 0.5) * 0.4,
        (torch.rand(B, A, generator=g) - 0.5) * 2,
    ], dim=-1)
    future[0, 0, :2] = state[0, 0, :2]  # zero displacement -> sign(abs(v)) == 0 branch
    d['future'] = npy(future)
    for name, kw in (('fit_bicycle', {}), ('fit_bicycle_lh', dict(left_handed=True))):
        m = KinematicBicycle(**kw)
        m.set_params(lr=lr.clone())
        m.set_state(state.clone())
        d[name] = npy(m.fit_action(future.clone()))
    for name, cls in (('fit_simple', SimpleKinematicModel), ('fit_oriented', OrientedKinematicModel)):
        m = cls()
        m.set_state(state.clone())
        d[name] = npy(m.fit_action(future.clone()))

    # known-answer
    m = KinematicBicycle()
    m.set_params(lr=torch.tensor([[1.5]]))
    m.set_state(torch.tensor([[[1.0, 2.0, 0.5, 3.0]]]))
    m.step(torch.tensor([[[0.4, -0.2]]]))
    d['kat_out'] = npy(m.get_state())
    np.savez_compressed(os.path.join(out, 'g1_kinematic.npz'), **d)
    print('g1', {k: v.shape for k, v in d.items()})


def gen_preraster_untrimmed(out, cv2, town):
    """G15: the call list of CV2Renderer with trim_mesh_before_rendering=False (rendering/cv2.py:15,32-41): EVERY face of the mesh goes to
    fillConvexPoly, in painter order.  A file of its own (the other fixtures keep their bytes).  Two huge triangles that have no vertex in any view
    are added to the background: trimmed they vanish, untrimmed they are drawn."""
    from torchdrivesim.mesh import BirdviewMesh
    from torchdrivesim.rendering import CV2RendererConfig, renderer_from_config
    g = seeded(1501)
    centre = (100.0, 60.0)
    crop = crop_mesh(town, centre, 30.0)
    v, f, vc = crop.verts[0], crop.faces[0], crop.vert_category[0]
    big = torch.tensor([[-400.0, -300.0], [600.0, -250.0], [150.0, 700.0], [-350.0, 500.0], [700.0, 400.0], [90.0, 55.0]]) + 0.25
    nv = v.shape[0]
    road_i = list(crop.categories).index('road')
    mesh = BirdviewMesh(verts=torch.cat([v, big])[None], faces=torch.cat([torch.tensor([[nv, nv + 1, nv + 2], [nv + 3, nv + 4, nv + 5]]), f])[None],
                        categories=list(crop.categories), colors=dict(crop.colors), zs=dict(crop.zs),
                        vert_category=torch.cat([vc, torch.full((6,), road_i, dtype=vc.dtype)])[None])
    B, A = 1, 4
    state = torch.cat([torch.tensor(centre) + (torch.rand(B, A, 2, generator=g) - 0.5) * 30, (torch.rand(B, A, 1, generator=g) - 0.5) * 2 * math.pi,
                       torch.zeros(B, A, 1)], -1)
    size = torch.tensor([4.5, 2.0]).expand(B, A, 2).clone()
    present = torch.tensor([[True, True, False, True]])
    d = dict(state=npy(state), size=npy(size), present=npy(present), road_verts=npy(mesh.verts[0]), road_faces=npy(mesh.faces[0]).astype(np.int32),
             road_vert_category=npy(mesh.vert_category[0]).astype(np.uint8), res=np.array(96), fov=np.array(35.0))
    for name, trim in (('untrimmed', False), ('trimmed', True)):
        r = renderer_from_config(CV2RendererConfig(trim_mesh_before_rendering=trim))
        sim = make_sim(state.clone(), size.clone(), present.clone(), road_mesh=mesh.expand(B), renderer=r)
        tris, cols, shape = render_record(cv2, sim, 96, 35.0)
        d[f'{name}_tris'], d[f'{name}_cols'] = tris, cols
        print('g15', name, tris.shape)
    psi = state[..., 2:3]
    d['cam_sc'] = npy(torch.cat([torch.sin(psi), torch.cos(psi)], -1))
    np.savez_compressed(os.path.join(out, 'g15_preraster_untrimmed.npz'), **d)


def gen_kinematic_displacement(out):
    """G1b: BicycleByDisplacement / BicycleByOrientedDisplacement (kinematic.py:526-587), a file of its own so that g1_kinematic.npz keeps its bytes"""
    from torchdrivesim.kinematic import BicycleByDisplacement, BicycleByOrientedDisplacement
    g = seeded(131)
    B, A = 3, 7
    state = torch.stack([(torch.rand(B, A, generator=g) - 0.5) * 600, (torch.rand(B, A, generator=g) - 0.5) * 600,
                         (torch.rand(B, A, generator=g) - 0.5) * 4 * math.pi, (torch.rand(B, A, generator=g) - 0.3) * 20], dim=-1)
    action = (torch.rand(B, A, 2, generator=g) - 0.5) * 2
    action[0, 0] = 0.0                                                   # no displacement: the sign(abs(v)) == 0 branch of the fitted steering
    lr = 1.0 + torch.rand(B, A, generator=g)
    future = state + torch.stack([(torch.rand(B, A, generator=g) - 0.5) * 3, (torch.rand(B, A, generator=g) - 0.5) * 3,
                                  (torch.rand(B, A, generator=g) - 0.5) * 0.4, (torch.rand(B, A, generator=g) - 0.5) * 2], dim=-1)
    d = dict(state=npy(state), action=npy(action), lr=npy(lr), future=npy(future))
    for name, cls, kw in (('disp', BicycleByDisplacement, {}), ('disp_max5', BicycleByDisplacement, dict(max_dx=5, dt=0.2)),
                          ('oriented', BicycleByOrientedDisplacement, {})):
        m = cls(**kw)
        m.set_params(lr=lr.clone())
        m.set_state(state.clone())
        d[f'fit_{name}'] = npy(m.fit_action(future.clone()))
        m.step(action.clone())
        d[f'out_{name}'] = npy(m.get_state())
        m.step(action.flip(1).clone(), dt=0.05)
        d[f'out_{name}_2steps'] = npy(m.get_state())
    np.savez_compressed(os.path.join(out, 'g1b_bicycle_displacement.npz'), **d)
    print('g1b', {k: v.shape for k, v in d.items()})


# --------------------------------------------------------------------------------------
# G2 boxes / IoU / discs / collision
# --------------------------------------------------------------------------------------
def curated_pairs():
    base = [0.0, 0.0, 4.0, 2.0, 0.0]
    p = [
        (base, base),  # identical
        (base, [2.0, 0.0, 4.0, 2.0, 0.0]),  # shifted
        (base, [0.0, 0.0, 4.0, 2.0, math.pi / 2]),  # perpendicular
        (base, [0.0, 0.0, 4.0, 2.0, math.pi / 4]),
        (base, [1.0, 0.5, 4.0, 2.0, math.radians(30)]),
        (base, [0.0, 0.0, 2.0, 1.0, 0.3]),  # contained
        (base, [4.0, 0.0, 4.0, 2.0, 0.0]),  # exactly touching
        (base, [0.0, 2.0, 4.0, 2.0, 0.0]),  # touching along long edge
        (base, [10.0, 10.0, 4.0, 2.0, 1.0]),  # far apart
        (base, [3.9, 0.0, 4.0, 2.0, 0.0]),  # sliver overlap
        (base, [0.0, 0.0, 1e-3, 1e-3, 0.0]),  # tiny
        ([0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0, 0.0, 0.0]),  # zero size
        ([5.0, -3.0, 4.5, 1.9, 2.0], [6.0, -2.0, 4.8, 2.1, -1.0]),
        ([5.0, -3.0, 1.9, 4.5, 2.0], [6.0, -2.0, 2.1, 4.8, -1.0]),  # wid > len (discs yaw+pi/2 branch)
        ([400.0, 330.0, 4.5, 2.0, 0.7], [401.5, 331.0, 4.5, 2.0, 2.4]),  # far from origin
    ]
    b1 = torch.tensor([a for a, _ in p], dtype=torch.float32)
    b2 = torch.tensor([b for _, b in p], dtype=torch.float32)
    return b1, b2


def gen_collision(out):
    from torchdrivesim import _iou_utils as iu
    from torchdrivesim.infractions import iou_differentiable, collision_detection_with_discs
    d = {}
    b1, b2 = curated_pairs()
    d['cur_box1'], d['cur_box2'] = npy(b1), npy(b2)
    d['cur_corners1'] = npy(iu.box2corners_th(b1[None]))[0]
    d['cur_iou'] = npy(iou_differentiable(b1[None], b2[None]))[0]
    d['cur_discs'] = npy(collision_detection_with_discs(b1[None], b2[None]))[0]

    def stages(box1, box2):
        c1, c2 = iu.box2corners_th(box1), iu.box2corners_th(box2)
        inters, mask_inter = iu.box_intersection_th(c1, c2)
        c12, c21 = iu.box_in_box_th(c1, c2)
        vertices, mask = iu.build_vertices(c1, c2, c12, c21, inters, mask_inter)
        idx = iu.sort_indices(vertices, mask.clone())
        area, _ = iu.calculate_area(idx, vertices)
        return vertices, mask, idx, area

    v, m, idx, area = stages(b1[None], b2[None])
    d['cur_vertices'], d['cur_mask'], d['cur_idx'], d['cur_area'] = npy(v)[0], npy(m)[0], npy(idx)[0], npy(area)[0]

    g = seeded(202)
    for tag, off in (('rnd0', (0.0, 0.0)), ('rnd400', (400.0, 330.0))):
        N = 4000
        c = (torch.rand(N, 2, generator=g) - 0.5) * 4
        box1 = torch.cat([c + torch.tensor(off), 3.5 + 2 * torch.rand(N, 1, generator=g),
                          1.5 + torch.rand(N, 1, generator=g), (torch.rand(N, 1, generator=g) - 0.5) * 2 * math.pi], -1)
        c2 = c + (torch.rand(N, 2, generator=g) - 0.5) * 9
        box2 = torch.cat([c2 + torch.tensor(off), 3.5 + 2 * torch.rand(N, 1, generator=g),
                          1.5 + torch.rand(N, 1, generator=g), (torch.rand(N, 1, generator=g) - 0.5) * 2 * math.pi], -1)
        d[f'{tag}_box1'], d[f'{tag}_box2'] = npy(box1), npy(box2)
        d[f'{tag}_iou'] = npy(iou_differentiable(box1[None], box2[None]))[0]
        d[f'{tag}_discs'] = npy(collision_detection_with_discs(box1[None], box2[None]))[0]
        v, m, idx, area = stages(box1[None], box2[None])
        d[f'{tag}_nvalid'] = npy(m.sum(-1))[0].astype(np.int8)
        d[f'{tag}_idx'] = npy(idx)[0].astype(np.int8)
        d[f'{tag}_area'] = npy(area)[0]
    np.savez_compressed(os.path.join(out, 'g2_boxes.npz'), **d)
    print('g2 boxes ok; valid-count histogram', np.bincount(d['rnd0_nvalid']))


def make_sim(state, size, present, lr=None, metric='iou', road_mesh=None, npc=None, left_handed=False,
             renderer=None, agent_types=None, agent_type_names=None):
    from torchdrivesim.simulator import Simulator, TorchDriveConfig, CollisionMetric, NPCController
    from torchdrivesim.kinematic import KinematicBicycle
    from torchdrivesim.mesh import BirdviewMesh
    from torchdrivesim.rendering import DummyRendererConfig
    B = state.shape[0]
    if road_mesh is None:
        road_mesh = BirdviewMesh.empty(batch_size=B)
    km = KinematicBicycle()
    km.set_params(lr=lr if lr is not None else torch.full(state.shape[:2], 1.5))
    km.set_state(state)
    cfg = TorchDriveConfig(collision_metric=CollisionMetric(metric), left_handed_coordinates=left_handed,
                           renderer=DummyRendererConfig())
    npc_controller = None
    if npc is not None:
        npc_controller = NPCController(npc_size=npc['size'], npc_state=npc['state'], npc_present_mask=npc['present'],
                                       npc_types=torch.zeros_like(npc['present']).long())
    return Simulator(road_mesh, km, size, present, cfg, renderer=renderer, npc_controller=npc_controller,
                     agent_types=agent_types, agent_type_names=agent_type_names)


def random_scene(g, B, A, spread=12.0, centre=(0.0, 0.0)):
    xy = (torch.rand(B, A, 2, generator=g) - 0.5) * spread + torch.tensor(centre)
    psi = (torch.rand(B, A, 1, generator=g) - 0.5) * 2 * math.pi
    v = torch.rand(B, A, 1, generator=g) * 10
    state = torch.cat([xy, psi, v], -1)
    size = torch.cat([4.5 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g)),
                      2.0 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g))], -1)
    present = torch.rand(B, A, generator=g) < 0.8
    present[:, 0] = True
    return state, size, present


def gen_scene_collision(out):
    d = {}
    g = seeded(303)
    state, size, present = random_scene(g, 6, 8)
    # quirk scene: three co-located boxes, present=[F,T,T]
    state[0, :3] = torch.tensor([[0., 0., 0., 0.], [0., 0., 0., 0.], [30., 30., 0., 0.]])
    size[0, :3] = torch.tensor([[4., 2.], [4., 2.], [4., 2.]])
    present[0, :3] = torch.tensor([False, True, True])
    state[1, 2, 0] = float('nan')  # NaN scrubbing path (simulator.py:1095-1103)
    d['state'], d['size'], d['present'] = npy(state), npy(size), npy(present)
    for metric in ('iou', 'discs'):
        sim = make_sim(state.clone(), size.clone(), present.clone(), metric=metric)
        d[f'coll_{metric}'] = npy(sim.compute_collision())
    # with NPCs
    ns, nz, npres = random_scene(g, 6, 3)
    d['npc_state'], d['npc_size'], d['npc_present'] = npy(ns), npy(nz), npy(npres)
    for metric in ('iou', 'discs'):
        sim = make_sim(state.clone(), size.clone(), present.clone(), metric=metric,
                       npc=dict(state=ns.clone(), size=nz.clone(), present=npres.clone()))
        d[f'coll_npc_{metric}'] = npy(sim.compute_collision())
    # far from the origin (fp32 translation sensitivity)
    state2, size2, present2 = random_scene(g, 4, 8, centre=(380.0, 310.0))
    d['far_state'], d['far_size'], d['far_present'] = npy(state2), npy(size2), npy(present2)
    for metric in ('iou', 'discs'):
        sim = make_sim(state2.clone(), size2.clone(), present2.clone(), metric=metric)
        d[f'far_coll_{metric}'] = npy(sim.compute_collision())
    np.savez_compressed(os.path.join(out, 'g2_scene_collision.npz'), **d)
    print('g2 scenes: quirk row', d['coll_iou'][0, :3], d['coll_discs'][0, :3])


# --------------------------------------------------------------------------------------
# Town01 data fixture + crops
# --------------------------------------------------------------------------------------
def load_town01():
    from torchdrivesim.mesh import BirdviewMesh
    path = os.path.join(REF, 'torchdrivesim/resources/maps/carla_Town01/carla_Town01_mesh.json')
    return BirdviewMesh.load(path)


def crop_mesh(mesh, centre, half):
    """Host-side crop used only to build a small fixture: faces with >=1 vertex inside the box."""
    from torchdrivesim.mesh import BirdviewMesh
    v = mesh.verts[0]
    f = mesh.faces[0]
    inside = ((v - torch.tensor(centre)).abs() <= half).all(-1)
    keep = inside[f].any(-1)
    f = f[keep]
    used = torch.unique(f)
    remap = torch.full((v.shape[0],), -1, dtype=torch.long)
    remap[used] = torch.arange(len(used))
    return BirdviewMesh(verts=v[used][None].clone(), faces=remap[f][None].clone(), categories=list(mesh.categories),
                        colors=dict(mesh.colors), zs=dict(mesh.zs), vert_category=mesh.vert_category[0][used][None].clone())


def gen_town01(out):
    m = load_town01()
    np.savez_compressed(os.path.join(out, 'town01_mesh.npz'),
                        verts=npy(m.verts[0]).astype(np.float32), faces=npy(m.faces[0]).astype(np.int32),
                        vert_category=npy(m.vert_category[0]).astype(np.uint8),
                        categories=np.array(m.categories))
    # a tiny json in the reference's own serialisation format (mesh.py:700-719) for the loader test
    small = crop_mesh(m, (100.0, 2.0), 6.0)
    small.save(os.path.join(out, 'town01_crop_small_mesh.json'))
    print('town01', m.verts.shape, m.faces.shape, 'small crop', small.verts.shape, small.faces.shape)
    return m


# --------------------------------------------------------------------------------------
# G3 offroad
# --------------------------------------------------------------------------------------
def gen_offroad(out, town):
    from torchdrivesim.infractions import offroad_infraction_loss
    from torchdrivesim.mesh import BaseMesh, BirdviewMesh
    d = {}
    # (a) 2-triangle road x in [-20,20], y in [-3,3]  (SURVEY R4 probe: 414.952)
    verts = torch.tensor([[[-20., -3.], [20., -3.], [20., 3.], [-20., 3.]]])
    faces = torch.tensor([[[0, 1, 2], [0, 2, 3]]])
    road = BaseMesh(verts=verts, faces=faces)
    st = torch.tensor([[[30., 0., 0.3, 0.], [0., 0., 0.1, 0.], [19., 2.5, 1.0, 0.], [0., 3.4, 0., 0.], [-21.5, 0., 0., 0.]]])
    lw = torch.tensor([[[4., 2.]]]).expand(1, 5, 2).contiguous()
    d['a_verts'], d['a_faces'], d['a_state'], d['a_lenwid'] = npy(verts[0]), npy(faces[0]), npy(st), npy(lw)
    d['a_off_t05'] = npy(offroad_infraction_loss(st, lw, road, threshold=0.5, use_pytorch3d=False))
    d['a_off_t0'] = npy(offroad_infraction_loss(st, lw, road, threshold=0.0, use_pytorch3d=False))
    # (b) Town01 crop, two scenes; second scene's mesh is padded with [0,0,0] faces via collate
    c1 = crop_mesh(town, (100.0, 2.0), 14.0)
    c2 = crop_mesh(town, (92.0, 60.0), 9.0)
    both = BirdviewMesh.collate([c1, c2])
    g = seeded(404)
    B, A = 2, 6
    centres = torch.tensor([[100.0, 2.0], [92.0, 60.0]])
    xy = centres[:, None] + (torch.rand(B, A, 2, generator=g) - 0.5) * 24
    st = torch.cat([xy, (torch.rand(B, A, 1, generator=g) - 0.5) * 6.3, torch.zeros(B, A, 1)], -1)
    lw = torch.cat([4.5 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g)), 2.0 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g))], -1)
    d['b_verts'], d['b_faces'] = npy(both.verts), npy(both.faces).astype(np.int32)
    d['b_nfaces'] = np.array([c1.faces_count, c2.faces_count])
    d['b_state'], d['b_lenwid'] = npy(st), npy(lw)
    d['b_off_t05'] = npy(offroad_infraction_loss(st, lw, both, threshold=0.5, use_pytorch3d=False))
    d['b_off_t0'] = npy(offroad_infraction_loss(st, lw, both, threshold=0.0, use_pytorch3d=False))
    # (c) through the Simulator (present-mask multiply, simulator.py:1043-1044), lenwid (B,2) form
    present = torch.rand(B, A, generator=g) < 0.7
    sim = make_sim(st.clone(), lw.clone(), present.clone(), road_mesh=both)
    d['c_present'] = npy(present)
    d['c_sim_offroad'] = npy(sim.compute_offroad())
    np.savez_compressed(os.path.join(out, 'g3_offroad.npz'), **d)
    print('g3 offroad', d['a_off_t05'], 'crop faces', d['b_nfaces'])


# --------------------------------------------------------------------------------------
# G4 / G5 mesh generation + pre-raster call lists
# --------------------------------------------------------------------------------------
def render_record(cv2, sim, res, fov, **kw):
    from torchdrivesim.utils import Resolution
    cv2.calls.clear()
    img = sim.render_egocentric(res=Resolution(res, res), fov=fov, **kw)
    n_img = img.shape[0] * img.shape[1]
    calls = list(cv2.calls)
    assert len(calls) % n_img == 0
    per = len(calls) // n_img
    tris = np.stack([c[0] for c in calls]).reshape(n_img, per, 3, 2).astype(np.int32)
    cols = np.array([c[1] for c in calls], dtype=np.uint8).reshape(n_img, per, 3)
    return tris, cols, tuple(img.shape)


def gen_mesh_and_preraster(out, cv2, town):
    from torchdrivesim.mesh import BirdviewRGBMeshGenerator, BirdviewMesh
    from torchdrivesim.rendering import CV2RendererConfig, renderer_from_config
    from torchdrivesim.rendering.base import get_default_color_map, get_default_rendering_levels
    d = {}
    g = seeded(505)
    # ---- G4: actor template + generate() on a small background
    B, A = 2, 4
    state, size, present = random_scene(g, B, A, spread=20.0, centre=(100.0, 2.0))
    small = crop_mesh(town, (100.0, 2.0), 6.0)
    bg = small.expand(B)
    gen = BirdviewRGBMeshGenerator(background_mesh=bg, color_map=get_default_color_map(),
                                   rendering_levels=get_default_rendering_levels())
    types_ = torch.zeros(B, A, dtype=torch.long)
    gen.initialize_actors_mesh(size, types_, ['vehicle'])
    d['g4_size'], d['g4_state'], d['g4_present'] = npy(size), npy(state), npy(present)
    d['g4_bg_verts'], d['g4_bg_faces'] = npy(small.verts[0]), npy(small.faces[0]).astype(np.int32)
    d['g4_bg_vert_category'] = npy(small.vert_category[0]).astype(np.uint8)
    d['g4_tmpl_verts'], d['g4_tmpl_faces'], d['g4_tmpl_attrs'] = npy(gen.actor_mesh.verts), npy(gen.actor_mesh.faces).astype(np.int32), npy(gen.actor_mesh.attrs)
    nc = A
    mask = present[:, None].expand(B, nc, A)
    rgb = gen.generate(nc, agent_state=state[:, None].expand(-1, nc, -1, -1), present_mask=mask)
    nbg = small.verts_count
    d['g4_gen_actor_verts'] = npy(rgb.verts[:, nbg:])
    d['g4_gen_actor_faces'] = npy(rgb.faces[:, small.faces_count:]).astype(np.int32)
    d['g4_gen_actor_attrs'] = npy(rgb.attrs[:, nbg:])
    d['g4_gen_bg_attrs_z'] = np.concatenate([npy(rgb.attrs[0, :nbg]), npy(rgb.verts[0, :nbg, 2:3])], -1)

    # ---- G5: full Simulator.render_egocentric with the CV2 backend up to the OpenCV boundary
    def cv2_sim(state, size, present, road, left_handed=False, **kw):
        r = renderer_from_config(CV2RendererConfig(left_handed_coordinates=left_handed))
        return make_sim(state.clone(), size.clone(), present.clone(), road_mesh=road, renderer=r, left_handed=left_handed, **kw)

    cases = []
    # case 0: config-1 shape: full Town01, B=2 x A=8, 128x128, fov 35
    B, A = 2, 8
    vroad = town.verts[0][town.vert_category[0] == town.categories.index('road')]
    pick = torch.randint(0, vroad.shape[0], (B, A), generator=g)
    xy = vroad[pick] + torch.randn(B, A, 2, generator=g)
    # keep agents of one scene within sight of each other
    xy = xy[:, :1] + (xy - xy[:, :1]) * 0 + (torch.rand(B, A, 2, generator=g) - 0.5) * 30
    state = torch.cat([xy, (torch.rand(B, A, 1, generator=g) - 0.5) * 2 * math.pi, torch.rand(B, A, 1, generator=g) * 10], -1)
    size = torch.cat([4.5 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g)), 2.0 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g))], -1)
    present = torch.rand(B, A, generator=g) < 0.85
    present[:, 0] = True
    cases.append(dict(name='town01_128', state=state, size=size, present=present, res=128, fov=35.0, lh=False, road='town01'))
    # case 1: same scene, left-handed flag, 64x64, fov 50, agent 0 of scene 1 absent (stray-dot quirk Q10)
    p2 = present.clone()
    p2[1, 0] = False
    cases.append(dict(name='town01_64_lh', state=state, size=size, present=p2, res=64, fov=50.0, lh=True, road='town01'))
    # case 2: crop background, 256x256 fov 35, not ego-rotated
    cases.append(dict(name='crop_256_norot', state=state[:1, :4].clone(), size=size[:1, :4].clone(), present=present[:1, :4].clone(),
                      res=256, fov=35.0, lh=False, road='crop', centre=state[0, 0, :2].tolist(), ego_rotate=False))
    # case 3: empty road mesh, agents only, small res (reference test shape tests/test_rendering.py)
    st3 = torch.tensor([[[0., 0., 0., 0.], [3., 1., 0.5, 0.], [-2., 4., 2.0, 0.]]])
    cases.append(dict(name='empty_32', state=st3, size=torch.tensor([[[4., 2.], [5., 2.2], [1., 1.]]]), present=torch.ones(1, 3, dtype=torch.bool),
                      res=32, fov=20.0, lh=False, road='empty'))
    meta = []
    for c in cases:
        if c['road'] == 'town01':
            road = town.expand(c['state'].shape[0])
        elif c['road'] == 'crop':
            crop = crop_mesh(town, c['centre'], 40.0)
            road = crop.expand(c['state'].shape[0])
            d[f"g5_{c['name']}_road_verts"] = npy(crop.verts[0])
            d[f"g5_{c['name']}_road_faces"] = npy(crop.faces[0]).astype(np.int32)
            d[f"g5_{c['name']}_road_vert_category"] = npy(crop.vert_category[0]).astype(np.uint8)
        else:
            road = BirdviewMesh.empty(batch_size=c['state'].shape[0])
        sim = cv2_sim(c['state'], c['size'], c['present'], road, left_handed=c['lh'])
        kw = {}
        if 'ego_rotate' in c:
            kw['ego_rotate'] = c['ego_rotate']
        tris, cols, shape = render_record(cv2, sim, c['res'], c['fov'], **kw)
        n = c['name']
        d[f'g5_{n}_state'], d[f'g5_{n}_size'], d[f'g5_{n}_present'] = npy(c['state']), npy(c['size']), npy(c['present'])
        d[f'g5_{n}_tris'], d[f'g5_{n}_cols'] = tris, cols
        # camera sin/cos exactly as the reference computes them (simulator.py:940)
        psi = c['state'][..., 2:3] if kw.get('ego_rotate', True) else torch.ones_like(c['state'][..., 2:3]) * (np.pi / 2)
        d[f'g5_{n}_cam_sc'] = npy(torch.cat([torch.sin(psi), torch.cos(psi)], -1))
        meta.append(dict(name=n, res=c['res'], fov=c['fov'], left_handed=c['lh'], road=c['road'],
                         ego_rotate=kw.get('ego_rotate', True), out_shape=list(shape)))
        print('g5', n, tris.shape, 'distinct colours', len({tuple(x) for x in cols.reshape(-1, 3).tolist()}))
    d['g5_meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(out, 'g45_mesh_preraster.npz'), **d)


# --------------------------------------------------------------------------------------
# G7 gradients
# --------------------------------------------------------------------------------------
def gen_grads(out, town):
    d = {}
    g = seeded(707)
    B, A = 2, 6
    crop = crop_mesh(town, (100.0, 2.0), 14.0)
    road = crop.expand(B)
    xy = torch.tensor([100.0, 2.0]) + (torch.rand(B, A, 2, generator=g) - 0.5) * torch.tensor([14.0, 8.0])
    state0 = torch.cat([xy, (torch.rand(B, A, 1, generator=g) - 0.5) * 1.0, 2 + 6 * torch.rand(B, A, 1, generator=g)], -1)
    size = torch.cat([4.5 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g)), 2.0 * (0.9 + 0.2 * torch.rand(B, A, 1, generator=g))], -1)
    present = torch.ones(B, A, dtype=torch.bool)
    present[1, 3] = False
    action = (torch.rand(B, A, 2, generator=g) - 0.5) * 2
    lr = 1.0 + torch.rand(B, A, generator=g)
    d.update(state0=npy(state0), size=npy(size), present=npy(present), action=npy(action), lr=npy(lr),
             road_verts=npy(crop.verts[0]), road_faces=npy(crop.faces[0]).astype(np.int32))
    for metric in ('iou', 'discs'):
        s = state0.clone().requires_grad_(True)
        a = action.clone().requires_grad_(True)
        sim = make_sim(s, size.clone(), present.clone(), lr=lr.clone(), metric=metric, road_mesh=road)
        sim.step(a)
        coll = sim.compute_collision()
        off = sim.compute_offroad()
        loss = coll.sum() + off.sum()
        loss.backward()
        d[f'{metric}_state1'] = npy(sim.get_state())
        d[f'{metric}_coll'], d[f'{metric}_off'] = npy(coll), npy(off)
        d[f'{metric}_grad_state'], d[f'{metric}_grad_action'] = npy(s.grad), npy(a.grad)
        # separate grads of each term w.r.t. the post-step state (isolates K2 backward from K1 backward)
        s1 = sim.get_state().detach().clone().requires_grad_(True)
        sim2 = make_sim(s1, size.clone(), present.clone(), lr=lr.clone(), metric=metric, road_mesh=road)
        c2 = sim2.compute_collision().sum()
        d[f'{metric}_grad_coll_wrt_state1'] = npy(torch.autograd.grad(c2, s1)[0])
        o2 = sim2.compute_offroad().sum()
        d[f'{metric}_grad_off_wrt_state1'] = npy(torch.autograd.grad(o2, s1)[0])
    np.savez_compressed(os.path.join(out, 'g7_grads.npz'), **d)
    print('g7 |grad_state|', np.abs(d['iou_grad_state']).max(), np.abs(d['discs_grad_state']).max())


# --------------------------------------------------------------------------------------
# G8: traffic controls: map metadata + stop lines of Town01 (data files of the reference's resources) and
#     TrafficLightControl.compute_violation / the state replay logic on them
# --------------------------------------------------------------------------------------
def gen_traffic(out):
    import shutil
    from torchdrivesim.traffic_controls import TrafficLightControl
    src = os.path.join(REF, 'torchdrivesim', 'resources', 'maps', 'carla_Town01')
    dst = os.path.join(out, 'maps', 'carla_Town01')
    os.makedirs(dst, exist_ok=True)
    for name in ('metadata.json', 'carla_Town01_stoplines.json'):          # data files, not source
        shutil.copyfile(os.path.join(src, name), os.path.join(dst, name))
    stop = json.load(open(os.path.join(src, 'carla_Town01_stoplines.json')))
    pos1 = torch.tensor([[s['x'], s['y'], s['length'], s['width'], s['orientation']] for s in stop if s['agent_type'] == 'traffic_light'])
    g = seeded(81)
    B, A, T = 3, 64, 5
    N = pos1.shape[0]
    pos = pos1.unsqueeze(0).expand(B, -1, -1).clone()
    mask = torch.rand(B, N, generator=g) < 0.85
    replay = torch.randint(0, 3, (B, N, T), generator=g)
    ctl = TrafficLightControl(pos, replay_states=replay, mask=mask)
    # agents: around randomly chosen stop lines (so that many rear boxes overlap one), random headings and sizes
    pick = torch.randint(0, N, (B, A), generator=g)
    centre = torch.gather(pos[..., :2], 1, pick[..., None].expand(-1, -1, 2))
    xy = centre + torch.randn(B, A, 2, generator=g) * 1.5
    lw = torch.stack([torch.rand(B, A, generator=g) * 2 + 3.5, torch.rand(B, A, generator=g) * 0.6 + 1.7], -1)
    psi = (torch.rand(B, A, 1, generator=g) * 2 - 1) * math.pi
    boxes = torch.cat([xy, lw, psi], -1)
    d = dict(pos=npy(pos), mask=npy(mask), replay=npy(replay), boxes=npy(boxes), corners=npy(ctl.corners))
    for t in range(T + 2):                                       # beyond T the state repeats (compute_state default)
        ctl.step(t)
        d[f'state_{t}'] = npy(ctl.state)
        d[f'violation_{t}'] = npy(ctl.compute_violation(boxes))
    np.savez_compressed(os.path.join(out, 'g8_traffic.npz'), **d)

    # G9: the RGB mesh the reference generates for traffic controls (stop lines as quads, lights coloured by state,
    # mesh.py:1007-1051,1105-1118,1150-1153) on a small background
    from torchdrivesim.mesh import BirdviewRGBMeshGenerator
    from torchdrivesim.rendering.base import get_default_color_map, get_default_rendering_levels
    from torchdrivesim.traffic_controls import StopSignControl, YieldControl
    g = seeded(99)
    town = load_town01()
    small = crop_mesh(town, (100.0, 2.0), 6.0)
    B, A, nc = 2, 3, 3
    state, size, present = random_scene(g, B, A, spread=20.0, centre=(100.0, 2.0))
    mk = lambda n: torch.cat([torch.tensor([100.0, 2.0]) + (torch.rand(B, n, 2, generator=g) - 0.5) * 24,
                              torch.rand(B, n, 1, generator=g) * 1.0 + 0.8, torch.rand(B, n, 1, generator=g) * 3 + 3,
                              (torch.rand(B, n, 1, generator=g) - 0.5) * 2 * math.pi], -1)
    tl_pos, ss_pos, ys_pos = mk(4), mk(2), mk(1)
    tl_mask = torch.tensor([[True, True, False, True], [True, True, True, True]])
    tl_state = torch.randint(0, 3, (B, 4), generator=g)
    controls = dict(stop_sign=StopSignControl(ss_pos), traffic_light=TrafficLightControl(tl_pos, mask=tl_mask), yield_sign=YieldControl(ys_pos))
    controls['traffic_light'].set_state(tl_state)
    gen = BirdviewRGBMeshGenerator(background_mesh=small.expand(B), color_map=get_default_color_map(),
                                   rendering_levels=get_default_rendering_levels())
    gen.initialize_actors_mesh(size, torch.zeros(B, A, dtype=torch.long), ['vehicle'])
    gen.initialize_traffic_controls_mesh(controls)
    rgb = gen.generate(nc, agent_state=state[:, None].expand(-1, nc, -1, -1), present_mask=present[:, None].expand(B, nc, A),
                       traffic_lights=controls['traffic_light'].extend(nc, in_place=False))
    d9 = dict(state=npy(state), size=npy(size), present=npy(present), tl_pos=npy(tl_pos), tl_mask=npy(tl_mask), tl_state=npy(tl_state),
              ss_pos=npy(ss_pos), ys_pos=npy(ys_pos), bg_verts=npy(small.verts[0]), bg_faces=npy(small.faces[0]).astype(np.int32),
              bg_vert_category=npy(small.vert_category[0]).astype(np.uint8),
              rgb_verts=npy(rgb.verts), rgb_faces=npy(rgb.faces).astype(np.int32), rgb_attrs=npy(rgb.attrs))
    np.savez_compressed(os.path.join(out, 'g9_traffic_mesh.npz'), **d9)
    print('g9 mesh', tuple(rgb.verts.shape), tuple(rgb.faces.shape))
    print('g8 violations per step', [int(d[f'violation_{t}'].sum()) for t in range(T + 2)], 'of', B * A)


# --------------------------------------------------------------------------------------
# G10: observation model for non-visual policies: occlusion mask of StandardSensingObservationNoise, the distance-dependent
#      standard deviation of its state noise, and get_noisy_all_agents_relative with the noise-free base model
# --------------------------------------------------------------------------------------
def gen_observation(out):
    from torchdrivesim.observation_noise import ObservationNoise, ObservationNoiseConfig, StandardSensingObservationNoise, \
        StandardSensingObservationNoiseConfig
    g = seeded(606)
    d = {}
    for tag, (B, A, P, spread) in dict(a=(3, 6, 4, 40.0), b=(2, 12, 0, 120.0), c=(2, 1, 5, 30.0)).items():
        state, size, present = random_scene(g, B, A + P, spread=spread)
        # put some agents exactly in line behind others (certain occlusions) and two at the same place
        state[0, 2, :2] = state[0, 0, :2] + 2.0 * (state[0, 1, :2] - state[0, 0, :2])
        if A + P > 4:
            state[1, 3, :2] = state[1, 4, :2]
        npc = dict(state=state[:, A:].clone(), size=size[:, A:].clone(), present=present[:, A:].clone()) if P else None
        sim = make_sim(state[:, :A].clone(), size[:, :A].clone(), present[:, :A].clone(), npc=npc)
        std = StandardSensingObservationNoise(StandardSensingObservationNoiseConfig())
        d[f'{tag}_state'], d[f'{tag}_size'], d[f'{tag}_present'] = npy(state), npy(size), npy(present)
        d[f'{tag}_n_exposed'] = np.array(A)
        d[f'{tag}_mask'] = npy(std.get_noisy_present_mask(sim))
        # the deterministic part of get_noisy_state: standard deviation per (ego, entity)
        torch.manual_seed(0)
        noisy = std.get_noisy_state(sim)
        torch.manual_seed(0)
        base = ObservationNoise(ObservationNoiseConfig()).get_noisy_state(sim)
        eps = torch.randn_like(base)
        dev_ = ((noisy - base) / eps)
        d[f'{tag}_deviation'] = npy(dev_[..., 0])
        sim.observation_noise_model = ObservationNoise(ObservationNoiseConfig())
        d[f'{tag}_noisy_relative'] = npy(sim.get_noisy_all_agents_relative())
        d[f'{tag}_noisy_absolute'] = npy(sim.get_noisy_all_agents_absolute())
    np.savez_compressed(os.path.join(out, 'g10_observation.npz'), **d)
    print('g10 occluded fraction', {t: float(1 - d[f'{t}_mask'].mean()) for t in 'abc'})


# --------------------------------------------------------------------------------------
# G11: waypoint goals ("waypoint meshes in the renderer"): WaypointGoal bookkeeping over a trajectory, the disc mesh,
#      generate() with per-camera waypoints, and the call list of render_egocentric with waypoint goals at the OpenCV boundary
# --------------------------------------------------------------------------------------
def gen_waypoints(out, cv2, town):
    from torchdrivesim.goals import WaypointGoal
    from torchdrivesim.mesh import BirdviewRGBMeshGenerator, generate_disc_mesh
    from torchdrivesim.rendering import CV2RendererConfig, renderer_from_config
    from torchdrivesim.rendering.base import get_default_color_map, get_default_rendering_levels
    g = seeded(808)
    d = {}
    # ---- bookkeeping: agents drive along x; waypoint collections are strung along their way, some off to the side (never reached)
    B, A, N, M, T = 2, 3, 4, 2, 12
    start = torch.cat([(torch.rand(B, A, 2, generator=g) - 0.5) * 10, torch.zeros(B, A, 2)], -1)
    wps = start[:, :, None, None, :2] + torch.stack([torch.arange(1, N + 1) * 6.0, torch.zeros(N)], -1)[None, None, :, None, :] \
        + (torch.rand(B, A, N, M, 2, generator=g) - 0.5) * torch.tensor([2.0, 6.0])
    mask = torch.rand(B, A, N, M, generator=g) < 0.8
    mask[0, 0] = True
    mask[1, 2, 1] = False                                     # a collection made of padding only
    goal = WaypointGoal(wps.clone(), mask.clone())
    d['wp'], d['wp_mask'] = npy(wps), npy(mask)
    traj = []
    for t in range(T):
        st = start.clone()
        st[..., 0] += 2.5 * (t + 1)
        st[..., 1] += 0.3 * math.sin(t)
        traj.append(st)
        goal.step(st, t + 1, threshold=2.0)
        d[f'state_{t}'], d[f'mask_{t}'] = npy(goal.state), npy(goal.mask)
        for c in (1, 3):
            d[f'get_wp_{c}_{t}'], d[f'get_mask_{c}_{t}'] = npy(goal.get_waypoints(c)), npy(goal.get_masks(c))
    d['traj'] = npy(torch.stack(traj))
    ext = goal.extend(2, in_place=False)
    d['ext_state'], d['ext_wp'] = npy(ext.state), npy(ext.waypoints)
    # ---- disc meshes
    for r, n in ((2.0, 10), (1.5, 6), (3.0, 2)):
        v, f = generate_disc_mesh(radius=r, num_triangles=n)
        d[f'disc_{n}_verts'], d[f'disc_{n}_faces'] = npy(v), npy(f).astype(np.int32)
    # ---- generate() with waypoints
    small = crop_mesh(town, (100.0, 2.0), 6.0)
    Bm, Am, Mw = 2, 3, 3
    state, size, present = random_scene(g, Bm, Am, spread=20.0, centre=(100.0, 2.0))
    gen = BirdviewRGBMeshGenerator(background_mesh=small.expand(Bm), color_map=get_default_color_map(), rendering_levels=get_default_rendering_levels())
    gen.initialize_actors_mesh(size, torch.zeros(Bm, Am, dtype=torch.long), ['vehicle'])
    wp = torch.tensor([100.0, 2.0]) + (torch.rand(Bm, Am, Mw, 2, generator=g) - 0.5) * 30
    wm = torch.rand(Bm, Am, Mw, generator=g) < 0.6
    wm[0, 0, 0] = False
    rgb = gen.generate(Am, agent_state=state[:, None].expand(-1, Am, -1, -1), present_mask=present[:, None].expand(Bm, Am, Am),
                       waypoints=wp, waypoints_rendering_mask=wm)
    nv0, nf0 = small.verts_count + 7 * Am, small.faces_count + 3 * Am
    d['m_state'], d['m_size'], d['m_present'], d['m_wp'], d['m_wmask'] = npy(state), npy(size), npy(present), npy(wp), npy(wm)
    d['m_bg_verts'], d['m_bg_faces'], d['m_bg_vert_category'] = npy(small.verts[0]), npy(small.faces[0]).astype(np.int32), npy(small.vert_category[0]).astype(np.uint8)
    d['m_wp_verts'], d['m_wp_faces'], d['m_wp_attrs'] = npy(rgb.verts[:, nv0:]), npy(rgb.faces[:, nf0:]).astype(np.int32), npy(rgb.attrs[:, nv0:])
    # ---- render_egocentric with waypoint goals, up to the OpenCV boundary
    crop = crop_mesh(town, (100.0, 2.0), 40.0)
    B2, A2 = 2, 4
    st2, sz2, pr2 = random_scene(g, B2, A2, spread=24.0, centre=(100.0, 2.0))
    pr2[:, 0] = True
    w2 = st2[:, :, None, None, :2] + (torch.rand(B2, A2, 3, 2, 2, generator=g) - 0.5) * 30
    m2 = torch.rand(B2, A2, 3, 2, generator=g) < 0.7
    r = renderer_from_config(CV2RendererConfig())
    sim = make_sim(st2.clone(), sz2.clone(), pr2.clone(), road_mesh=crop.expand(B2), renderer=r)
    sim.waypoint_goals = WaypointGoal(w2.clone(), m2.clone())
    sim.waypoint_goals.state[1, 2] = 2                          # the window of two collections runs past the end for this agent
    d['r_state'], d['r_size'], d['r_present'], d['r_wp'], d['r_wmask'] = npy(st2), npy(sz2), npy(pr2), npy(w2), npy(m2)
    d['r_goal_state'] = npy(sim.waypoint_goals.state)
    d['r_road_verts'], d['r_road_faces'], d['r_road_vert_category'] = npy(crop.verts[0]), npy(crop.faces[0]).astype(np.int32), npy(crop.vert_category[0]).astype(np.uint8)
    for count in (1, 2):
        tris, cols, shape = render_record(cv2, sim, 96, 35.0, n_subsequent_waypoints=count)
        d[f'r_tris_{count}'], d[f'r_cols_{count}'] = tris, cols
        print('g11 render', count, tris.shape, 'waypoint-coloured calls', int((cols == np.array([139, 64, 0], np.uint8)).all(-1).sum()))
    np.savez_compressed(os.path.join(out, 'g11_waypoints.npz'), **d)
    print('g11 goal states', d[f'state_{T - 1}'].reshape(-1).tolist())


# --------------------------------------------------------------------------------------
# G12: traffic-light programmes (traffic_lights.py): the data files of the reference's own tests and of Town01, and a replay of the
#      reference's TrafficLightController on Town01's programmes
# --------------------------------------------------------------------------------------
def gen_traffic_lights(out):
    import random
    import shutil
    from torchdrivesim.traffic_lights import TrafficLightController, current_light_state_tensor_from_controller
    dst = os.path.join(out, 'traffic_lights')
    os.makedirs(os.path.join(dst, 'machines'), exist_ok=True)
    tsrc = os.path.join(REF, 'tests', 'resources')
    for name in sorted(os.listdir(os.path.join(tsrc, 'traffic_lights'))):                      # data files, not source
        if name.endswith('.json'):
            shutil.copyfile(os.path.join(tsrc, 'traffic_lights', name), os.path.join(dst, 'machines', name))
    shutil.copyfile(os.path.join(tsrc, 'traffic_lights_controller', 'intersection_controller.json'), os.path.join(dst, 'intersection_controller.json'))
    town = os.path.join(REF, 'torchdrivesim', 'resources', 'maps', 'carla_Town01')
    ctl_path = os.path.join(out, 'maps', 'carla_Town01', 'carla_Town01_traffic_light_controller.json')
    shutil.copyfile(os.path.join(town, 'carla_Town01_traffic_light_controller.json'), ctl_path)
    stop = json.load(open(os.path.join(town, 'carla_Town01_stoplines.json')))
    ids = [s['actor_id'] for s in stop if s['agent_type'] == 'traffic_light']
    random.seed(12)
    ctl = TrafficLightController.from_json(ctl_path)
    n = ctl.get_number_of_light_groups()
    rng = random.Random(1212)
    script, trace = [], []

    def snap():
        trace.append(dict(state_per_machine=list(ctl.state_per_machine), time_remaining=[float(t) for t in ctl.time_remaining],
                          names=ctl.current_state_with_name, tensor=current_light_state_tensor_from_controller(ctl, ids).tolist()))
    snap()                                                        # after the seeded reset
    ops = [('set_to', [[rng.randint(-1, 7), rng.choice([0.0, 0.5, 2.0, 3.0, 100.0])] for _ in range(n)])]
    ops += [('tick', dt) for dt in (0.1, 0.1, 0.1, 1.0, 0.0, 2.0, 0.5, 3.7, 25.0, 0.1, 61.3, 1e-3, 10.0)]
    ops += [('set_to', [[rng.randint(0, 5), float(rng.randint(0, 12))] for _ in range(n // 2)])]
    ops += [('tick', float(rng.choice([0.1, 0.1, 0.1, 1.0, 2.0, 5.0, 7.5, 30.0]))) for _ in range(60)]
    for op, arg in ops:
        getattr(ctl, op)(arg)
        script.append([op, arg])
        snap()
    with open(os.path.join(out, 'g12_traffic_lights.json'), 'w') as f:
        json.dump(dict(seed=12, ids=ids, script=script, trace=trace, to_json=json.loads(ctl.to_json())), f)


# --------------------------------------------------------------------------------------
# G13: collision_detection_with_discs with num_discs other than 5
# --------------------------------------------------------------------------------------
def gen_discs_n(out):
    from torchdrivesim.infractions import collision_detection_with_discs
    g = seeded(1313)
    n = 600
    xy1 = torch.rand(n, 2, generator=g) * 8 - 4
    xy2 = xy1 + torch.randn(n, 2, generator=g) * 2.5
    lw = lambda: torch.stack([torch.rand(n, generator=g) * 4 + 1.5, torch.rand(n, generator=g) * 2 + 0.8], -1)
    psi = lambda: (torch.rand(n, 1, generator=g) * 2 - 1) * math.pi
    b1, b2 = torch.cat([xy1, lw(), psi()], -1), torch.cat([xy2, lw(), psi()], -1)
    b1[:40, 2:4] = b1[:40, 2:4].flip(-1)                          # wid > len: the yaw + pi/2 branch
    d = dict(box1=npy(b1), box2=npy(b2))
    for k in (3, 7, 9, 25):
        d[f'discs_{k}'] = npy(collision_detection_with_discs(b1[None], b2[None], num_discs=k))[0]
    np.savez_compressed(os.path.join(out, 'g13_discs_n.npz'), **d)
    print('g13: nonzero fractions', {k: float((d[f'discs_{k}'] > 0).mean()) for k in (3, 7, 9, 25)})


# --------------------------------------------------------------------------------------
# G14: noisy perception (simulator.py:951-978): the observation model's background mesh, lane markers and traffic controls in the
#      rendered frame -- the call list of render_egocentric(noisy_perception=True) at the OpenCV boundary
# --------------------------------------------------------------------------------------
def gen_noisy_perception(out, cv2, town):
    from torchdrivesim.lanelet2 import LaneFeatures
    from torchdrivesim.observation_noise import MapObservationNoiseFromLog, StandardSensingObservationNoiseConfig
    from torchdrivesim.rendering import CV2RendererConfig, renderer_from_config
    from torchdrivesim.traffic_controls import TrafficLightControl, StopSignControl
    g = seeded(1414)
    centre = (100.0, 2.0)
    crop = crop_mesh(town, centre, 40.0)
    B, A, M = 2, 3, 6
    state, size, present = random_scene(g, B, A, spread=20.0, centre=centre)
    present[:, 0] = True
    # what the policy is shown: a smaller crop of the map shifted by 1.5 m, six lane markers (two masked), lights in other states
    noisy_bg = crop_mesh(town, centre, 25.0)
    noisy_bg = type(noisy_bg)(verts=noisy_bg.verts + torch.tensor([1.5, -0.75]), faces=noisy_bg.faces, categories=noisy_bg.categories,
                              colors=noisy_bg.colors, zs=noisy_bg.zs, vert_category=noisy_bg.vert_category, _cat_fill=noisy_bg._cat_fill).expand(B)
    markers = torch.cat([torch.tensor(centre) + (torch.rand(B, M, 2, generator=g) - 0.5) * 30, (torch.rand(B, M, 1, generator=g) * 2 - 1) * math.pi,
                         torch.rand(B, M, 1, generator=g) * 2 + 0.5], -1)
    mmask = torch.ones(B, M, dtype=torch.bool)
    mmask[0, 1] = mmask[1, 4] = False
    lights = torch.cat([torch.tensor(centre) + (torch.rand(B, 3, 2, generator=g) - 0.5) * 24, torch.tensor([1.0, 4.0]).expand(B, 3, 2),
                        (torch.rand(B, 3, 1, generator=g) * 2 - 1) * math.pi], -1)
    true_tc = {'traffic_light': TrafficLightControl(lights.clone())}
    true_tc['traffic_light'].set_state(torch.zeros(B, 3, dtype=torch.long))
    noisy_tc = {'traffic_light': TrafficLightControl(lights.clone() + torch.tensor([0.5, 0.5, 0, 0, 0])),
                'stop_sign': StopSignControl(torch.cat([torch.tensor(centre) + (torch.rand(B, 2, 2, generator=g) - 0.5) * 24,
                                                        torch.tensor([1.0, 3.5]).expand(B, 2, 2), torch.zeros(B, 2, 1)], -1))}
    noisy_tc['traffic_light'].set_state(torch.tensor([[2, 1, 0], [1, 2, 2]]))
    model = MapObservationNoiseFromLog(StandardSensingObservationNoiseConfig(),
                                       noisy_lane_features=[LaneFeatures(dense_lane_features=markers, dense_lane_features_mask=mmask)],
                                       noisy_background_mesh=[noisy_bg], noisy_traffic_controls=[noisy_tc])
    from torchdrivesim.simulator import Simulator, TorchDriveConfig
    from torchdrivesim.kinematic import KinematicBicycle
    km = KinematicBicycle()
    km.set_params(lr=torch.full((B, A), 1.5))
    km.set_state(state.clone())
    sim = Simulator(crop.expand(B), km, size.clone(), present.clone(), TorchDriveConfig(), renderer=renderer_from_config(CV2RendererConfig()),
                    traffic_controls=true_tc, observation_noise_model=model,
                    lane_features=LaneFeatures(dense_lane_features=markers[:, :2] + torch.tensor([2.0, 2.0, 0.3, 0.0]), dense_lane_features_mask=mmask[:, :2] | True))
    d = dict(state=npy(state), size=npy(size), present=npy(present), markers=npy(markers), markers_mask=npy(mmask), lights=npy(lights),
             noisy_lights=npy(noisy_tc['traffic_light'].pos), noisy_light_state=npy(noisy_tc['traffic_light'].state),
             noisy_stop=npy(noisy_tc['stop_sign'].pos),
             road_verts=npy(crop.verts[0]), road_faces=npy(crop.faces[0]).astype(np.int32), road_vert_category=npy(crop.vert_category[0]).astype(np.uint8),
             noisy_verts=npy(noisy_bg.verts[0]), noisy_faces=npy(noisy_bg.faces[0]).astype(np.int32),
             noisy_vert_category=npy(noisy_bg.vert_category[0]).astype(np.uint8))
    for tag, kw in (('noisy', dict(noisy_perception=True)), ('plain', dict())):
        tris, cols, shape = render_record(cv2, sim, 96, 35.0, **kw)
        d[f'tris_{tag}'], d[f'cols_{tag}'] = tris, cols
        print('g14', tag, tris.shape, 'marker-coloured calls', int((cols == np.array([255, 0, 255], np.uint8)).all(-1).sum()))
    sim.internal_time = 1                                       # the log is exhausted: noisy perception shows the truth
    tris, cols, _ = render_record(cv2, sim, 96, 35.0, noisy_perception=True)
    d['tris_after'], d['cols_after'] = tris, cols
    np.savez_compressed(os.path.join(out, 'g14_noisy_perception.npz'), **d)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    ap.add_argument('--only', default=None, help='regenerate one group only (e.g. waypoints)')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    cv2 = install_stubs()
    sys.path.insert(0, REF)
    torch.set_num_threads(1)
    import torchdrivesim  # noqa: F401  (the reference)
    assert os.path.realpath(torchdrivesim.__path__[0]).startswith(os.path.realpath(REF))
    if args.only == 'noisy':
        gen_noisy_perception(args.out, cv2, load_town01())
        return
    if args.only == 'discs_n':
        gen_discs_n(args.out)
        return
    if args.only == 'traffic_lights':
        gen_traffic_lights(args.out)
        return
    if args.only == 'displacement':
        gen_kinematic_displacement(args.out)
        return
    if args.only == 'untrimmed':
        gen_preraster_untrimmed(args.out, cv2, load_town01())
        return
    if args.only == 'waypoints':
        gen_waypoints(args.out, cv2, load_town01())
        return
    gen_kinematic(args.out)
    gen_collision(args.out)
    gen_scene_collision(args.out)
    town = gen_town01(args.out)
    gen_offroad(args.out, town)
    gen_mesh_and_preraster(args.out, cv2, town)
    gen_grads(args.out, town)
    gen_traffic(args.out)
    gen_observation(args.out)
    gen_waypoints(args.out, cv2, town)
    gen_traffic_lights(args.out)
    gen_discs_n(args.out)
    gen_noisy_perception(args.out, cv2, town)
    gen_kinematic_displacement(args.out)
    gen_preraster_untrimmed(args.out, cv2, town)
    with open(os.path.join(args.out, 'PROVENANCE.txt'), 'w') as f:
        f.write(f'generated by tools/gen_golden.py from the reference at {REF} (torchdrivesim {torchdrivesim.__version__}), '
                f'torch {torch.__version__} CPU, numpy {np.__version__}\n')


if __name__ == '__main__':
    main()
