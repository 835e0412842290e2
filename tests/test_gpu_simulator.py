"""End-to-end parity through the reference-shaped API (Simulator / KinematicBicycle / HipRenderer) on an MI355X."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def town_mesh(B, crop=None):
    from torchdrivesim_amd.mesh import BirdviewMesh
    t = load_golden('town01_mesh.npz')
    return BirdviewMesh(verts=torch.from_numpy(t['verts'])[None], faces=torch.from_numpy(t['faces'].astype(np.int64))[None],
                        categories=[str(c) for c in t['categories']], colors={}, zs={},
                        vert_category=torch.from_numpy(t['vert_category'].astype(np.int64))[None]).expand(B).to(DEV), t


def make_sim(state, size, present, road, metric='iou', npc=None, lr=None, **kw):
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig, CollisionMetric, NPCController
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    km = KinematicBicycle()
    km.set_params(lr=d(lr) if lr is not None else torch.full(state.shape[:2], 1.5, device=DEV))
    km.set_state(d(state))
    ctrl = None
    if npc is not None:
        ctrl = NPCController(npc_size=d(npc['size']), npc_state=d(npc['state']), npc_present_mask=d(npc['present']))
    cfg = TorchDriveConfig(collision_metric=CollisionMetric(metric), renderer=HipRendererConfig())
    return Simulator(road, km, d(size), d(present), cfg, npc_controller=ctrl, **kw)


def test_smoke_entry_point():
    import __graft_entry__
    __graft_entry__.smoke()


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_compute_collision_matches_reference(metric):
    from torchdrivesim_amd.mesh import BirdviewMesh
    g = load_golden('g2_scene_collision.npz')
    B = g['state'].shape[0]
    road = BirdviewMesh.empty(batch_size=B).to(DEV)
    sim = make_sim(g['state'], g['size'], g['present'], road, metric)
    out = sim.compute_collision().cpu().numpy()
    np.testing.assert_array_equal(out > 0, g['coll_' + metric] > 0)
    np.testing.assert_allclose(out, g['coll_' + metric], atol=2e-6, rtol=0)
    sim = make_sim(g['state'], g['size'], g['present'], road, metric, npc=dict(state=g['npc_state'], size=g['npc_size'], present=g['npc_present']))
    out = sim.compute_collision().cpu().numpy()
    assert out.shape == (B, 8)
    np.testing.assert_array_equal(out > 0, g['coll_npc_' + metric] > 0)
    np.testing.assert_allclose(out, g['coll_npc_' + metric], atol=2e-6, rtol=0)


def test_compute_offroad_matches_reference():
    from torchdrivesim_amd.mesh import BaseMesh
    g = load_golden('g3_offroad.npz')
    road = BaseMesh(verts=torch.from_numpy(g['b_verts']), faces=torch.from_numpy(g['b_faces'].astype(np.int64))).to(DEV)   # two different scenes
    sim = make_sim(g['b_state'], g['b_lenwid'], g['c_present'], road)
    out = sim.compute_offroad().cpu().numpy()
    np.testing.assert_allclose(out, g['c_sim_offroad'], rtol=2e-6, atol=1e-6)
    np.testing.assert_array_equal(out > 0, g['c_sim_offroad'] > 0)
    from torchdrivesim_amd.infractions import offroad_infraction_loss
    d = lambda a: torch.from_numpy(a).to(DEV)
    np.testing.assert_allclose(offroad_infraction_loss(d(g['b_state']), d(g['b_lenwid']), road, threshold=0.0).cpu().numpy(), g['b_off_t0'],
                               rtol=2e-6, atol=1e-6)


def test_render_egocentric_matches_reference_pipeline(oracle):
    """G5 case `town01_128`: images through Simulator.render_egocentric vs the oracle fed with the golden inputs; the
    oracle's pre-raster stage is pinned to the reference's call lists in test_oracle_golden.py."""
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g45_mesh_preraster.npz')
    meta = {m['name']: m for m in json.loads(str(g['g5_meta']))}
    for name in ('town01_128', 'town01_64_lh'):
        m = meta[name]
        st, sz, pr = g[f'g5_{name}_state'], g[f'g5_{name}_size'], g[f'g5_{name}_present']
        B, A = st.shape[:2]
        road, t = town_mesh(B)
        sim = make_sim(st, sz, pr, road)
        sim.cfg.left_handed_coordinates = m['left_handed']
        img = sim.render_egocentric(res=Resolution(m['res'], m['res']), fov=m['fov'])
        assert tuple(img.shape) == tuple(m['out_shape'])
        sv, sa, sf = oracle.static_mesh_arrays(t['verts'], t['faces'], t['vert_category'], [str(c) for c in t['categories']])
        s = sim.get_state()
        sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1).cpu().numpy()
        mask = np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A)))
        ref = oracle.render_scenes(st, sz, mask, st[..., :2].copy(), sc, sv, sa, sf, m['fov'], m['res'], agent_sc=sc)
        np.testing.assert_array_equal(img.cpu().numpy(), ref)
        # the generic BirdviewRenderer.render_frame dataflow (explicit per-camera mesh) gives the same pixels
        if name == 'town01_64_lh':
            gen = sim.birdview_mesh_generator
            rgb = gen.generate(A, agent_state=s[:, None].expand(-1, A, -1, -1), present_mask=torch.from_numpy(mask).to(DEV))
            cam_sc = torch.from_numpy(sc).to(DEV)
            img2 = sim.renderer.render_frame(rgb, s[..., :2], cam_sc, res=Resolution(m['res'], m['res']), fov=m['fov'])
            np.testing.assert_array_equal(img2.reshape(img.shape).cpu().numpy(), ref)


def test_step_render_loop_and_single_agent_rendering():
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g45_mesh_preraster.npz')
    st, sz, pr = g['g5_town01_128_state'], g['g5_town01_128_size'], g['g5_town01_128_present']
    road, _ = town_mesh(st.shape[0])
    sim = make_sim(st, sz, pr, road)
    before = sim.get_state().clone()
    act = torch.zeros(st.shape[0], st.shape[1], 2, device=DEV)
    act[0, 0, 0] = 1.0
    sim.step(act)
    moved = (sim.get_state() - before).abs().sum(-1) > 0
    assert bool(moved[0, 0])                                   # test_simulator.py:134-138: the actuated agent moves
    img = sim.render_egocentric(res=Resolution(64, 64))
    assert img.shape == (2, 8, 3, 64, 64) and img.dtype == torch.float32
    sim.cfg.single_agent_rendering = True
    solo = sim.render_egocentric(res=Resolution(64, 64))
    veh = torch.tensor([32.0, 74.0, 135.0], device=DEV).view(1, 1, 3, 1, 1)
    assert ((solo == veh).all(2).flatten(2).sum(-1) <= (img == veh).all(2).flatten(2).sum(-1)).all()


def test_custom_agent_colors_fused_equals_reference_dataflow(oracle):
    """custom_agent_colors (simulator.py:979-984 -> mesh.py:1092-1099): the fused path (per-camera actor keys) paints the same
    pixels as the reference's dataflow -- explicit per-camera RGB mesh from generate(), then render_frame -- which is checked
    against the oracle's render_rgb_mesh."""
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g45_mesh_preraster.npz')
    st, sz, pr = g['g5_town01_128_state'], g['g5_town01_128_size'], g['g5_town01_128_present']
    B, A = st.shape[:2]
    road, t = town_mesh(B)
    sim = make_sim(st, sz, pr, road)
    gen = np.random.default_rng(4)
    palette = np.array([[255, 0, 0], [0, 255, 0], [32, 74, 135], [250, 250, 10], [1, 2, 3]], np.float32) / 255.0
    cc = torch.from_numpy(palette[gen.integers(0, len(palette), (B, A, A))]).to(DEV)
    res = Resolution(96, 96)
    img = sim.render_egocentric(res=res, fov=35.0, custom_agent_colors=cc)
    plain = sim.render_egocentric(res=res, fov=35.0)
    assert (img != plain).any()
    s = sim.get_state()
    mask = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(pr[:, None, :], (B, A, A)))).to(DEV)
    rgb = sim.birdview_mesh_generator.generate(A, agent_state=s[:, None].expand(-1, A, -1, -1), present_mask=mask, custom_agent_colors=cc)
    cam_sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1)
    img2 = sim.renderer.render_frame(rgb, s[..., :2], cam_sc, res=res, fov=35.0).reshape(img.shape)
    np.testing.assert_array_equal(img.cpu().numpy(), img2.cpu().numpy())
    ref = oracle.render_rgb_mesh(rgb.verts.cpu().numpy(), rgb.attrs.cpu().numpy(), rgb.faces.cpu().numpy().astype(np.int32),
                                 s[..., :2].reshape(-1, 2).cpu().numpy(), cam_sc.reshape(-1, 2).cpu().numpy(), 2.0 / 35.0, 96)     # n x H x W x 3
    np.testing.assert_array_equal(img.reshape((-1,) + tuple(img.shape[2:])).cpu().numpy(), np.transpose(ref, (0, 3, 1, 2)))


def test_traffic_light_violations_match_reference():
    """TrafficLightControl.compute_violation on K2a's box-intersection kernel and Simulator.compute_traffic_lights_violations
    against the reference's outputs (g8_traffic.npz), incl. stepping the replayed light states"""
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.traffic_controls import TrafficLightControl
    g = load_golden('g8_traffic.npz')
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    ctl = TrafficLightControl(d(g['pos']), replay_states=d(g['replay']), mask=d(g['mask']))
    boxes = d(g['boxes'])
    for t in range(7):
        ctl.step(t)
        np.testing.assert_array_equal(ctl.compute_violation(boxes).cpu().numpy(), g[f'violation_{t}'])
    # through the Simulator: state [x, y, psi, v] + sizes; stepping with zero speed keeps the boxes where they are
    B, A = boxes.shape[:2]
    state = np.concatenate([g['boxes'][..., :2], g['boxes'][..., 4:5], np.zeros((B, A, 1), np.float32)], -1)
    sim = make_sim(state, g['boxes'][..., 2:4], np.ones((B, A), bool), BirdviewMesh.empty(batch_size=B).to(DEV))
    sim.traffic_controls = {'traffic_light': TrafficLightControl(d(g['pos']), replay_states=d(g['replay']), mask=d(g['mask']))}
    np.testing.assert_array_equal(sim.compute_traffic_lights_violations().cpu().numpy() > 0, g['violation_0'])
    for t in range(1, 4):
        sim.step(torch.zeros(B, A, 2, device=DEV))
        assert sim.internal_time == t
        np.testing.assert_array_equal(sim.compute_traffic_lights_violations().cpu().numpy() > 0, g[f'violation_{t}'])
    half = sim.select_batch_elements(torch.tensor([1]), in_place=False)
    assert half.get_traffic_controls()['traffic_light'].pos.shape[0] == 1


def test_traffic_controls_are_rendered_like_the_reference_mesh(oracle):
    """G9: stop lines as quads, lights coloured by their state.  The fused path (quads ride along with the actors) paints the pixels
    of the RGB mesh the REFERENCE generated for the same scene (tests/golden/g9_traffic_mesh.npz) as drawn by the oracle, and so does
    the generic dataflow of this framework (generate() -> render_frame)."""
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.traffic_controls import StopSignControl, TrafficLightControl, YieldControl
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g9_traffic_mesh.npz')
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    B, A = g['state'].shape[:2]
    road = BirdviewMesh(verts=d(g['bg_verts'])[None], faces=d(g['bg_faces'].astype(np.int64))[None], categories=['right_lane', 'left_lane', 'road'],
                        colors={}, zs={}, vert_category=d(g['bg_vert_category'].astype(np.int64))[None]).expand(B)
    tl = TrafficLightControl(d(g['tl_pos']), mask=d(g['tl_mask']))
    tl.set_state(d(g['tl_state']))
    controls = dict(stop_sign=StopSignControl(d(g['ss_pos'])), traffic_light=tl, yield_sign=YieldControl(d(g['ys_pos'])))
    sim = make_sim(g['state'], g['size'], g['present'], road, traffic_controls=controls)
    s = sim.get_state()
    cam_sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1)
    for res, fov in ((128, 35.0), (96, 60.0)):
        ref = oracle.render_rgb_mesh(g['rgb_verts'], g['rgb_attrs'], g['rgb_faces'], s[..., :2].reshape(-1, 2).cpu().numpy(),
                                     cam_sc.reshape(-1, 2).cpu().numpy(), 2.0 / fov, res)                      # n x H x W x 3
        ref = np.transpose(ref, (0, 3, 1, 2)).reshape(B, A, 3, res, res)
        img = sim.render_egocentric(res=Resolution(res, res), fov=fov)
        np.testing.assert_array_equal(img.cpu().numpy(), ref)
        assert any((ref[:, :, 0] == c[0]).any() for c in ((224, 53, 49), (240, 189, 39), (81, 179, 100)))       # a light is in view
        mask = d(np.ascontiguousarray(np.broadcast_to(g['present'][:, None, :], (B, A, A))))
        rgb = sim.birdview_mesh_generator.generate(A, agent_state=s[:, None].expand(-1, A, -1, -1), present_mask=mask,
                                                   traffic_lights=controls['traffic_light'].extend(A, in_place=False))
        np.testing.assert_array_equal(rgb.faces.cpu().numpy(), g['rgb_faces'])
        np.testing.assert_allclose(rgb.verts.cpu().numpy(), g['rgb_verts'], atol=2e-5, rtol=0)     # device sin / cos differ from the CPU's by an ulp
        np.testing.assert_allclose(rgb.attrs.cpu().numpy(), g['rgb_attrs'], atol=1e-6, rtol=0)     # colour / 255 on the device
        img2 = sim.renderer.render_frame(rgb, s[..., :2], cam_sc, res=Resolution(res, res), fov=fov).reshape(img.shape)
        ref2 = oracle.render_rgb_mesh(rgb.verts.cpu().numpy(), rgb.attrs.cpu().numpy(), rgb.faces.cpu().numpy().astype(np.int32),
                                      s[..., :2].reshape(-1, 2).cpu().numpy(), cam_sc.reshape(-1, 2).cpu().numpy(), 2.0 / fov, res)
        np.testing.assert_array_equal(img2.cpu().numpy(), np.transpose(ref2, (0, 3, 1, 2)).reshape(img.shape))
        assert (img2.cpu().numpy() != ref).mean() < 1e-3
    # a light that changes state changes colour
    before = sim.render_egocentric(res=Resolution(96, 96), fov=60.0)
    tl.set_state((tl.state + 1) % 3)
    assert (sim.render_egocentric(res=Resolution(96, 96), fov=60.0) != before).any()


def test_standard_sensing_occlusion_matches_reference():
    """tds_occlusion_mask_f32 through StandardSensingObservationNoise / Simulator.get_noisy_present_mask against the reference's
    masks (g10_observation.npz), and the shape / statistics of the noisy states"""
    from test_observation import sim_of
    from torchdrivesim_amd.observation_noise import StandardSensingObservationNoise
    g = load_golden('g10_observation.npz')
    for tag in 'abc':
        sim = sim_of(g, tag, device=DEV, noise=StandardSensingObservationNoise())
        np.testing.assert_array_equal(sim.get_noisy_present_mask().cpu().numpy(), g[f'{tag}_mask'])
        ns = sim.get_noisy_state()
        assert ns.shape == g[f'{tag}_noisy_absolute'].shape[:3] + (4,)
        A = sim.agent_count
        idx = torch.arange(A, device=DEV)
        assert torch.equal(ns[:, idx, idx], sim.get_state())                     # an agent perceives itself exactly (distance 0)
        assert sim.get_noisy_all_agents_relative().shape[2] == sim.agent_count + sim.npc_count - 1


def test_scenes_with_different_maps(oracle):
    """a batch whose scenes have DIFFERENT road meshes (collated with padding, mesh.py:69): one device map per scene, for rendering
    and for the off-road query alike"""
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g45_mesh_preraster.npz')
    t = load_golden('town01_mesh.npz')
    cats = [str(c) for c in t['categories']]
    crop = BirdviewMesh(verts=torch.from_numpy(g['g4_bg_verts'])[None], faces=torch.from_numpy(g['g4_bg_faces'].astype(np.int64))[None], categories=cats,
                        colors={}, zs={}, vert_category=torch.from_numpy(g['g4_bg_vert_category'].astype(np.int64))[None])
    keep = (t['verts'][t['faces']][..., 0].max(1) < 130) & (t['verts'][t['faces']][..., 1].max(1) < 40)
    part = BirdviewMesh(verts=torch.from_numpy(t['verts'])[None], faces=torch.from_numpy(t['faces'][keep].astype(np.int64))[None], categories=cats,
                        colors={}, zs={}, vert_category=torch.from_numpy(t['vert_category'].astype(np.int64))[None])
    road = BirdviewMesh.collate([crop, part]).to(DEV)
    gen = np.random.default_rng(8)
    B, A = 2, 5
    state = np.concatenate([np.array([100.0, 2.0]) + gen.uniform(-8, 8, (B, A, 2)), gen.uniform(-np.pi, np.pi, (B, A, 1)), np.zeros((B, A, 1))], -1).astype(np.float32)
    size = np.tile(np.array([4.5, 2.0], np.float32), (B, A, 1))
    sim = make_sim(state, size, np.ones((B, A), bool), road)
    img = sim.render_egocentric(res=Resolution(128, 128), fov=40.0).cpu().numpy()
    s = sim.get_state()
    sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1).cpu().numpy()
    off = sim.compute_offroad().cpu().numpy()
    for b, m in enumerate((crop, part)):
        v, f, vc = m.verts[0].numpy(), m.faces[0].numpy().astype(np.int32), m.vert_category[0].numpy()
        sv, sa, sf = oracle.static_mesh_arrays(v, f, vc, cats)
        ref = oracle.render_scenes(state[b:b + 1], size[b:b + 1], np.ones((1, A, A), bool), state[b:b + 1, :, :2].copy(), sc[b:b + 1], sv, sa, sf, 40.0, 128,
                                   agent_sc=sc[b:b + 1])
        np.testing.assert_array_equal(img[b:b + 1], ref)
        ref_off = oracle.offroad(state[b:b + 1], size[b:b + 1], v, f, threshold=0.5, present=np.ones((1, A), bool), sc=sc[b:b + 1])
        np.testing.assert_allclose(off[b:b + 1], ref_off, rtol=1e-5, atol=1e-6)
    assert (img[0] != img[1]).any()
    # off-road gradients through the map set equal those through one map per scene
    from torchdrivesim_amd import _ops
    st = sim.get_state().detach().clone()
    st[..., :2] += torch.tensor([30.0, 25.0], device=DEV)             # off the road: non-zero losses
    sz = sim.get_agent_size()
    s1 = st.clone().requires_grad_(True)
    sim.kinematic_model.set_state(s1)
    loss = sim.compute_offroad()
    assert (loss > 0).any()
    loss.sum().backward()
    for b, m in enumerate((crop, part)):
        single = _ops.StaticMap(m.verts[0], m.faces[0], device=DEV)
        s2 = st[b:b + 1].clone().requires_grad_(True)
        l2 = _ops.offroad(single, s2, sz[b:b + 1], threshold=0.5, present=torch.ones(1, A, dtype=torch.bool, device=DEV))
        np.testing.assert_allclose(loss[b:b + 1].detach().cpu().numpy(), l2.detach().cpu().numpy(), rtol=1e-6)
        l2.sum().backward()
        np.testing.assert_allclose(s1.grad[b:b + 1].cpu().numpy(), s2.grad.cpu().numpy(), rtol=1e-5, atol=1e-6)


def two_towns(order):
    """Town01 and Town02 collated (mesh.py:232-245: padded to the larger one) and laid out as the batch `order` (0 = Town01, 1 = Town02) on the device"""
    from torchdrivesim_amd.mesh import BirdviewMesh
    towns = []
    for name in ('town01_mesh.npz', 'town02_mesh.npz'):
        t = load_golden(name)
        towns.append(BirdviewMesh(verts=torch.from_numpy(t['verts'])[None], faces=torch.from_numpy(t['faces'].astype(np.int64))[None],
                                  categories=[str(c) for c in t['categories']], colors={}, zs={},
                                  vert_category=torch.from_numpy(t['vert_category'].astype(np.int64))[None]))
    pair = BirdviewMesh.collate(towns)
    return pair.to(DEV)[list(order)], pair


def test_mixed_maps_batch_builds_one_map_per_distinct_mesh(oracle):
    """B = 64 scenes collated from Town01 / Town02 in mixed order build exactly TWO rendering maps (and two geometry-only
    maps for the off-road query) -- counted through _ops.map_creations, i.e. tds_map_create calls --, pixels and off-road losses equal the
    oracle's for scenes of both towns, and copy / select_batch_elements / extend / shard_simulator create none."""
    from torchdrivesim_amd import _ops, parallel
    from torchdrivesim_amd.utils import Resolution
    B, A, res, fov = 64, 6, 64, 35.0
    gen = np.random.default_rng(21)
    order = gen.integers(0, 2, B)
    order[:4] = [1, 0, 0, 1]
    road, pair = two_towns(order)
    cats = pair.categories
    state = np.zeros((B, A, 4), np.float32)
    for b in range(B):
        v = pair.verts[order[b]].numpy()[pair.vert_category[order[b]].numpy() == cats.index('road')]
        v = v[np.abs(v).sum(1) > 0]                          # (padding rows)
        state[b, :, :2] = v[gen.integers(0, len(v))] + gen.normal(0, 6.0, (A, 2))
        state[b, :, 2] = gen.uniform(-np.pi, np.pi, A)
    size = np.tile(np.array([4.5, 2.0], np.float32), (B, A, 1))
    present = np.ones((B, A), bool)
    _ops.map_cache.clear()
    sim = make_sim(state, size, present, road)
    n0 = _ops.map_creations
    img = sim.render_egocentric(res=Resolution(res, res), fov=fov)
    assert _ops.map_creations == n0 + 2, 'one rendering map per DISTINCT mesh'
    smap = sim._scene()['maps'][0][0]
    assert isinstance(smap, _ops.StaticMapSet) and len(smap.maps) == 2
    np.testing.assert_array_equal(smap.scene_map.cpu().numpy(), order if order[0] == 0 else 1 - order)       # groups are numbered by first occurrence
    off = sim.compute_offroad()
    assert _ops.map_creations == n0 + 4, 'one geometry-only map per distinct mesh for the off-road query'
    img, off = img.cpu().numpy(), off.cpu().numpy()
    s = sim.get_state()
    sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1).cpu().numpy()
    for b in (0, 1, 2, 3, B - 1):
        m = pair[[int(order[b])]]
        v, f, vc = m.verts[0].numpy(), m.faces[0].numpy().astype(np.int32), m.vert_category[0].numpy()
        sv, sa, sf = oracle.static_mesh_arrays(v, f, vc, cats)
        ref = oracle.render_scenes(state[b:b + 1], size[b:b + 1], np.ones((1, A, A), bool), state[b:b + 1, :, :2].copy(), sc[b:b + 1], sv, sa, sf, fov, res,
                                   agent_sc=sc[b:b + 1])
        np.testing.assert_array_equal(img[b:b + 1], ref)
        assert ref.std() > 0
        ref_off = oracle.offroad(state[b:b + 1], size[b:b + 1], v, f, threshold=0.5, present=np.ones((1, A), bool), sc=sc[b:b + 1])
        np.testing.assert_allclose(off[b:b + 1], ref_off, rtol=1e-5, atol=1e-6)
    # batch operations find the handles in the cache
    n1 = _ops.map_creations
    idx = [5, 1, 2, 40, 41, 0]
    sub = sim.select_batch_elements(idx, in_place=False)
    np.testing.assert_array_equal(sub.render_egocentric(res=Resolution(res, res), fov=fov).cpu().numpy(), img[idx])
    np.testing.assert_array_equal(sub.compute_offroad().cpu().numpy(), off[idx])
    only1 = sim.select_batch_elements([b for b in range(B) if order[b] == 1][:3], in_place=False)          # a sub-batch on ONE town: a plain map, from the cache
    sub_img = only1.render_egocentric(res=Resolution(res, res), fov=fov).cpu().numpy()
    np.testing.assert_array_equal(sub_img, img[[b for b in range(B) if order[b] == 1][:3]])
    twice = sub.extend(2, in_place=False)
    np.testing.assert_array_equal(twice.render_egocentric(res=Resolution(res, res), fov=fov).cpu().numpy(), np.repeat(img[idx], 2, axis=0))
    cp = sim.copy()
    np.testing.assert_array_equal(cp.render_egocentric(res=Resolution(res, res), fov=fov).cpu().numpy(), img)
    shard = parallel.shard_simulator(sim, 1, 2)
    np.testing.assert_array_equal(shard.render_egocentric(res=Resolution(res, res), fov=fov).cpu().numpy(), img[B // 2:])
    np.testing.assert_array_equal(shard.compute_offroad().cpu().numpy(), off[B // 2:])
    assert _ops.map_creations == n1, 'batch operations must reuse the device maps'


def test_group_rows_confirms_hash_groups_exactly():
    """_ops.group_rows: groups numbered by first occurrence, expanded batches are one group, one differing word splits a group"""
    from torchdrivesim_amd import _ops
    g = torch.Generator().manual_seed(3)
    rows = torch.randn(3, 1000, 3, generator=g)
    order = [2, 0, 2, 1, 0, 2]
    a = rows[order].to(DEV)
    faces = torch.randint(0, 1000, (3, 777, 3), generator=g)[order].to(DEV)
    sm, reps, hashes = _ops.group_rows([a, faces])
    assert sm.tolist() == [0, 1, 0, 2, 1, 0] and reps == [0, 1, 3] and len(set(hashes)) == 3
    b = a.clone()
    b[5, 999, 2] += 1.0                                       # the last word of the last row
    sm, reps, _ = _ops.group_rows([b, faces])
    assert sm.tolist() == [0, 1, 0, 2, 1, 3] and reps == [0, 1, 3, 5]
    c = a.clone()
    c[2, 0, 0], c[2, 0, 1] = a[2, 0, 1].clone(), a[2, 0, 0].clone()   # two words swapped: the hash depends on the position
    assert _ops.group_rows([c, faces])[0].tolist() == [0, 1, 2, 3, 1, 0]
    sm, reps, _ = _ops.group_rows([a[:1].expand(5, -1, -1), faces[:1].expand(5, -1, -1)])
    assert sm.tolist() == [0] * 5 and reps == [0]
    # rows that only differ in a tensor that is not expanded
    sm, _, _ = _ops.group_rows([a[:1].expand(6, -1, -1), faces])
    assert sm.tolist() == [0, 1, 0, 2, 1, 0]
    # the exact confirmation: identical hashes forced by hashing a constant column, different bytes elsewhere -> (collision path) every row its own group
    saved = _ops.row_hashes
    try:
        _ops.row_hashes = lambda tensors: torch.zeros(tensors[0].shape[0], 2, dtype=torch.int64, device=tensors[0].device)
        sm, reps, _ = _ops.group_rows([a])
        assert sm.tolist() == [0, 1, 0, 2, 3, 0] and reps == [0, 1, 3, 4]
        for i in range(6):
            for j in range(6):
                if sm[i] == sm[j]:
                    assert order[i] == order[j]
    finally:
        _ops.row_hashes = saved


def test_waypoint_goals_are_drawn_by_the_fused_path(oracle):
    """render_egocentric with waypoint goals (simulator.py:1013-1029 -> mesh.py:1120-1145): the fused path draws the discs as per-camera
    triangles; same pixels as the reference's dataflow (explicit mesh from generate(), render_frame) and as the oracle fed with that mesh,
    whose call list is pinned to the reference's in tests/test_waypoints.py.  Includes masked waypoints (a dot at the centre of the
    camera's first waypoint) and a window of collections that runs past the end."""
    from test_waypoints import sim_with_goals
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g11_waypoints.npz')
    sim = sim_with_goals(g, device=DEV)
    s = sim.get_state()
    B, A = s.shape[:2]
    mask = sim.get_present_mask()[:, None].expand(B, A, A)
    cam_sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1)
    res = Resolution(96, 96)
    sim.waypoint_goals, goals = None, sim.waypoint_goals
    plain = sim.render_egocentric(res=res, fov=35.0)
    sim.waypoint_goals = goals
    imgs = {}
    for count, res, fov in ((1, Resolution(96, 96), 35.0), (2, Resolution(96, 96), 35.0), (3, Resolution(320, 320), 60.0)):   # 320: two strips per camera
        if res.width != 96:
            sim.waypoint_goals = None
            plain = sim.render_egocentric(res=res, fov=fov)
            sim.waypoint_goals = goals
        img = imgs[count] = sim.render_egocentric(res=res, fov=fov, n_subsequent_waypoints=count)
        assert (img != plain).any()
        rgb = sim.birdview_mesh_generator.generate(A, agent_state=s[:, None].expand(-1, A, -1, -1), present_mask=mask,
                                                   waypoints=sim.get_waypoints(count), waypoints_rendering_mask=sim.get_waypoints_mask(count))
        img2 = sim.renderer.render_frame(rgb, s[..., :2], cam_sc, res=res, fov=fov).reshape(img.shape)
        ref = oracle.render_rgb_mesh(rgb.verts.cpu().numpy(), rgb.attrs.cpu().numpy(), rgb.faces.cpu().numpy().astype(np.int32),
                                     s[..., :2].reshape(-1, 2).cpu().numpy(), cam_sc.reshape(-1, 2).cpu().numpy(), 2.0 / fov, res.width)
        ref = np.transpose(ref, (0, 3, 1, 2)).reshape(img.shape)
        np.testing.assert_array_equal(img2.cpu().numpy(), ref)
        np.testing.assert_array_equal(img.cpu().numpy(), ref)
        wp_col = torch.tensor([139.0, 64.0, 0.0], device=DEV).view(1, 1, 3, 1, 1)
        assert int((img == wp_col).all(2).sum()) > 50
    # explicit waypoints through render(), without a rendering mask, on the packed-key kernels too (no key table -> no bit planes)
    from torchdrivesim_amd import _ops
    res = Resolution(96, 96)
    sim.waypoint_goals = None
    plain = sim.render_egocentric(res=res, fov=35.0)
    sim.waypoint_goals = goals
    wp = sim.get_waypoints(1)
    a = sim.render(s[..., :2], s[..., 2:3], res=res, fov=35.0, waypoints=wp)
    _ops.use_bitplanes = False
    try:
        b = sim.render(s[..., :2], s[..., 2:3], res=res, fov=35.0, waypoints=wp)
    finally:
        _ops.use_bitplanes = True
    assert torch.equal(a, b) and (a != plain).any()
    # the differentiable wrapper carries the discs along
    st = s.detach().clone().requires_grad_(True)
    sim.kinematic_model.set_state(st)
    c = sim.render_egocentric(res=res, fov=35.0)
    assert c.requires_grad and torch.equal(c.detach(), imgs[1])
    c.sum().backward()
    assert torch.isfinite(st.grad).all()


# ------------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json's configurations 2, 3 and 5 at their FULL size (B = 256 x A = 64, 256 x 256), through size-independent properties plus the
# slice of the batch the oracle can afford (these sizes had only been run by the benchmark scripts)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def full_size():
    import bench
    from oracle import lanelet_oracle
    from torchdrivesim_amd import lanelet2
    import os
    osm = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'carla_Town01.osm.gz')
    lanes = lanelet2.load_lanelet_map(osm, origin=(0.0, 0.0))
    sim, actions, host = bench.build_simulator(256, 64, torch.device(DEV), seed=77, lanelet_map=lanes)
    sim.step(actions[0])
    return sim, actions, host, lanelet_oracle.load_osm(osm, origin=(0.0, 0.0))


def test_config2_full_size_render_and_collision(full_size, oracle):
    from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config
    from torchdrivesim_amd.utils import Resolution
    sim, actions, host, _ = full_size
    state0, size, present, act, verts, faces, vcat, cats = host
    res = Resolution(256, 256)
    img = sim.render_egocentric(res=res, fov=35.0)
    col = sim.compute_collision()
    assert img.shape == (256, 64, 3, 256, 256) and img.dtype == torch.float32 and col.shape == (256, 64)
    # scenes are independent: any sub-batch renders / collides to the same bits as inside the full batch
    pick = [3, 100, 255]
    sub = sim.select_batch_elements(torch.tensor(pick), in_place=False)
    assert torch.equal(sub.render_egocentric(res=res, fov=35.0), img[pick])
    assert torch.equal(sub.compute_collision(), col[pick])
    # the uint8 mode shows the same values
    u8 = sim.copy()
    u8.renderer = renderer_from_config(HipRendererConfig(out_dtype='uint8'), res=res, fov=35.0)
    u8._scene_cache = None
    assert torch.equal(u8.render_egocentric(res=res, fov=35.0)[pick].float(), img[pick])
    # the oracle on what it can afford: one scene of images, eight scenes of collisions
    s1 = sim.get_state().cpu().numpy()
    sc = torch.stack([torch.sin(sim.get_state()[..., 2]), torch.cos(sim.get_state()[..., 2])], -1).cpu().numpy()
    sv, sa, sf = oracle.static_mesh_arrays(verts, faces, vcat, cats)
    k = 1
    mask = np.ascontiguousarray(np.broadcast_to(present[:k, None, :], (k, 64, 64)))
    ref = oracle.render_scenes(s1[:k], size[:k], mask, s1[:k, :, :2].copy(), sc[:k], sv, sa, sf, 35.0, 256, agent_sc=sc[:k])
    np.testing.assert_array_equal(img[:k].cpu().numpy(), ref)
    boxes = np.concatenate([s1[..., :2], size, s1[..., 2:3]], -1)
    np.testing.assert_array_equal(col[:8].cpu().numpy(), oracle.collision(boxes[:8], present[:8], metric='iou', sc=sc[:8]))
    assert (col > 0).float().mean() > 0.01 and 0.2 < (img[:4] > 0).float().mean() < 0.9


def test_config3_full_size_offroad_and_wrong_way(full_size, oracle):
    from oracle import lanelet_oracle
    sim, actions, host, oracle_lanes = full_size
    state0, size, present, act, verts, faces, vcat, cats = host
    off, ww = sim.compute_offroad(), sim.compute_wrong_way()
    assert off.shape == ww.shape == (256, 64)
    pick = [0, 17, 200]
    sub = sim.select_batch_elements(torch.tensor(pick), in_place=False)
    assert torch.equal(sub.compute_offroad(), off[pick]) and torch.equal(sub.compute_wrong_way(), ww[pick])
    s1 = sim.get_state().cpu().numpy()
    sc = torch.stack([torch.sin(sim.get_state()[..., 2]), torch.cos(sim.get_state()[..., 2])], -1).cpu().numpy()
    np.testing.assert_array_equal(off[:4].cpu().numpy(), oracle.offroad(s1[:4], size[:4], verts, faces, 0.5, present=present[:4], sc=sc[:4]))
    ref = lanelet_oracle.lanelet_orientation_loss([oracle_lanes], s1[:1]) * present[:1]
    np.testing.assert_allclose(ww[:1].cpu().numpy(), ref, rtol=0, atol=2e-6)
    assert not off[~torch.from_numpy(present).to(off.device)].any() and (off > 0).any() and (ww > 0).any()


def test_config5_full_size_backward(full_size):
    """gradients through kinematics, IoU, off-road and the rasteriser at B = 256: finite everywhere, zero for the speed column of the image
    term, and equal (to float32 summation order) to the gradients of a sub-batch run on its own"""
    from torchdrivesim_amd.utils import Resolution
    sim, actions, host, _ = full_size
    res = Resolution(256, 256)
    w = torch.rand(256, 64, 3, 256, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))

    def grads(s, act, weight):
        s0 = s.get_state().detach().clone().requires_grad_(True)
        a = act.clone().requires_grad_(True)
        s.kinematic_model.set_state(s0)
        s.step(a)
        img = s.render_egocentric(res=res, fov=35.0)
        loss = (img * weight).sum() / 255.0 + s.compute_collision().sum() + s.compute_offroad().sum()
        loss.backward()
        return s0.grad, a.grad

    full = sim.copy()
    pick = [5, 128, 254]
    part = sim.select_batch_elements(torch.tensor(pick), in_place=False)
    g_state, g_act = grads(full, actions[1], w)
    assert g_state.shape == (256, 64, 4) and g_act.shape == (256, 64, 2)
    assert torch.isfinite(g_state).all() and torch.isfinite(g_act).all() and g_state.abs().sum() > 0 and g_act.abs().sum() > 0
    p_state, p_act = grads(part, actions[1][pick], w[pick])
    np.testing.assert_allclose(p_state.cpu().numpy(), g_state[pick].cpu().numpy(), rtol=2e-4, atol=2e-4 * float(g_state[pick].abs().max()))
    np.testing.assert_allclose(p_act.cpu().numpy(), g_act[pick].cpu().numpy(), rtol=2e-4, atol=2e-4 * float(g_act[pick].abs().max()))


# ------------------------------------------------------------------------------------------------------------------------------------
# The headline shard -- B = 1024 x A = 64 x 256 x 256 per GPU, which is also BASELINE.json's config 4 (8192 scenes over 8 GPUs) as seen by
# one GPU (this size had only ever been run by bench.py and the fuzz scripts).  simulator.py:480-511 of the reference is the
# batch-axis contract: scenes are independent, so any sub-batch must reproduce the full batch bit for bit.
# ------------------------------------------------------------------------------------------------------------------------------------
def test_headline_shard_full_size(oracle):
    import bench
    from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config
    from torchdrivesim_amd.utils import Resolution
    B, A = 1024, 64
    sim, actions, host = bench.build_simulator(B, A, torch.device(DEV), seed=1234 + 3)          # the shard of rank 3 in bench.py
    state0, size, present, act, verts, faces, vcat, cats = host
    sim.step(actions[0])
    res = Resolution(256, 256)
    sim.overlap_infractions = True                    # the metrics of this test run beside the raster launch (second stream) ...
    img = sim.render_egocentric(res=res, fov=35.0)
    col, off = sim.compute_collision(), sim.compute_offroad()
    assert img.shape == (B, A, 3, 256, 256) and img.dtype == torch.float32 and col.shape == off.shape == (B, A)
    # a caller-owned output buffer receives the same image, and the metrics are the same whether they ran beside the rasteriser or after it.
    # The SAME 51.5 GB block serves as that buffer (a second one costs 12 k hipMemCreate and twice its size transiently: VERDICT r4 item 5):
    # every 16th scene is kept as a copy, every camera as two checksums of its bits; the block is cleared and rendered into through `out=`.
    def checksums(t):
        bits = t.view(torch.int32).flatten(2)
        return bits.sum(-1, dtype=torch.int64), (bits[..., ::7].to(torch.int64) * 31 + bits[..., 3::7].to(torch.int64)).sum(-1)
    keep, sums = img[::16].clone(), checksums(img)
    img.zero_()
    assert sim.render_egocentric(res=res, fov=35.0, out=img) is img and torch.equal(img[::16], keep)
    assert all(torch.equal(a, b) for a, b in zip(checksums(img), sums))
    del keep, sums
    sim.overlap_infractions = False                   # ... and behind it (the default)
    assert torch.equal(sim.compute_collision(), col) and torch.equal(sim.compute_offroad(), off)
    # sub-batches (first, middle, last scene and a run across the XCD boundaries of the launch) reproduce the full batch bit for bit
    pick = [0, 127, 128, 511, 512, 640, 1023]
    sub = sim.select_batch_elements(torch.tensor(pick), in_place=False)
    assert torch.equal(sub.render_egocentric(res=res, fov=35.0), img[pick])
    assert torch.equal(sub.compute_collision(), col[pick]) and torch.equal(sub.compute_offroad(), off[pick])
    # the uint8 mode shows the same values over the whole shard (compared slice by slice: no 51 GB temporary)
    u8 = sim.copy()
    u8.renderer = renderer_from_config(HipRendererConfig(out_dtype='uint8'), res=res, fov=35.0)
    u8._scene_cache = None
    img8 = u8.render_egocentric(res=res, fov=35.0)
    assert img8.dtype == torch.uint8
    for lo in range(0, B, 64):
        assert torch.equal(img8[lo:lo + 64].float(), img[lo:lo + 64])
    del img8, u8
    # the oracle on what it can afford: one scene of images (the last one of the shard), eight scenes of collisions and off-road
    s1 = sim.get_state().cpu().numpy()
    sc = torch.stack([torch.sin(sim.get_state()[..., 2]), torch.cos(sim.get_state()[..., 2])], -1).cpu().numpy()
    sv, sa, sf = oracle.static_mesh_arrays(verts, faces, vcat, cats)
    b = B - 1
    mask = np.ascontiguousarray(np.broadcast_to(present[b:b + 1, None, :], (1, A, A)))
    ref = oracle.render_scenes(s1[b:b + 1], size[b:b + 1], mask, s1[b:b + 1, :, :2].copy(), sc[b:b + 1], sv, sa, sf, 35.0, 256, agent_sc=sc[b:b + 1])
    np.testing.assert_array_equal(img[b:b + 1].cpu().numpy(), ref)
    boxes = np.concatenate([s1[..., :2], size, s1[..., 2:3]], -1)
    sl = slice(B - 8, B)
    np.testing.assert_array_equal(col[sl].cpu().numpy(), oracle.collision(boxes[sl], present[sl], metric='iou', sc=sc[sl]))
    np.testing.assert_array_equal(off[sl].cpu().numpy(), oracle.offroad(s1[sl], size[sl], verts, faces, 0.5, present=present[sl], sc=sc[sl]))
    # size-independent sanity of the whole shard: every image has content, absent agents have no off-road loss
    assert bool((img.flatten(2).amax(-1) > 0).all()) and not off[~torch.from_numpy(present).to(off.device)].any()
    assert (col > 0).float().mean() > 0.01


def test_infractions_beside_the_rasteriser_equal_the_serial_ones():
    """compute_collision / compute_offroad / compute_wrong_way asked for after a render run on a second stream beside the raster launch
    (Simulator._beside_render); whatever invalidates the fork -- a new state, an edited mask, another stream, gradients -- takes them
    back to the caller's stream.  Either way the bits are the same."""
    import bench, os
    from torchdrivesim_amd import lanelet2
    from torchdrivesim_amd.utils import Resolution
    osm = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'carla_Town01.osm.gz')
    lanes = lanelet2.load_lanelet_map(osm, origin=(0.0, 0.0))
    sim, actions, _ = bench.build_simulator(32, 64, torch.device(DEV), seed=5, lanelet_map=lanes)
    res = Resolution(256, 256)
    for metric in ('iou', 'discs', 'nograd'):
        from torchdrivesim_amd.simulator import CollisionMetric
        sim.cfg.collision_metric = CollisionMetric(metric)
        sim.step(actions[1])
        sim.overlap_infractions = False
        serial = (sim.compute_collision(), sim.compute_offroad(), sim.compute_wrong_way())
        sim.overlap_infractions = True
        img = sim.render_egocentric(res=res, fov=35.0)
        assert sim._fork is not None
        beside = (sim.compute_collision(), sim.compute_offroad(), sim.compute_wrong_way())
        for a, b in zip(serial, beside):
            assert torch.equal(a, b)
        # results are usable on the caller's stream right away (joined by an event): consume them there
        total = sum(float(t.sum()) for t in beside)
        assert np.isfinite(total)
        # a mask edited in place after the render invalidates the fork: the metric runs in place and sees the edit
        sim.present_mask[:, 1] = False
        col2 = sim.compute_collision()
        sim.overlap_infractions = False
        assert torch.equal(col2, sim.compute_collision())
        sim.overlap_infractions = True
        sim.present_mask[:, 1] = True
    assert type(sim).overlap_infractions is False          # off unless asked for
    # gradients: a differentiable step forks too (round 6; test_differentiable_step_with_metrics_beside_the_rasteriser compares the gradients)
    s0 = sim.get_state().detach().clone().requires_grad_(True)
    sim.kinematic_model.set_state(s0)
    sim.cfg.collision_metric = CollisionMetric('iou')
    img = sim.render_egocentric(res=Resolution(64, 64), fov=35.0)
    (sim.compute_collision().sum() + sim.compute_offroad().sum() + img.sum() / 255.0).backward()
    assert torch.isfinite(s0.grad).all()


@pytest.mark.parametrize('mode', [True, 'reserved'])
def test_differentiable_step_with_metrics_beside_the_rasteriser(mode):
    """BASELINE config 5 with overlap_infractions on (round 6): the metric nodes of a differentiable step are autograd nodes of the side stream --
    forward beside the raster launch, backward beside the rasteriser's backward (the engine runs a node's backward on the stream of its
    forward).  Same values and the same gradients, bit for bit, over several steps; the second one takes the foreseen path."""
    import bench
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device(DEV)
    sim, actions, _ = bench.build_simulator(16, 64, dev, seed=11)
    res = Resolution(256, 256)
    state0 = sim.get_state().clone()
    w = torch.rand(16, 64, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(2))

    def run(overlap):
        sim.overlap_infractions = overlap
        # (the serial reference of 'reserved' runs on the same CU-masked stream: the rasteriser's backward sizes its launch by the CUs it may use,
        # and with it the order of its float sums)
        stream = sim.raster_stream() if mode == 'reserved' else torch.cuda.current_stream(dev)
        torch.cuda.synchronize(dev)
        out, forked = [], 0
        with torch.cuda.stream(stream):
            for i in range(3):
                s0 = state0.clone().requires_grad_(True)
                act = actions[i].clone().requires_grad_(True)
                sim.kinematic_model.set_state(s0)
                sim.step(act)
                img = sim.render_egocentric(res=res, fov=35.0)
                forked += sim._fork is not None
                col, off = sim.compute_collision(), sim.compute_offroad()
                loss = (img * w).sum() / 255.0 + col.sum() + (off * off).sum()
                loss.backward()
                out.append((col.detach().clone(), off.detach().clone(), s0.grad.clone(), act.grad.clone()))
            torch.cuda.synchronize(dev)
        sim.overlap_infractions = False
        return out, forked

    serial, n0 = run(False)
    beside, n1 = run(mode)
    assert n0 == 0 and n1 == 3, 'the differentiable render did not fork'
    for a, b in zip(serial, beside):
        # values AND gradients bit for bit: every backward kernel of the step sums in a fixed order (the whole-scene collision backward since round 6:
        # until then its ds_add_f32 moved an ulp of a gradient from run to run, serial or not)
        for x, y in zip(a, b):
            assert torch.isfinite(x).all() and torch.equal(x, y)
    assert serial[0][2].abs().sum() > 0 and serial[0][1].abs().sum() > 0


def test_foreseen_infractions_are_enqueued_ahead_of_the_raster_launch():
    """With overlap_infractions the metrics a loop asked for after its previous render are computed on the side stream right BEFORE the next
    raster launch (the persistent launch never gives a CU back: VERDICT r3 item 3) and compute_* hands the finished tensors out; a metric
    that was not foreseen runs beside the launch and is foreseen from then on, one that is no longer asked for is dropped.  Same bits as the
    serial order throughout (loop: examples/gym_env.py:83-126 of the reference)."""
    import bench, os
    from torchdrivesim_amd import lanelet2
    from torchdrivesim_amd.utils import Resolution
    osm = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'carla_Town01.osm.gz')
    lanes = lanelet2.load_lanelet_map(osm, origin=(0.0, 0.0))
    sim, actions, _ = bench.build_simulator(16, 64, torch.device(DEV), seed=8, lanelet_map=lanes)
    ref, _, _ = bench.build_simulator(16, 64, torch.device(DEV), seed=8, lanelet_map=lanes)
    res = Resolution(256, 256)
    sim.overlap_infractions = True
    out = torch.empty(16, 64, 3, 256, 256, device=DEV)
    for i in range(5):
        sim.step(actions[i])
        ref.step(actions[i])
        img = sim.render_egocentric(res=res, fov=35.0, out=out)
        ready = sim._fork[3]
        if i == 0:
            assert ready == {}                                               # nothing foreseen yet
        elif i < 4:
            assert set(ready) == {('collision', None), ('offroad',)} | ({('wrong_way',)} if i >= 3 else set())
        col, off = sim.compute_collision(), sim.compute_offroad()
        if 1 <= i:
            assert col is ready[('collision', None)][0] and off is ready[('offroad',)][0]
        assert torch.equal(col, ref.compute_collision()) and torch.equal(off, ref.compute_offroad())
        assert torch.equal(img, ref.render_egocentric(res=res, fov=35.0))
        if i in (2, 3):
            ww = sim.compute_wrong_way()                                      # step 2: not foreseen -> beside the launch; step 3: foreseen
            assert torch.equal(ww, ref.compute_wrong_way())
            assert sim.compute_wrong_way() is ww                              # asked for twice: the same tensor
    assert ('wrong_way',) in sim._fork[3]                                    # step 4 still foresaw it (asked for at step 3) ...
    # an action in between invalidates what was computed ahead: the result is computed again, in place
    sim.step(actions[5])
    ref.step(actions[5])
    sim.render_egocentric(res=res, fov=35.0, out=out)
    assert set(sim._fork[3]) == {('collision', None), ('offroad',)}          # ... and step 5 no longer does
    sim.step(actions[6])
    ref.step(actions[6])
    assert torch.equal(sim.compute_collision(), ref.compute_collision())


@pytest.mark.parametrize('own_stream', [False, True])
def test_metrics_on_reserved_cus_beside_the_raster_launch(own_stream):
    """overlap_infractions = 'reserved' (round 4): the raster launch runs on a stream that is kept off four CUs per XCD (tds_stream_create,
    hipExtStreamCreateWithCUMask) and the metrics on a stream confined to those -- either with the loop itself on the raster stream
    (`with torch.cuda.stream(sim.raster_stream())`) or with the library's detour from the caller's stream.  Same kernels, same bits as the serial
    order, step after step (loop: examples/gym_env.py:83-126 of the reference)."""
    import bench, contextlib
    from torchdrivesim_amd import _ops
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = bench.build_simulator(16, 64, torch.device(DEV), seed=9)
    ref, _, _ = bench.build_simulator(16, 64, torch.device(DEV), seed=9)
    rs, ms = _ops.reserved_streams(torch.device(DEV))
    assert rs.cuda_stream != ms.cuda_stream and sim.raster_stream().cuda_stream == rs.cuda_stream
    res = Resolution(256, 256)
    sim.overlap_infractions = 'reserved'
    out = torch.empty(16, 64, 3, 256, 256, device=DEV)
    torch.cuda.synchronize()
    with (torch.cuda.stream(rs) if own_stream else contextlib.nullcontext()):
        for i in range(4):
            sim.step(actions[i])
            img = sim.render_egocentric(res=res, fov=35.0, out=out)
            col, off = sim.compute_collision(), sim.compute_offroad()
            if i >= 1:
                assert col is sim._fork[3][('collision', None)][0]               # foreseen: enqueued on the reserved CUs ahead of the launch
            ref.step(actions[i])
            torch.cuda.current_stream().synchronize()
            assert torch.equal(img, ref.render_egocentric(res=res, fov=35.0))
            assert torch.equal(col, ref.compute_collision()) and torch.equal(off, ref.compute_offroad())
        # without out=: the image comes from the pool, written on the raster stream, consumed here
        img2 = sim.render_egocentric(res=res, fov=35.0)
        assert torch.equal(img2, out)
    torch.cuda.synchronize()


def test_reserved_detour_waits_for_the_inputs_built_after_the_fork():
    """with overlap_infractions = 'reserved' and the caller NOT on the raster stream, the launch moves to the stream that is
    kept off the reserved CUs.  Its per-launch inputs -- per-camera colour keys (custom_agent_colors), waypoint triangles and their keys -- are
    built on the caller's stream AFTER the fork event, so the launch has to wait for an event recorded behind them.  Rendered many times with
    fresh inputs each time and compared bit for bit with the serial order (loop matched: examples/gym_env.py:83-126 of the reference)."""
    import bench
    from torchdrivesim_amd.utils import Resolution
    B, A = 16, 64
    sim, actions, _ = bench.build_simulator(B, A, torch.device(DEV), seed=21)
    ref, _, _ = bench.build_simulator(B, A, torch.device(DEV), seed=21)
    res = Resolution(256, 256)
    sim.overlap_infractions = 'reserved'
    gen = torch.Generator(device=DEV).manual_seed(5)
    for i in range(6):
        sim.step(actions[i % actions.shape[0]])
        ref.step(actions[i % actions.shape[0]])
        state = sim.get_state()
        # fresh per-launch inputs every time: the values the kernel reads are written right before the launch is enqueued
        cc = torch.rand(B, A, A, 3, device=DEV, generator=gen)
        wp = state[..., None, :2] + torch.rand(B, A, 3, 2, device=DEV, generator=gen) * 20 - 10
        wm = torch.rand(B, A, 3, device=DEV, generator=gen) > 0.2
        kw = dict(res=res, fov=35.0, custom_agent_colors=cc, waypoints=wp, waypoints_rendering_mask=wm)
        img = sim.render(state[..., :2], state[..., 2:3], **kw)
        col = sim.compute_collision()
        assert sim._fork is not None                                   # the detour was taken (float32 256 x 256 is write-bound, the layout verified)
        exp = ref.render(state[..., :2], state[..., 2:3], **kw)
        torch.cuda.synchronize()
        assert torch.equal(img, exp) and torch.equal(col, ref.compute_collision())
        assert (img.flatten(2).amax(-1) > 0).all()


def test_image_ring_is_chosen_among_candidates():
    """rendering.allocate_image_ring on the device: the buffers come from the library's allocator (spread-out physical pages, not torch's
    pool), are distinct, hold a rendered image, and the report carries the yardstick.  The decision logic itself is tested on the CPU with
    injected timings (tests/test_image_ring.py)."""
    import bench
    from torchdrivesim_amd.rendering import allocate_image_ring
    from torchdrivesim_amd.utils import Resolution
    sim, actions, _ = bench.build_simulator(4, 8, torch.device(DEV), seed=3)
    res = Resolution(64, 64)
    ref = sim.render_egocentric(res=res, fov=35.0)
    render = lambda out: sim.render_egocentric(res=res, fov=35.0, out=out)
    before = torch.cuda.memory_allocated(DEV)
    bufs, rep = allocate_image_ring(render, tuple(ref.shape), torch.float32, DEV, count=2, candidates=4)
    assert torch.cuda.memory_allocated(DEV) == before                                   # not torch's memory
    assert len(bufs) == 2 and bufs[0].data_ptr() != bufs[1].data_ptr() and not rep['aliased'] and len(set(rep['kept'])) == 2
    assert len(rep['launch_ms']) == len(rep['first_touch_ms']) >= 2 and rep['fill_ms'] > 0
    assert all(torch.equal(b, ref) for b in bufs)                                      # the probe rendered into them
    assert rep['write_bound'] is False                 # a 4 x 8 x 64 x 64 render is no write stream: the first two candidates are taken
    assert rep['kept'] == [0, 1] and len(rep['launch_ms']) == 2
    del bufs                                           # the buffers go back to the driver with their last tensor
