"""Waypoint goals ("waypoint meshes in the renderer"): WaypointGoal bookkeeping, the disc mesh and generate()'s waypoint
branch against G11 from the reference (goals.py, mesh.py:885-909,1120-1145,1243-1271), and -- through the oracle -- the call list the
reference hands to OpenCV when render_egocentric draws waypoint goals.  CPU only."""
from collections import Counter

import numpy as np
import torch

from conftest import load_golden


def make_goal(g):
    from torchdrivesim_amd.goals import WaypointGoal
    return WaypointGoal(torch.from_numpy(g['wp']), torch.from_numpy(g['wp_mask']))


def test_waypoint_goal_bookkeeping_matches_reference():
    g = load_golden('g11_waypoints.npz')
    goal = make_goal(g)
    assert goal.state.shape == (2, 3, 1) and goal.state.dtype == torch.long and goal.max_goal_idx == 4
    traj = torch.from_numpy(g['traj'])
    for t in range(traj.shape[0]):
        goal.step(traj[t], t + 1, threshold=2.0)
        np.testing.assert_array_equal(goal.state.numpy(), g[f'state_{t}'])
        np.testing.assert_array_equal(goal.mask.numpy(), g[f'mask_{t}'])
        for c in (1, 3):
            np.testing.assert_array_equal(goal.get_waypoints(c).numpy(), g[f'get_wp_{c}_{t}'])
            np.testing.assert_array_equal(goal.get_masks(c).numpy(), g[f'get_mask_{c}_{t}'])
    assert int(goal.state.max()) == 3                                      # some agents reached the last collection ...
    assert int(goal.state.min()) < 3                                       # ... and one is stuck behind a padding-only collection
    ext = goal.extend(2, in_place=False)
    np.testing.assert_array_equal(ext.state.numpy(), g['ext_state'])
    np.testing.assert_array_equal(ext.waypoints.numpy(), g['ext_wp'])
    assert goal.state.shape[0] == 2
    sel = goal.select_batch_elements([1], in_place=False)
    assert torch.equal(sel.waypoints[0], goal.waypoints[1]) and torch.equal(sel.state[0], goal.state[1])
    cp = goal.copy()
    cp.state += 1
    assert not torch.equal(cp.state, goal.state)


def test_disc_mesh_matches_reference():
    from torchdrivesim_amd.mesh import generate_disc_mesh
    g = load_golden('g11_waypoints.npz')
    for r, n in ((2.0, 10), (1.5, 6), (3.0, 2)):
        v, f = generate_disc_mesh(radius=r, num_triangles=n)
        np.testing.assert_array_equal(v.numpy(), g[f'disc_{n}_verts'])
        np.testing.assert_array_equal(f.numpy(), g[f'disc_{n}_faces'])


def _generator(verts, faces, vert_category, B, size):
    from torchdrivesim_amd.mesh import BirdviewMesh, BirdviewRGBMeshGenerator
    from torchdrivesim_amd.rendering import get_default_color_map, get_default_rendering_levels
    town = load_golden('town01_mesh.npz')
    bg = BirdviewMesh(verts=torch.from_numpy(verts)[None], faces=torch.from_numpy(faces.astype(np.int64))[None],
                      categories=[str(c) for c in town['categories']], vert_category=torch.from_numpy(vert_category.astype(np.int64))[None],
                      colors=dict(), zs=dict()).expand(B)
    gen = BirdviewRGBMeshGenerator(bg, get_default_color_map(), get_default_rendering_levels())
    gen.initialize_actors_mesh(size, torch.zeros(size.shape[:2], dtype=torch.long), ['vehicle'])
    return gen


def test_generate_with_waypoints_matches_reference():
    g = load_golden('g11_waypoints.npz')
    state, size, present = (torch.from_numpy(g[k]) for k in ('m_state', 'm_size', 'm_present'))
    B, A = state.shape[:2]
    gen = _generator(g['m_bg_verts'], g['m_bg_faces'], g['m_bg_vert_category'], B, size)
    rgb = gen.generate(A, agent_state=state[:, None].expand(-1, A, -1, -1), present_mask=present[:, None].expand(B, A, A),
                       waypoints=torch.from_numpy(g['m_wp']), waypoints_rendering_mask=torch.from_numpy(g['m_wmask']))
    nv0, nf0 = g['m_bg_verts'].shape[0] + 7 * A, g['m_bg_faces'].shape[0] + 3 * A
    np.testing.assert_array_equal(rgb.verts[:, nv0:].numpy(), g['m_wp_verts'])
    np.testing.assert_array_equal(rgb.faces[:, nf0:].numpy(), g['m_wp_faces'])       # masked waypoints alias the first waypoint vertex
    np.testing.assert_array_equal(rgb.attrs[:, nv0:].numpy(), g['m_wp_attrs'])
    # copies of the generator keep the disc
    assert gen.expand(2).waypoint_mesh.batch_size == 2 * B and gen.select_batch_elements([0]).waypoint_mesh.batch_size == 1


def sim_with_goals(g, device='cpu', renderer=None):
    from torchdrivesim_amd.goals import WaypointGoal
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig
    town = load_golden('town01_mesh.npz')
    st, sz, pr = (torch.from_numpy(g[k]).to(device) for k in ('r_state', 'r_size', 'r_present'))
    B = st.shape[0]
    road = BirdviewMesh(verts=torch.from_numpy(g['r_road_verts'])[None], faces=torch.from_numpy(g['r_road_faces'].astype(np.int64))[None],
                        categories=[str(c) for c in town['categories']], vert_category=torch.from_numpy(g['r_road_vert_category'].astype(np.int64))[None],
                        colors=dict(), zs=dict()).expand(B).to(device)
    km = KinematicBicycle()
    km.set_params(lr=torch.full(st.shape[:2], 1.5, device=device))
    km.set_state(st)
    goals = WaypointGoal(torch.from_numpy(g['r_wp']).to(device), torch.from_numpy(g['r_wmask']).to(device))
    goals.state = torch.from_numpy(g['r_goal_state']).to(device)
    return Simulator(road, km, sz, pr, TorchDriveConfig(renderer=HipRendererConfig()), waypoint_goals=goals, renderer=renderer)


def test_render_call_list_with_waypoints_matches_reference(oracle):
    """the explicit per-camera mesh of generate() with the simulator's current waypoints, fed to the oracle's render_rgb_mesh, makes the
    calls the reference made at the OpenCV boundary"""
    g = load_golden('g11_waypoints.npz')
    sim = sim_with_goals(g)
    assert sim.get_waypoints(2).shape == (2, 4, 4, 2) and sim.get_waypoints_mask(2).shape == (2, 4, 4) and sim.get_waypoints_state().shape == (2, 4, 1)
    s = sim.get_state()
    B, A = s.shape[:2]
    mask = sim.get_present_mask()[:, None].expand(B, A, A)
    cam_sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1)
    for count in (1, 2):
        rgb = sim.birdview_mesh_generator.generate(A, agent_state=s[:, None].expand(-1, A, -1, -1), present_mask=mask,
                                                   waypoints=sim.get_waypoints(count), waypoints_rendering_mask=sim.get_waypoints_mask(count))
        _, tris, cols, cnt = oracle.render_rgb_mesh(rgb.verts.numpy(), rgb.attrs.numpy(), rgb.faces.numpy().astype(np.int32),
                                                    s[..., :2].reshape(-1, 2).numpy(), cam_sc.reshape(-1, 2).numpy(), 2.0 / 35.0, 96, record=True)
        gt, gc = g[f'r_tris_{count}'].reshape(B * A, -1, 6), g[f'r_cols_{count}']
        seen_wp = 0
        for i in range(B * A):
            mine = [tuple(tris[i, k]) + tuple(cols[i, k]) for k in range(cnt[i])]
            ref = [tuple(gt[i, k]) + tuple(gc[i, k]) for k in range(gt.shape[1])]
            cm, cr = Counter(mine), Counter(ref)
            assert not (cm - cr), f'count {count} image {i}: calls the reference never made'
            for k in (cr - cm):                                  # batch padding of trim: dots on an already-drawn vertex
                assert k[0] == k[2] == k[4] and k[1] == k[3] == k[5]
            runs = lambda seq: [c for j, c in enumerate(seq) if j == 0 or c != seq[j - 1]]
            extra = dict(cr - cm)
            ref_wo_pad = [k for k in ref if not (extra.get(k, 0) > 0 and not extra.__setitem__(k, extra[k] - 1))]
            assert runs([k[6:] for k in mine]) == runs([k[6:] for k in ref_wo_pad])
            seen_wp += sum(1 for k in mine if k[6:] == (139, 64, 0))
        assert seen_wp > 0


def test_simulator_steps_and_plumbs_waypoint_goals():
    g = load_golden('g11_waypoints.npz')
    sim = sim_with_goals(g)
    other = sim.copy()
    other.waypoint_goals.state += 1
    assert not torch.equal(other.get_waypoints_state(), sim.get_waypoints_state())
    assert sim.extend(2, in_place=False).get_waypoints().shape[0] == 4
    assert sim[[1]].get_waypoints_mask().shape[0] == 1
    # put agent (0, 0) on its first valid current waypoint: the collection is ticked off by the bookkeeping step of Simulator.step
    goals = sim.waypoint_goals
    j = int(torch.nonzero(goals.get_masks()[0, 0])[0])
    before = goals.state.clone()
    here = goals.get_waypoints()[0, 0, j]
    goals.step(torch.cat([here, torch.zeros(2)])[None, None].expand(2, 4, 4).clone(), 1, threshold=sim.cfg.waypoint_removal_threshold)
    assert int(goals.state[0, 0]) == int(before[0, 0]) + 1
