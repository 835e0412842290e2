"""Map packages (metadata.json / stop lines) and traffic controls (N3) on the CPU: file formats, state replay, and the
violation predicate restated with the oracle against the reference's golden outputs (tests/golden/g8_traffic.npz)."""
import json
import os

import numpy as np
import torch

from conftest import GOLDEN, load_golden

MAPS = os.path.join(GOLDEN, 'maps')


def test_map_config_round_trip_and_stoplines(tmp_path):
    from torchdrivesim_amd.map import MapConfig, Stopline, find_map_config, load_map_config, store_map_config, traffic_controls_from_map_config
    from torchdrivesim_amd.traffic_controls import TrafficLightControl
    cfg = find_map_config('carla_Town01', resource_path=[MAPS])
    assert cfg is not None and cfg.name == 'carla_Town01' and cfg.left_handed_coordinates is True
    assert os.path.isabs(cfg.stoplines_path) and cfg.stoplines_path.endswith('carla_Town01_stoplines.json')
    assert cfg.mesh_path == 'carla_Town01_mesh.json'            # not shipped with the fixtures: stays relative (map.py:101-113)
    assert find_map_config('no_such_map', resource_path=[MAPS]) is None
    lines = cfg.stoplines
    assert len(lines) == 36 and all(isinstance(s, Stopline) and s.agent_type == 'traffic_light' for s in lines)
    assert Stopline(1, 'stop-sign', 0, 0, 1, 1, 0).agent_type == 'stop_sign' and Stopline(1, 'yield', 0, 0, 1, 1, 0).agent_type == 'yield_sign'
    ctl = traffic_controls_from_map_config(cfg)
    assert set(ctl) == {'traffic_light'} and isinstance(ctl['traffic_light'], TrafficLightControl)
    assert tuple(ctl['traffic_light'].pos.shape) == (1, 36, 5)
    g = load_golden('g8_traffic.npz')
    np.testing.assert_array_equal(ctl['traffic_light'].pos[0].numpy(), g['pos'][0])
    # store -> load keeps the content and writes bare file names
    p = tmp_path / 'metadata.json'
    store_map_config(cfg, str(p))
    raw = json.load(open(p))
    assert raw['stoplines_path'] == 'carla_Town01_stoplines.json'
    again = load_map_config(str(p), resolve_paths=False)
    assert again == MapConfig(**raw)


def test_control_state_replay_and_batch_ops():
    from torchdrivesim_amd.traffic_controls import TrafficLightControl
    g = load_golden('g8_traffic.npz')
    ctl = TrafficLightControl(torch.from_numpy(g['pos']), replay_states=torch.from_numpy(g['replay']), mask=torch.from_numpy(g['mask']))
    assert ctl.allowed_states == ['red', 'yellow', 'green'] and ctl.total_replay_time == 5
    np.testing.assert_allclose(ctl.corners.numpy(), g['corners'], atol=2e-5)
    for t in range(7):
        ctl.step(t)
        np.testing.assert_array_equal(ctl.state.numpy(), g[f'state_{t}'])
    big = ctl.extend(2, in_place=False)
    assert big.pos.shape[0] == 6 and torch.equal(big.state[0], big.state[1]) and torch.equal(big.state[2], ctl.state[1])
    sel = ctl.select_batch_elements(torch.tensor([2, 0]), in_place=False)
    assert torch.equal(sel.pos[0], ctl.pos[2]) and ctl.pos.shape[0] == 3
    cp = ctl.copy()
    cp.set_state(torch.zeros_like(cp.state))
    assert not torch.equal(cp.state, ctl.state) or bool((ctl.state == 0).all())


def test_violation_predicate_restated_with_the_oracle(oracle):
    """rear tenth of the agent box (box2corners_with_rear_factor) overlaps the stop line of a red light: the same boxes through the
    oracle's Rotated-IoU give the reference's answers"""
    from torchdrivesim_amd.traffic_controls import _rear_boxes
    g = load_golden('g8_traffic.npz')
    boxes, pos, mask = g['boxes'], g['pos'], g['mask']
    B, A, N = boxes.shape[0], boxes.shape[1], pos.shape[1]
    rear = _rear_boxes(torch.from_numpy(boxes), 0.1).numpy()
    lines = np.where(mask[..., None], pos, np.array([-1000, -1000, 0, 0, 0], np.float32))
    b1 = np.broadcast_to(rear[:, :, None], (B, A, N, 5)).reshape(-1, 5)
    b2 = np.broadcast_to(lines[:, None], (B, A, N, 5)).reshape(-1, 5)
    overlap = (oracle.iou_pairs(b1, b2) > 0).reshape(B, A, N)
    for t in range(7):
        red = g[f'state_{t}'] == 0
        np.testing.assert_array_equal((overlap & red[:, None]).any(-1), g[f'violation_{t}'])


def test_generator_reproduces_the_reference_mesh_with_traffic_controls():
    """generate() with stop-sign / yield / traffic-light quads against the mesh the reference generated (g9_traffic_mesh.npz)"""
    from torchdrivesim_amd.mesh import BirdviewMesh, BirdviewRGBMeshGenerator
    from torchdrivesim_amd.rendering import get_default_color_map, get_default_rendering_levels
    from torchdrivesim_amd.traffic_controls import StopSignControl, TrafficLightControl, YieldControl
    g = load_golden('g9_traffic_mesh.npz')
    t = torch.from_numpy
    B, A = g['state'].shape[:2]
    bg = BirdviewMesh(verts=t(g['bg_verts'])[None], faces=t(g['bg_faces'].astype(np.int64))[None], categories=['right_lane', 'left_lane', 'road'],
                      colors={}, zs={}, vert_category=t(g['bg_vert_category'].astype(np.int64))[None]).expand(B)
    tl = TrafficLightControl(t(g['tl_pos']), mask=t(g['tl_mask']))
    tl.set_state(t(g['tl_state']))
    controls = dict(stop_sign=StopSignControl(t(g['ss_pos'])), traffic_light=tl, yield_sign=YieldControl(t(g['ys_pos'])))
    gen = BirdviewRGBMeshGenerator(bg, get_default_color_map(), get_default_rendering_levels())
    gen.initialize_actors_mesh(t(g['size']), torch.zeros(B, A, dtype=torch.long), ['vehicle'])
    gen.initialize_traffic_controls_mesh(controls)
    nc = g['rgb_verts'].shape[0] // B
    rgb = gen.generate(nc, agent_state=t(g['state'])[:, None].expand(-1, nc, -1, -1), present_mask=t(g['present'])[:, None].expand(B, nc, A),
                       traffic_lights=tl.extend(nc, in_place=False))
    np.testing.assert_array_equal(rgb.faces.numpy(), g['rgb_faces'])
    np.testing.assert_allclose(rgb.verts.numpy(), g['rgb_verts'], atol=0, rtol=0)
    np.testing.assert_array_equal(rgb.attrs.numpy(), g['rgb_attrs'])
