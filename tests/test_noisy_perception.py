"""
render_egocentric(noisy_perception=True) (reference simulator.py:951-978): the frame shows the observation model's background mesh, its
lane markers (drawn as one-metre arrows of the `stop_sign` category) and its traffic controls instead of the simulator's own.
  * CPU: the mesh this framework assembles for such a frame, fed to the oracle's render_rgb_mesh, makes the calls the reference made at
    the OpenCV boundary (tests/golden/g14_noisy_perception.npz, tools/gen_golden.py:gen_noisy_perception);
  * GPU: the fused path (static map rebuilt from the noisy background, controls as extra quads) gives the oracle's pixels for that mesh.
"""
from collections import Counter

import numpy as np
import pytest
import torch

from conftest import load_golden


def make_sim(g, device='cpu'):
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.lanelet2 import LaneFeatures
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.observation_noise import MapObservationNoiseFromLog
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig
    from torchdrivesim_amd.traffic_controls import StopSignControl, TrafficLightControl
    town = load_golden('town01_mesh.npz')
    cats = [str(c) for c in town['categories']]
    t = lambda k: torch.from_numpy(g[k])
    B = g['state'].shape[0]

    def mesh(prefix):
        return BirdviewMesh(verts=t(prefix + '_verts')[None], faces=t(prefix + '_faces').long()[None], categories=cats,
                            vert_category=t(prefix + '_vert_category').long()[None], colors=dict(), zs=dict()).expand(B)
    true_tc = {'traffic_light': TrafficLightControl(t('lights').clone())}
    true_tc['traffic_light'].set_state(torch.zeros(B, 3, dtype=torch.long))
    noisy_tc = {'traffic_light': TrafficLightControl(t('noisy_lights').clone()), 'stop_sign': StopSignControl(t('noisy_stop').clone())}
    noisy_tc['traffic_light'].set_state(t('noisy_light_state').long())
    markers, mmask = t('markers'), t('markers_mask')
    # the logged observations live where the simulator lives (Simulator.to does not move the observation model, as in the reference)
    model = MapObservationNoiseFromLog(noisy_lane_features=[LaneFeatures(dense_lane_features=markers, dense_lane_features_mask=mmask).to(device)],
                                       noisy_background_mesh=[mesh('noisy').to(device)], noisy_traffic_controls=[{k: v.to(device) for k, v in noisy_tc.items()}])
    km = KinematicBicycle()
    km.set_params(lr=torch.full(g['state'].shape[:2], 1.5))
    km.set_state(t('state').clone())
    truth = LaneFeatures(dense_lane_features=markers[:, :2] + torch.tensor([2.0, 2.0, 0.3, 0.0]), dense_lane_features_mask=torch.ones(B, 2, dtype=torch.bool))
    sim = Simulator(mesh('road'), km, t('size').clone(), t('present').clone(), TorchDriveConfig(renderer=HipRendererConfig()), traffic_controls=true_tc,
                    observation_noise_model=model, lane_features=truth)
    return sim.to(device) if device != 'cpu' else sim


def frame_mesh(sim, noisy):
    """the explicit per-camera mesh of the reference's dataflow for this frame"""
    s = sim.get_state()
    B, A = s.shape[:2]
    gen, controls = sim._noisy_scene_sources() if noisy else (sim.birdview_mesh_generator, sim.traffic_controls)
    mask = sim.get_present_mask()[:, None].expand(B, A, A)
    lights = controls['traffic_light'].extend(A, in_place=False) if controls and 'traffic_light' in controls else None
    rgb = gen.generate(A, agent_state=s[:, None].expand(-1, A, -1, -1), present_mask=mask, traffic_lights=lights)
    cam_sc = torch.stack([torch.sin(s[..., 2]), torch.cos(s[..., 2])], -1)
    return rgb, s[..., :2].reshape(-1, 2), cam_sc.reshape(-1, 2)


@pytest.mark.parametrize('tag', ['noisy', 'plain', 'after'])
def test_call_list_matches_reference(oracle, tag):
    g = load_golden('g14_noisy_perception.npz')
    sim = make_sim(g)
    if tag == 'after':
        sim.internal_time = 1                                   # the log has one entry: from step 1 on the truth is shown again
    rgb, cam_xy, cam_sc = frame_mesh(sim, noisy=tag != 'plain')
    _, tris, cols, cnt = oracle.render_rgb_mesh(rgb.verts.numpy(), rgb.attrs.numpy(), rgb.faces.numpy().astype(np.int32), cam_xy.numpy(), cam_sc.numpy(),
                                                2.0 / 35.0, 96, record=True)
    gt, gc = g[f'tris_{tag}'].reshape(tris.shape[0], -1, 6), g[f'cols_{tag}']
    marker_calls = 0
    for i in range(tris.shape[0]):
        mine = Counter(tuple(tris[i, k]) + tuple(cols[i, k]) for k in range(cnt[i]))
        ref = Counter(tuple(gt[i, k]) + tuple(gc[i, k]) for k in range(gt.shape[1]))
        assert not (mine - ref), f'{tag} image {i}: calls the reference never made'
        for k in (ref - mine):                                   # batch padding of trim: dots on an already-drawn vertex
            assert k[0] == k[2] == k[4] and k[1] == k[3] == k[5]
        marker_calls += sum(v for k, v in mine.items() if k[6:] == (72, 60, 50))
    assert (marker_calls > 0) == (tag != 'plain')


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['noisy', 'after'])
def test_fused_path_equals_the_oracle(oracle, tag):
    from torchdrivesim_amd.utils import Resolution
    g = load_golden('g14_noisy_perception.npz')
    sim = make_sim(g, device='cuda:0')
    if tag == 'after':
        sim.internal_time = 1
    plain = sim.render_egocentric(res=Resolution(96, 96), fov=35.0)
    img = sim.render_egocentric(res=Resolution(96, 96), fov=35.0, noisy_perception=True)
    assert img.shape == plain.shape and (img != plain).any()
    assert torch.equal(sim.render_egocentric(res=Resolution(96, 96), fov=35.0), plain)          # the ordinary scene is back in place
    rgb, cam_xy, cam_sc = frame_mesh(sim, noisy=True)
    ref = oracle.render_rgb_mesh(rgb.verts.cpu().numpy(), rgb.attrs.cpu().numpy(), rgb.faces.cpu().numpy().astype(np.int32), cam_xy.cpu().numpy(),
                                 cam_sc.cpu().numpy(), 2.0 / 35.0, 96)
    ref = np.transpose(ref, (0, 3, 1, 2)).reshape(img.shape)
    assert np.array_equal(img.cpu().numpy(), ref)
    for res, fov in ((256, 35.0), (64, 50.0)):
        big = sim.render_egocentric(res=Resolution(res, res), fov=fov, noisy_perception=True)
        ref = oracle.render_rgb_mesh(rgb.verts.cpu().numpy(), rgb.attrs.cpu().numpy(), rgb.faces.cpu().numpy().astype(np.int32), cam_xy.cpu().numpy(),
                                     cam_sc.cpu().numpy(), 2.0 / fov, res)
        assert np.array_equal(big.cpu().numpy(), np.transpose(ref, (0, 3, 1, 2)).reshape(big.shape))
