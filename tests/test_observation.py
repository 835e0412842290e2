"""Observation model for non-visual policies on the CPU: the oracle's occlusion restatement and this framework's host
logic (noise levels, per-observer accessors) against the reference's outputs (tests/golden/g10_observation.npz)."""
import numpy as np
import torch

from conftest import load_golden


def sim_of(g, tag, device='cpu', noise=None):
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BirdviewMesh
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import NPCController, Simulator, TorchDriveConfig
    A = int(g[f'{tag}_n_exposed'])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    state, size, present = t(g[f'{tag}_state']), t(g[f'{tag}_size']), t(g[f'{tag}_present'])
    B, E = present.shape
    km = KinematicBicycle()
    km.set_params(lr=torch.full((B, A), 1.5, device=device))
    km.set_state(state[:, :A].contiguous())
    npc = NPCController(npc_size=size[:, A:].contiguous(), npc_state=state[:, A:].contiguous(), npc_present_mask=present[:, A:].contiguous()) if E > A else None
    return Simulator(BirdviewMesh.empty(batch_size=B).to(device), km, size[:, :A].contiguous(), present[:, :A].contiguous(),
                     TorchDriveConfig(renderer=HipRendererConfig()), npc_controller=npc, observation_noise_model=noise)


def test_oracle_occlusion_matches_reference(oracle):
    g = load_golden('g10_observation.npz')
    for tag in 'abc':
        m = oracle.occlusion_mask(g[f'{tag}_state'], g[f'{tag}_size'], g[f'{tag}_present'], int(g[f'{tag}_n_exposed']))
        np.testing.assert_array_equal(m, g[f'{tag}_mask'])
        assert (~m & g[f'{tag}_present'][:, None, :]).any()          # somebody is hidden behind somebody


def test_noise_levels_and_noise_free_accessors():
    from torchdrivesim_amd.observation_noise import StandardSensingObservationNoise
    g = load_golden('g10_observation.npz')
    for tag in 'abc':
        sim = sim_of(g, tag)
        dev = StandardSensingObservationNoise().deviation(sim)[..., 0].numpy()
        ref = g[f'{tag}_deviation']
        ok = np.isfinite(ref)
        np.testing.assert_allclose(dev[ok], ref[ok], atol=1e-2)      # the golden is (noisy - true) / eps: the steps are 0.19 .. 3.83
        # base model: what the reference's Simulator returns with the noise-free ObservationNoise
        np.testing.assert_array_equal(sim.get_noisy_all_agents_absolute().numpy(), g[f'{tag}_noisy_absolute'])
        np.testing.assert_allclose(sim.get_noisy_all_agents_relative().numpy(), g[f'{tag}_noisy_relative'], atol=1e-5)
        np.testing.assert_array_equal(sim.get_noisy_all_agents_relative().numpy(), sim.get_all_agents_relative().numpy())
