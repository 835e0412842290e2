"""Gradients of the HIP backward kernels (BASELINE config 5) against the reference's torch-autograd gradients captured in
fixture G7 (tools/gen_golden.py: loss = sum(compute_collision) + sum(compute_offroad) after one KinematicBicycle step)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make_sim(g, metric, state, action=None):
    from torchdrivesim_amd.kinematic import KinematicBicycle
    from torchdrivesim_amd.mesh import BaseMesh
    from torchdrivesim_amd.rendering import HipRendererConfig
    from torchdrivesim_amd.simulator import Simulator, TorchDriveConfig, CollisionMetric
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    B = state.shape[0]
    road = BaseMesh(verts=d(g['road_verts'])[None].expand(B, -1, -1).contiguous(), faces=d(g['road_faces'].astype(np.int64))[None].expand(B, -1, -1).contiguous())
    km = KinematicBicycle()
    km.set_params(lr=d(g['lr']))
    km.set_state(state)
    cfg = TorchDriveConfig(collision_metric=CollisionMetric(metric), renderer=HipRendererConfig())
    return Simulator(road, km, d(g['size']), d(g['present']), cfg)


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_collision_and_offroad_gradients_wrt_state(metric):
    g = load_golden('g7_grads.npz')
    s1 = torch.from_numpy(g[f'{metric}_state1']).to(DEV).requires_grad_(True)
    sim = make_sim(g, metric, s1)
    coll = sim.compute_collision()
    np.testing.assert_allclose(coll.detach().cpu().numpy(), g[f'{metric}_coll'], atol=2e-6)
    gc, = torch.autograd.grad(coll.sum(), s1)
    ref = g[f'{metric}_grad_coll_wrt_state1']
    np.testing.assert_allclose(gc.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(ref).max()))
    assert np.abs(ref).max() > 0
    off = sim.compute_offroad()
    np.testing.assert_allclose(off.detach().cpu().numpy(), g[f'{metric}_off'], rtol=1e-5, atol=1e-5)
    go, = torch.autograd.grad(off.sum(), s1)
    ref = g[f'{metric}_grad_off_wrt_state1']
    np.testing.assert_allclose(go.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * max(1.0, np.abs(ref).max()))
    assert np.abs(ref).max() > 0


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_full_step_gradients_match_reference(metric):
    """state0, action -> step (K1) -> collision (K2a) + offroad (K2b) -> backward through all three kernels."""
    g = load_golden('g7_grads.npz')
    s0 = torch.from_numpy(g['state0']).to(DEV).requires_grad_(True)
    act = torch.from_numpy(g['action']).to(DEV).requires_grad_(True)
    sim = make_sim(g, metric, s0)
    sim.step(act)
    np.testing.assert_allclose(sim.get_state().detach().cpu().numpy(), g[f'{metric}_state1'], rtol=1e-5, atol=1e-5)
    loss = sim.compute_collision().sum() + sim.compute_offroad().sum()
    loss.backward()
    for got, key in ((s0.grad, f'{metric}_grad_state'), (act.grad, f'{metric}_grad_action')):
        ref = g[key]
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=3e-3, atol=3e-4 * max(1.0, np.abs(ref).max()))


def test_collision_gradient_finite_differences():
    """independent check of the analytic IoU / discs gradients: central differences of the forward kernel in float64-ish steps"""
    from torchdrivesim_amd import _ops
    gen = torch.Generator().manual_seed(3)
    B, N = 2, 5
    xy = (torch.rand(B, N, 2, generator=gen) - 0.5) * 5
    boxes = torch.cat([xy, 4 + torch.rand(B, N, 1, generator=gen), 1.8 + 0.4 * torch.rand(B, N, 1, generator=gen),
                       (torch.rand(B, N, 1, generator=gen) - 0.5) * 6], -1).to(DEV)
    present = torch.ones(B, N, dtype=torch.bool, device=DEV)
    wgt = torch.rand(B, N, generator=gen).to(DEV)
    for metric in ('iou', 'discs'):
        b = boxes.clone().requires_grad_(True)
        (_ops.collision(b, present, metric=metric) * wgt).sum().backward()
        an = b.grad.cpu().numpy()
        num = np.zeros_like(an)
        eps = 2e-3
        for bi in range(B):
            for ni in range(N):
                for k in range(5):
                    bp, bm = boxes.clone(), boxes.clone()
                    bp[bi, ni, k] += eps
                    bm[bi, ni, k] -= eps
                    fp = (_ops.collision(bp, present, metric=metric) * wgt).sum().item()
                    fm = (_ops.collision(bm, present, metric=metric) * wgt).sum().item()
                    num[bi, ni, k] = (fp - fm) / (2 * eps)
        np.testing.assert_allclose(an, num, rtol=0.05, atol=0.02 * max(1.0, np.abs(num).max()))


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_collision_gradient_of_a_large_scene_equals_that_of_its_far_apart_halves(metric):
    """Scenes of up to 16 384 pairs take the whole-scene backward kernel, larger ones one wavefront per row: a scene of 132 agents in two
    clusters a kilometre apart (17 424 pairs: row kernel) must give each cluster the gradients it gets as a scene of its own (scene kernel)."""
    from torchdrivesim_amd import _ops
    gen = torch.Generator().manual_seed(11)
    B, H = 3, 66
    def cluster():
        xy = (torch.rand(B, H, 2, generator=gen) - 0.5) * 40
        return torch.cat([xy, 4 + torch.rand(B, H, 1, generator=gen), 1.8 + 0.4 * torch.rand(B, H, 1, generator=gen),
                          (torch.rand(B, H, 1, generator=gen) - 0.5) * 6], -1)
    c1, c2 = cluster(), cluster()
    c2[..., 0] += 1000.0
    present = torch.rand(B, 2 * H, generator=gen) > 0.1
    wgt = torch.rand(B, 2 * H, generator=gen)
    wgt[:, ::7] = 0.0                                   # rows without a gradient are skipped by both kernels
    whole = torch.cat([c1, c2], 1).to(DEV).requires_grad_(True)
    out = _ops.collision(whole, present.to(DEV), metric=metric)
    (out * wgt.to(DEV)).sum().backward()
    assert (out > 0).sum().item() > 20
    for k, c in enumerate((c1, c2)):
        part = c.to(DEV).requires_grad_(True)
        sl = slice(k * H, (k + 1) * H)
        o = _ops.collision(part, present[:, sl].to(DEV), metric=metric)
        torch.testing.assert_close(o, out[:, sl].detach(), rtol=0, atol=0)
        (o * wgt[:, sl].to(DEV)).sum().backward()
        assert part.grad.abs().max().item() > 0
        torch.testing.assert_close(whole.grad[:, sl], part.grad, rtol=1e-5, atol=1e-6 * part.grad.abs().max().item())


@pytest.mark.parametrize('metric', ['iou', 'discs'])
def test_collision_gradient_of_a_scene_is_bit_reproducible(metric):
    """The whole-scene collision backward is deterministic since round 6 (it summed per box with LDS float atomics in arrival order: an ulp of a
    gradient moved from run to run): near list in pair order, a slot per pair, every box collects its contributions in a fixed order.  Ten
    runs of the same backward give the same bits -- on a sparse scene and on a DENSE one (64 agents within a few metres: about 4 000 near pairs, worked
    off in several chunks of the 1 024-pair table) -- and the dense gradients agree with the row kernel's (one wavefront per row, atomics) to rounding."""
    from torchdrivesim_amd import _ops
    gen = torch.Generator().manual_seed(5)
    for spread, B, A in ((60.0, 8, 64), (6.0, 4, 64)):
        xy = (torch.rand(B, A, 2, generator=gen) - 0.5) * spread
        boxes = torch.cat([xy, 4 + torch.rand(B, A, 1, generator=gen), 1.8 + 0.4 * torch.rand(B, A, 1, generator=gen), (torch.rand(B, A, 1, generator=gen) - 0.5) * 6], -1).to(DEV)
        present = (torch.rand(B, A, generator=gen) > 0.1).to(DEV)
        wgt = torch.rand(B, A, generator=gen).to(DEV)

        def grad(t):
            t = t.clone().requires_grad_(True)
            out = _ops.collision(t, present, metric=metric)
            (out * wgt).sum().backward()
            return out.detach(), t.grad

        out, first = grad(boxes)
        assert (out > 0).sum().item() > 10 and first.abs().max().item() > 0 and torch.isfinite(first).all()
        for _ in range(9):
            assert torch.equal(grad(boxes)[1], first)
        # exposed agents among more boxes (NPCs: 64 x 100 = 6 400 pairs, beyond the forward scene kernel's 4 096): the same kernel, the same bits every run
        npc = torch.cat([(torch.rand(B, 36, 2, generator=gen) - 0.5) * spread, 4 + torch.rand(B, 36, 1, generator=gen), 1.9 + torch.zeros(B, 36, 1),
                         (torch.rand(B, 36, 1, generator=gen) - 0.5) * 6], -1).to(DEV)
        allb, allp = torch.cat([boxes, npc], 1), torch.cat([present, torch.ones(B, 36, dtype=torch.bool, device=DEV)], 1)

        def grad_npc():
            t = allb.clone().requires_grad_(True)
            out = _ops.collision(t, allp, n_exposed=A, metric=metric)
            (out * wgt).sum().backward()
            return t.grad

        g0 = grad_npc()
        assert g0[:, A:].abs().max().item() > 0                      # the NPC boxes get gradients too
        for _ in range(5):
            assert torch.equal(grad_npc(), g0)
        if spread < 10:
            # the same scene as rows of a scene too large for the whole-scene kernel (129 x 129 > 16 384 pairs: 65 extra boxes, absent and far away)
            far = torch.tensor([1.0e4, 1.0e4, 4.0, 2.0, 0.0], device=DEV).expand(B, 65, 5)
            big = torch.cat([boxes, far], 1).clone().requires_grad_(True)
            o2 = _ops.collision(big, torch.cat([present, torch.zeros(B, 65, dtype=torch.bool, device=DEV)], 1), metric=metric)
            (o2[:, :A] * wgt).sum().backward()
            torch.testing.assert_close(big.grad[:, :A], first, rtol=2e-4, atol=2e-5 * first.abs().max().item())
