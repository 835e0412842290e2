"""The compact autograd nodes of the differentiable step (round 4: 120 -> 56 launches per step) against what torch's own autograd gives for the
reference's expressions -- [sin psi, cos psi] of a state (simulator.py:940, utils.py:40-53) and the [x, y, length, width, psi] boxes
(simulator.py:1093).  Pure torch, so they run on the CPU."""
import torch

from torchdrivesim_amd import _ops


def test_state_heading_sc_equals_torch_autograd():
    torch.manual_seed(0)
    state = torch.randn(3, 5, 4, requires_grad=True)
    w = torch.randn(3, 5, 2)
    sc = _ops.state_heading_sc(state)
    ref_state = state.detach().clone().requires_grad_(True)
    ref = torch.stack([torch.sin(ref_state[..., 2]), torch.cos(ref_state[..., 2])], dim=-1)
    assert torch.equal(sc, ref)                                        # the same torch.sin / torch.cos, bit for bit
    # three consumers share the node: their gradients are summed before its backward runs once
    ((sc * w).sum() + (sc ** 2 * w).sum() + sc[..., 0].sum()).backward()
    ((ref * w).sum() + (ref ** 2 * w).sum() + ref[..., 0].sum()).backward()
    torch.testing.assert_close(state.grad, ref_state.grad, rtol=1e-6, atol=1e-7)
    assert torch.equal(state.grad[..., [0, 1, 3]], torch.zeros(3, 5, 3))
    # the node can be walked again (nothing was freed with the first backward)
    state.grad = None
    sc2 = _ops.state_heading_sc(state)
    loss = (sc2 * w).sum()
    loss.backward(retain_graph=True)
    g1 = state.grad.clone()
    loss.backward()
    torch.testing.assert_close(state.grad, 2 * g1)


def test_heading_sc_without_gradients_is_torch_sin_cos():
    psi = torch.linspace(-7.0, 7.0, 1001)
    sc = _ops.heading_sc(psi)
    assert torch.equal(sc[..., 0], torch.sin(psi)) and torch.equal(sc[..., 1], torch.cos(psi)) and sc.is_contiguous()
    p = psi.clone().requires_grad_(True)
    assert _ops.heading_sc(p).requires_grad                            # the differentiable form stays plain torch


def test_boxes_node_equals_torch_autograd():
    torch.manual_seed(1)
    state = torch.randn(2, 7, 4, requires_grad=True)
    size = torch.rand(2, 7, 2, requires_grad=True)
    w = torch.randn(2, 7, 5)
    boxes = _ops.state_boxes(state, size)
    rs, rz = state.detach().clone().requires_grad_(True), size.detach().clone().requires_grad_(True)
    ref = torch.cat([rs[..., :2], rz, rs[..., 2:3]], dim=-1)
    assert torch.equal(boxes, ref)
    (boxes * w).sum().backward()
    (ref * w).sum().backward()
    assert torch.equal(state.grad, rs.grad) and torch.equal(size.grad, rz.grad)
    # sizes that need no gradient get none
    b2 = _ops.state_boxes(state, size.detach())
    state.grad = None
    (b2 * w).sum().backward()
    assert torch.equal(state.grad, rs.grad)
