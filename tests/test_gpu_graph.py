"""The whole step -- Simulator.step -> render_egocentric(out=) -> compute_collision -> compute_offroad -- captured into a HIP graph and
replayed, with the metrics behind the launch, beside it on a plain second stream, and beside it on reserved CUs (the CU masks
are not carried into a graph; the results are): no per-call entry point of the library allocates or synchronises (the work queues of the persistent raster
launch live in the caller's workspace and are cleared by a zeroing KERNEL, tds::zero_async -- a memset node of a captured graph is not ordered before
the kernel node that follows it on ROCm 7.0), so a captured step equals the eager one bit for bit.
Loop matched: examples/gym_env.py:83-126 of the reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('overlap', [False, True, 'reserved'])
def test_a_captured_step_replays_bit_for_bit(overlap):
    import bench
    from torchdrivesim_amd.utils import Resolution
    B, A = 24, 64
    res = Resolution(256, 256)
    sim, actions, _ = bench.build_simulator(B, A, torch.device(DEV), seed=21)
    ref, _, _ = bench.build_simulator(B, A, torch.device(DEV), seed=21)
    sim.overlap_infractions = overlap
    state = sim.get_state().clone()                       # static tensors of the graph: state in / out, action in, image out
    action = actions[0].clone()
    image = torch.empty(B, A, 3, 256, 256, device=DEV)
    sim.kinematic_model.set_state(state)

    def step():
        sim.kinematic_model.set_state(state)
        sim.step(action)
        img = sim.render_egocentric(res=res, fov=35.0, out=image)
        col, off = sim.compute_collision(), sim.compute_offroad()
        new = sim.get_state()
        return img, col, off, new

    # eager warm-up on a side stream (as torch asks for before capture): builds the device scene, the workspaces, the side stream
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        img, col, off, new = step()
        state_next = new.clone()
    for i in range(4):
        # the eager reference
        ref.kinematic_model.set_state(state.clone())
        ref.step(actions[i])
        want_img = ref.render_egocentric(res=res, fov=35.0)
        want = (want_img, ref.compute_collision(), ref.compute_offroad(), ref.get_state())
        action.copy_(actions[i])
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(img, want[0]) and torch.equal(col, want[1]) and torch.equal(off, want[2]) and torch.equal(new, want[3])
        assert bool(img.flatten(2).amax(-1).gt(0).all())
        state.copy_(state_next)                           # the next replay continues from here


def test_a_captured_differentiable_step_replays_with_the_eager_gradients():
    """Forward AND backward of BASELINE config 5's step -- kinematics, rasteriser (index slices + its backward), IoU collision, off-road -- captured
    into one HIP graph (round 6): nothing in the library's autograd functions allocates outside torch's allocator or
    synchronises, so `torch.autograd.grad` inside `torch.cuda.graph` records the whole step; replays give the eager gradients.  (Measured at B = 256: the replay is not faster than the eager loop on
    ROCm 7.0 -- 9.85 against 9.60 ms per step incl. the loss probe, tools/config5_graph_probe.py -- so bench.py keeps config 5 eager.)"""
    import bench
    from torchdrivesim_amd.utils import Resolution
    dev = torch.device(DEV)
    B, A = 12, 64
    res = Resolution(256, 256)
    sim, actions, _ = bench.build_simulator(B, A, dev, seed=33)
    state0 = sim.get_state().clone()
    w = torch.rand(B, A, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(4))

    def fwd_bwd(s0, act):
        sim.kinematic_model.set_state(s0)
        sim.step(act)
        img = sim.render_egocentric(res=res, fov=35.0)
        col, off = sim.compute_collision(), sim.compute_offroad()
        loss = (img * w).sum() / 255.0 + col.sum() + (off * off).sum()
        return torch.autograd.grad(loss, [s0, act]) + (col.detach(), off.detach())

    s_in = state0.clone().requires_grad_(True)
    a_in = actions[0].clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd(s_in, a_in)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd_bwd(s_in, a_in)
    for i in range(3):
        with torch.no_grad():
            a_in.copy_(actions[i + 1])
        g.replay()
        torch.cuda.synchronize()
        want = fwd_bwd(state0.clone().requires_grad_(True), actions[i + 1].clone().requires_grad_(True))
        assert torch.equal(out[2], want[2]) and torch.equal(out[3], want[3])                  # forward values: bit for bit
        for x, y in zip(want[:2], out[:2]):
            assert torch.isfinite(y).all() and float(x.abs().max()) > 0
            torch.testing.assert_close(y, x, rtol=1e-5, atol=1e-6 * float(x.abs().max()))
