#!/usr/bin/env python3
"""BASELINE config 5 (B = 256 x 64: step + render + collision + off-road, then backward through all of it) captured into ONE HIP graph -- forward AND
backward -- against the eager loop of bench.other_configs ("or capture forward + backward in a HIP graph").  Prints ms per step of
both and checks that the replayed gradients equal the eager ones."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from torchdrivesim_amd import _ops                                  # noqa: E402
from torchdrivesim_amd.utils import Resolution                      # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    B, A, steps = 256, 64, 20
    res = Resolution(bench.RES, bench.RES)
    sim, actions, _ = bench.build_simulator(B, A, dev, seed=1234)
    state0 = sim.get_state().clone()
    w = torch.rand(B, A, 3, bench.RES, bench.RES, device=dev)

    def fwd_bwd(s0, act):
        sim.kinematic_model.set_state(s0)
        sim.step(act)
        img = sim.render_egocentric(res=res, fov=bench.FOV)
        loss = bench._ImageProbe.apply(img, w) + sim.compute_collision().sum() + sim.compute_offroad().sum()
        return torch.autograd.grad(loss, [s0, act])

    def eager(i):
        s0 = state0.clone().requires_grad_(True)
        act = actions[i % 8].clone().requires_grad_(True)
        return fwd_bwd(s0, act)

    for i in range(3):
        g_ref = eager(i)
    torch.cuda.synchronize()
    bench._ImageProbe.events.clear()
    t0 = time.perf_counter()
    for i in range(steps):
        eager(i)
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / steps * 1e3
    probe = float(np.mean([a.elapsed_time(b) for a, b in bench._ImageProbe.events[-steps:]]))
    bench._ImageProbe.events.clear()
    print(f'eager: {t_eager:.3f} ms per step, loss probe {probe:.3f} -> library {t_eager - probe:.3f} ms', flush=True)

    # static inputs of the graph
    s_in = state0.clone().requires_grad_(True)
    a_in = actions[0].clone().requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd(s_in, a_in)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gs, ga = fwd_bwd(s_in, a_in)
    torch.cuda.synchronize()
    # same gradients as eager?
    for i in range(3):
        with torch.no_grad():
            a_in.copy_(actions[i % 8])
        g.replay()
        torch.cuda.synchronize()
        want = eager(i)
        ok = torch.allclose(gs, want[0], rtol=1e-5, atol=1e-6 * float(want[0].abs().max())) and torch.allclose(ga, want[1], rtol=1e-5, atol=1e-6 * float(want[1].abs().max()))
        print(f'replay {i}: gradients {"equal" if ok else "DIFFER from"} the eager step (max |d| state {float((gs - want[0]).abs().max()):.3e}, action {float((ga - want[1]).abs().max()):.3e})', flush=True)
    bench._ImageProbe.events.clear()
    t0 = time.perf_counter()
    for i in range(steps):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / steps * 1e3
    print(f'graph replay: {t_graph:.3f} ms per step (the loss probe is inside: {probe:.3f} eager) -> library about {t_graph - probe:.3f} ms', flush=True)


if __name__ == '__main__':
    main()
