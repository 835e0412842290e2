#!/usr/bin/env python3
"""Head / tail of the persistent raster launch in strips (raster.hip: claim_work): launch time at B = 256 and B = 1024 (x 64 cameras, 256 x 256
float32, 'reserved' stream: 224 CUs) for several (head, tail, strips) choices, every image compared with the launch without strips.
Testing build (tds_raster_set_tail_split)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from torchdrivesim_amd import _native as nat, _ops                  # noqa: E402
from torchdrivesim_amd.utils import Resolution                      # noqa: E402


def pack(head, tail, lg):
    return head | (tail << 12) | (lg << 24)


def main():
    import ctypes
    if not hasattr(ctypes.CDLL(nat.TESTING_LIB_PATH), 'tds_raster_set_tail_split'):
        raise SystemExit('this probe drove an experiment of round 6 through a testing hook (tds_raster_set_tail_split) that was removed with the experiment: '
                         'it is kept as the record of what was measured (profiles/r06_tail_attempts.log), not as a runnable tool')
    dev = torch.device('cuda', 0)
    res = Resolution(bench.RES, bench.RES)
    with nat.testing() as L:
        for B in (256, 512, 1024):
            sim, actions, _ = bench.build_simulator(B, 64, dev, seed=1234)
            sim.step(actions[0])
            stream = sim.raster_stream()
            bufs = [_ops.owned_image((B, 64, 3, bench.RES, bench.RES), torch.float32, dev) for _ in range(4 if B < 1024 else 2)]
            with torch.cuda.stream(stream):
                for ib, buf in enumerate(bufs):
                    row = []
                    for n in (0, 84, 82, 80, 76, 72, 0):
                        L.tds_raster_set_tail_split(n)
                        for _ in range(2):
                            sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
                        torch.cuda.synchronize()
                        _ops.raster_events = []
                        for _ in range(10):
                            sim.render_egocentric(res=res, fov=bench.FOV, out=buf)
                        torch.cuda.synchronize()
                        ms = [a.elapsed_time(b) for a, b in _ops.raster_events]
                        _ops.raster_events = None
                        row.append(f'{n or "off"}: {np.mean(ms):.3f}')
                    print(f'B={B} buffer {ib}:  ' + '  '.join(row), flush=True)
            L.tds_raster_set_tail_split(0)
            del sim, bufs
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
