#!/usr/bin/env python3
"""The split bit-plane form as a pipeline (raster.hip: K3s of the next slice of the cameras beside K3r of this one): ms per render call at
B = 1024 x 64 for 1 (no pipeline) .. 8 slices, float32 64 / 128 and uint8 128 / 192, every image compared with the unsliced call.
Testing build (tds_raster_set_split_chunks)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                       # noqa: E402
from torchdrivesim_amd import _native as nat, _ops                  # noqa: E402
from torchdrivesim_amd.rendering import HipRendererConfig, renderer_from_config   # noqa: E402
from torchdrivesim_amd.utils import Resolution                      # noqa: E402


def main():
    import ctypes
    if not hasattr(ctypes.CDLL(nat.TESTING_LIB_PATH), 'tds_raster_set_split_chunks'):
        raise SystemExit('this probe drove an experiment of round 6 through a testing hook (tds_raster_set_split_chunks) that was removed with the experiment: '
                         'it is kept as the record of what was measured (profiles/r06_split_pipeline_attempt.log), not as a runnable tool')
    dev = torch.device('cuda', 0)
    B = int(os.environ.get('B', 1024))
    with nat.testing() as L:
        for dtype, r in (('float32', 64), ('float32', 128), ('uint8', 128), ('uint8', 192), ('float32', 96)):
            res = Resolution(r, r)
            sim, actions, _ = bench.build_simulator(B, 64, dev, seed=1234)
            sim.renderer = renderer_from_config(HipRendererConfig(out_dtype=dtype), res=res, fov=bench.FOV)
            sim._scene_cache = None
            sim.step(actions[0])
            L.tds_raster_set_split_chunks(1)
            ref = sim.render_egocentric(res=res, fov=bench.FOV).clone()
            row = []
            for n in (1, 2, 3, 4, 6, 8, 0, 1):
                L.tds_raster_set_split_chunks(n)
                img = sim.render_egocentric(res=res, fov=bench.FOV)
                same = bool(torch.equal(img, ref))
                for _ in range(2):
                    sim.render_egocentric(res=res, fov=bench.FOV)
                torch.cuda.synchronize()
                _ops.raster_events = []
                for _ in range(12):
                    sim.render_egocentric(res=res, fov=bench.FOV)
                torch.cuda.synchronize()
                ms = [a.elapsed_time(b) for a, b in _ops.raster_events]
                _ops.raster_events = None
                row.append(f'{n}: {np.mean(ms):.3f}{"" if same else " DIFFERS"}')
            print(f'B={B} {dtype} {r}x{r}  slices -> ms   ' + '   '.join(row), flush=True)
            L.tds_raster_set_split_chunks(0)
            del sim, ref, img
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
