#!/usr/bin/env python3
"""
BASELINE.json's single-GPU configurations other than the headline one, one JSON line each -- the same code bench.py runs for its `configs`
array (bench.other_configs), callable on its own so that rocprofv3 can trace one configuration at a time:
  config2   B=256 x A=64: KinematicBicycle.step + render_egocentric 256x256 + compute_collision(iou)
  config3   + compute_offroad + compute_wrong_way (Town01 lane map)
  config5   config 3's forward without wrong-way, state and action requiring grad, then backward through kinematics, IoU, off-road and the
            rasteriser (loss = <image, fixed random weights> + sum(collision) + sum(offroad))
    python tools/bench_configs.py [--steps 20] [--warmup 3] [--only config2]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--only', default=None, help='config2, config3 or config5')
    ap.add_argument('--overlap', default='reserved', choices=('reserved', 'stream', 'off'), help='configs 2 / 3: where the metrics run (bench.py --overlap)')
    ap.add_argument('--overlap-diff', default='off', choices=('reserved', 'stream', 'off'), help='config 5: the same for the differentiable step')
    args = ap.parse_args()
    for entry in bench.other_configs(torch.device('cuda', 0), args.steps, args.warmup, only=args.only, overlap=args.overlap, overlap_diff=args.overlap_diff):
        print(json.dumps(entry), flush=True)


if __name__ == '__main__':
    main()
