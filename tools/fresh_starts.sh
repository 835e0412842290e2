#!/bin/bash
# VERDICT r3 items 1 + 2: N fresh-process starts of the headline bench; one line each: ms/step, raster launch (HIP events over the timed
# region), what the ring probe saw, and the reference-shaped path without out= (images from the pool).  Run on the GPU box via gpurun.
N=${1:-20}
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $N); do
  python3 $R/bench.py --no-configs --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
l = json.loads(sys.stdin.read()); r = l['roofline']; d = l.get('default_path', {})
print('start $i: ms/step %.3f raster %.3f | ring probe launch_ms %s fill %.2f fast %s kept %s | without out=: ms/step %.3f raster avg %.3f min %.3f max %.3f' % (
    l['ms_per_step'], r['avg_launch_ms'], [round(x, 3) for x in r['ring_probe']['launch_ms']], r['ring_probe']['fill_ms'], r['ring_probe']['fast'], r['ring_probe']['kept'],
    d.get('ms_per_step', 0), d.get('avg_launch_ms', 0), d.get('min_launch_ms', 0), d.get('max_launch_ms', 0)))"
done
