#!/bin/bash
# launches and GPU time per differentiable step (BASELINE config 5), small kernels apart; run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5 -o c5 -- python3 $R/tools/bench_configs.py --only config5 --steps 20 --warmup 3 > /tmp/c5.log 2>&1
grep "^{" /tmp/c5.log | python3 -c "
import json, sys
for l in sys.stdin:
    for c in json.loads(l) if l.strip().startswith('[') else [json.loads(l)]:
        print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.items() if k in ('config', 'ms_per_step', 'ms_per_step_without_loss_probe', 'dominant_kernel_ms', 'raster_backward_kernel_ms', 'loss_probe_ms')})"
python3 - <<'PY'
import csv, glob
rows = list(csv.reader(open(glob.glob('/tmp/c5/**/*kernel_stats.csv', recursive=True)[0])))[1:]
steps = 23
small = [r for r in rows if float(r[3]) < 20e3]
for r in sorted(small, key=lambda r: -int(r[1]))[:22]:
    print('   %5.1f per step  %6.1f us  %s' % (int(r[1]) / steps, float(r[3]) / 1e3, r[0][:150]))
print('launches per step %.1f, of them under 20 us: %.1f (%.3f ms per step); all kernels %.3f ms per step' % (
    sum(int(r[1]) for r in rows) / steps, sum(int(r[1]) for r in small) / steps, sum(float(r[2]) for r in small) / 1e6 / steps, sum(float(r[2]) for r in rows) / 1e6 / steps))
PY
