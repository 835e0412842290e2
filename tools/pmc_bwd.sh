#!/bin/bash
# HBM bytes the raster backward (index-slice kernel) fetches per launch at B = 256 (FETCH_SIZE / WRITE_SIZE passes; run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pb
  rocprofv3 --pmc $set --kernel-include-regex "bwd_idx" --output-format csv -d /tmp/pb -o pb -- python3 $R/tools/bench_configs.py --only ${1:-config5} --steps 4 --warmup 1 > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pb/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print('%s mean %.4g over %d launches%s' % (k, sum(v) / len(v), len(v), ' = %.3f GB (KiB units; FETCH x 2 on gfx950: %.3f GB)' % (sum(v) / len(v) * 1024 / 1e9, 2 * sum(v) / len(v) * 1024 / 1e9) if 'SIZE' in k else ''))
PY
done
