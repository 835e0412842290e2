#!/bin/bash
# per-kernel times of the raster path at a resolution (rocprofv3 --kernel-trace --stats over tools/profile_raster.py); run on the GPU box
# usage: tools/kernel_split.sh <label> <profile_raster args...>
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$L
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$L -o ks -- python3 $R/tools/profile_raster.py --batch 1024 --iters 6 "$@" > /tmp/ks_$L.log 2>&1
F=$(find /tmp/ks_$L -name "*kernel_stats.csv" | head -1)
echo "== $L: $@"; grep "median" /tmp/ks_$L.log | tail -1
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[1:6]:
    print('   %-70s calls %4s avg %9.3f us min %9.3f' % (r[0][:70], r[1], float(r[3]) / 1e3, float(r[5]) / 1e3))
PY
