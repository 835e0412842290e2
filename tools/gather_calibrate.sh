#!/bin/bash
# FETCH_SIZE of four known gather patterns (tools/gather_calibrate.hip); run on the GPU box
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/tools/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/_build/gather_calibrate $R/tools/gather_calibrate.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for mode in dense sector line sparse; do
  rm -rf /tmp/gc
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/gc -o gc -- $R/tools/_build/gather_calibrate $mode 2>/dev/null | grep "pieces"
  python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/gc/**/*counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'FETCH_SIZE' and 'gather_kernel' in r['Kernel_Name']]
print('    FETCH_SIZE raw (KiB units) last launch: %.4g = %.3f GiB' % (v[-1], v[-1] * 1024 / 2**30))
PY
done
