#!/bin/bash
# PMC passes over tools/valu_calibrate (run on the GPU box via gpurun): what SQ_ACTIVE_INST_VALU reads per instruction
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "3 64" "6 64" "3 16"; do
  rm -rf /tmp/vc
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d /tmp/vc -o vc -- $R/tools/_build/valu_calibrate $cfg 2>/dev/null | grep "workgroups per CU"
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/vc/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'add_chain' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
c = {k: v[-1] for k, v in acc.items()}          # the second (long) launch
print('   ', {k: f'{v:.4g}' for k, v in c.items()})
print('    SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = %.3f   SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = %.2f   ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES = %.2f (of 32 SIMDs per SE?)   SQ_WAVE_CYCLES / SQ_BUSY_CYCLES = %.2f' % (
    c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU'], c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU'], 4 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_BUSY_CYCLES'], c['SQ_WAVE_CYCLES'] / c['SQ_BUSY_CYCLES']))
PY
done
