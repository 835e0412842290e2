#!/bin/bash
# Instruction-count ablation of the K3 bit-plane kernel: one counter pass, one dispatch per debug mask, values listed per dispatch.
# Usage (on the GPU box): tools/pmc_ablate.sh <outdir> "<debug masks>" [profile_raster args]
set -u
OUT=$1; shift
MASKS=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "raster" --output-format csv -d $OUT/a$i -o a$i -- python3 $R/tools/profile_raster.py --iters 1 --debug $MASKS "$@" > $OUT/a$i.log 2>&1
done
python3 - "$OUT" <<'PY' > $OUT/ablate.json
import csv, glob, json, sys
from collections import defaultdict
out = defaultdict(dict)
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    per = defaultdict(float)
    for row in csv.DictReader(open(path)):
        per[(int(row['Dispatch_Id']), row['Counter_Name'])] += float(row['Counter_Value'])
    for (d, c), v in per.items():
        out[d][c] = v
print(json.dumps([dict(dispatch=d, **out[d]) for d in sorted(out)], indent=1))
PY
rm -rf $OUT/a[0-9]
cat $OUT/ablate.json
