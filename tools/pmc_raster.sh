#!/bin/bash
# PMC passes over the K3 raster kernel (run on the GPU box via gpurun). Counters are collected in their own runs
# (never combined with tracing) as the MI355X guide prescribes.  Usage: tools/pmc_raster.sh <outdir> [profile_raster args]
set -u
OUT=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-include-regex "raster|bin_faces" --output-format csv -d $OUT/p$i -o p$i -- python3 $R/tools/profile_raster.py "$@" > $OUT/p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT "raster|bin_faces" > $OUT/summary.json; rm -rf $OUT/p[0-9]; cat $OUT/summary.json
