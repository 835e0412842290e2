#!/bin/bash
# Where does the split form (K3s + K3r) of the bit-plane path stop paying against the fused kernel?  One consistent sweep on ONE build and box
# (the comments beside SPLIT_MAX_RES_* quoted figures of different rounds): ms per render call at B = 1024 x 64, median of 6,
# testing build, fused (debug flag 8192) against split (16384).  Run on the GPU box.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for mode in "f32" "u8"; do
  list="${F32_LIST:-96 112 120 128 136 144 160 176 192}"; extra=""
  [ $mode = u8 ] && { list="${U8_LIST:-128 160 176 192 208 224 256}"; extra="--u8"; }
  for res in $list; do
    out=$(python3 tools/profile_raster.py --batch 1024 --iters 6 --res $res --debug 8192 16384 $extra 2>/dev/null | grep "ms median")
    f=$(echo "$out" | grep "debug= 8192" | sed -E 's/.* ([0-9.]+) ms median.*/\1/')
    s=$(echo "$out" | grep "debug=16384" | sed -E 's/.* ([0-9.]+) ms median.*/\1/')
    echo "$mode ${res}x${res}: fused $f ms, split $s ms"
  done
done
