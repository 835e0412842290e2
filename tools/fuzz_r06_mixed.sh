#!/bin/bash
# Randomised raster parity of batches on SEVERAL maps (round 6: one device map per distinct mesh, one launch through a map set): every scene of a
# batch on Town01 or Town02 at random, the oracle renders each scene with its own town's mesh.  Fused and split forms, both output types.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r06_fuzz_raster_mixed}.log
SC=${2:-1}
: > $OUT
for cfg in "--seeds $((150*SC)) --batch 8 --agents 64 --res 64 --fov 35" "--seeds $((100*SC)) --batch 8 --agents 64 --res 128 --fov 50 --u8" "--seeds $((100*SC)) --batch 8 --agents 64 --res 128 --fov 35" \
           "--seeds $((60*SC)) --batch 8 --agents 64 --res 256 --fov 35" "--seeds $((60*SC)) --batch 8 --agents 64 --res 256 --fov 60 --u8" "--seeds $((60*SC)) --batch 8 --agents 64 --res 160 --fov 25" \
           "--seeds $((60*SC)) --batch 8 --agents 64 --res 216 --fov 80 --u8" "--seeds $((200*SC)) --batch 8 --agents 64 --res 0" "--seeds $((100*SC)) --batch 8 --agents 64 --res 0 --u8"; do
  echo "# tests/fuzz_raster.py $cfg --map mixed" >> $OUT
  python tests/fuzz_raster.py $cfg --map mixed 2>&1 | tail -1 >> $OUT
done
cat $OUT
