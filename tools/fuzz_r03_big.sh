#!/bin/bash
# The LARGER randomised raster parity run of round 3 (four times the seeds of tools/fuzz_r03.sh: a superset of its images) (on the GPU box, via gpurun): the persistent fused kernel (256 x 256 and above), the split form
# K3s + K3r (up to 144 x 144 float32 / 208 x 208 uint8), both output types, and the third family of VERDICT r2: resolutions 4 .. 60,
# fields of view 5 .. 200 m, on Town02.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/r03_fuzz_raster_big.log
: > $OUT
for cfg in "--seeds 600 --batch 8 --agents 64 --res 256 --fov 35" "--seeds 240 --batch 8 --agents 64 --res 256 --fov 20 --u8" "--seeds 240 --batch 8 --agents 64 --res 256 --fov 80" \
           "--seeds 600 --batch 8 --agents 64 --res 128 --fov 35" "--seeds 400 --batch 8 --agents 64 --res 128 --fov 60 --u8" "--seeds 800 --batch 8 --agents 64 --res 64 --fov 35" \
           "--seeds 400 --batch 8 --agents 64 --res 64 --fov 50 --u8 --map town02" "--seeds 240 --batch 8 --agents 64 --res 192 --fov 120 --u8" "--seeds 240 --batch 8 --agents 64 --res 144 --fov 25" \
           "--seeds 160 --batch 4 --agents 64 --res 320 --fov 50" "--seeds 96 --batch 4 --agents 32 --res 512 --fov 60" \
           "--seeds 2400 --batch 8 --agents 64 --res 0 --map town02" "--seeds 1200 --batch 8 --agents 64 --res 0 --map town02 --u8" "--seeds 1200 --batch 8 --agents 64 --res 0"; do
  echo "# tests/fuzz_raster.py $cfg" >> $OUT
  python tests/fuzz_raster.py $cfg 2>&1 | tail -2 >> $OUT
done
python tests/fuzz_raster_modes.py --seeds 48 --batch 4 --agents 24 2>&1 | tail -3 >> $OUT
cat $OUT
