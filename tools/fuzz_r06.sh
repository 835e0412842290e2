#!/bin/bash
# Randomised raster parity runs of round 6 (on the GPU box, via gpurun): tools/fuzz_r05.sh plus the resolutions on both sides of round 6's split rule
# (float32 116 / 124 / 160, uint8 216 / 224; raster.hip: split_serves).  From round 5: as tools/fuzz_r04.sh -- the split form now walks the rendering grid with
# PAIRED faces (scan_faces_kernel: one fetch, projection and trim per pair; poly records in the lists; K3r expands them), the fused kernel the
# grid of single triangles -- plus resolutions on both sides of the float32 split rule (104, 120, 136, 144, 160).  As r04: weighted towards the resolutions where the
# split form K3s + K3r runs and its short path for small faces (process_small_bits) takes most faces: 4 .. 144 float32, .. 208 uint8,
# narrow and wide fields of view (faces of one or two rows), Town01 and Town02.
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/${1:-r06_fuzz_raster}.log
SC=${2:-1}
S0=${3:-0}          # first seed of every family (a second run with other scenes)
: > $OUT
for cfg in "--seeds $((200*SC)) --batch 8 --agents 64 --res 64 --fov 35" "--seeds $((100*SC)) --batch 8 --agents 64 --res 64 --fov 50 --u8 --map town02" \
           "--seeds $((100*SC)) --batch 8 --agents 64 --res 64 --fov 120" "--seeds $((60*SC)) --batch 8 --agents 64 --res 32 --fov 35 --u8" \
           "--seeds $((150*SC)) --batch 8 --agents 64 --res 128 --fov 35" "--seeds $((100*SC)) --batch 8 --agents 64 --res 128 --fov 60 --u8" \
           "--seeds $((60*SC)) --batch 8 --agents 64 --res 96 --fov 200 --map town02" "--seeds $((60*SC)) --batch 8 --agents 64 --res 144 --fov 25" \
           "--seeds $((60*SC)) --batch 8 --agents 64 --res 192 --fov 120 --u8" "--seeds $((60*SC)) --batch 8 --agents 64 --res 208 --fov 300 --u8" \
           "--seeds $((60*SC)) --batch 8 --agents 64 --res 256 --fov 35" "--seeds $((40*SC)) --batch 8 --agents 64 --res 104 --fov 35" "--seeds $((40*SC)) --batch 8 --agents 64 --res 120 --fov 50 --map town02" "--seeds $((40*SC)) --batch 8 --agents 64 --res 136 --fov 35" "--seeds $((40*SC)) --batch 8 --agents 64 --res 160 --fov 70 --u8" "--seeds $((30*SC)) --batch 8 --agents 64 --res 256 --fov 20 --u8" \
           "--seeds $((40*SC)) --batch 8 --agents 64 --res 160 --fov 35" "--seeds $((30*SC)) --batch 8 --agents 64 --res 116 --fov 35" "--seeds $((30*SC)) --batch 8 --agents 64 --res 124 --fov 50 --map town02" "--seeds $((40*SC)) --batch 8 --agents 64 --res 216 --fov 35 --u8" "--seeds $((30*SC)) --batch 8 --agents 64 --res 224 --fov 60 --u8" \
           "--seeds $((300*SC)) --batch 8 --agents 64 --res 0 --map town02" "--seeds $((150*SC)) --batch 8 --agents 64 --res 0 --map town02 --u8" "--seeds $((150*SC)) --batch 8 --agents 64 --res 0"; do
  echo "# tests/fuzz_raster.py $cfg" >> $OUT
  python tests/fuzz_raster.py $cfg --seed0 $S0 2>&1 | tail -2 >> $OUT
done
python tests/fuzz_raster_modes.py --seeds 12 --batch 4 --agents 24 2>&1 | tail -3 >> $OUT
cat $OUT
