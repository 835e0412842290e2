#!/bin/bash
# Times the raster launch through every variant library of tools/build_variants.sh (and through the tree's own testing build: "base"), B = 1024 x 64:
# uint8 256 x 256, float32 256 x 256, float32 128 x 128 and 64 x 64.  Run on the GPU box.  Median of 8 launches each; one process per (library, mode).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for lib in base $(ls tools/_build/variants/libtdship_testing_*.so 2>/dev/null); do
  name=$(basename $lib .so); name=${name#libtdship_testing_}
  arg=""; [ $lib != base ] && arg="--lib $lib"
  [ $lib = base ] && arg="--lib torchdrivesim_amd/lib/libtdship_testing.so"
  line="$name:"
  for mode in "--u8" "" "--res 128" "--res 64"; do
    t=$(python3 tools/profile_raster.py --batch 1024 --iters 8 $arg $mode 2>/dev/null | grep "ms median" | sed -E 's/.* ([0-9.]+) ms median.*/\1/')
    line="$line  [${mode:-f32 256}] $t"
  done
  echo "$line"
done
