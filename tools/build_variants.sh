#!/bin/bash
# Variant builds of libtdship_testing.so for parameter sweeps of raster.hip's compile-time knobs (TDS_FCHUNK rows per row item, TDS_VCHUNK / TDS_HCHUNK
# rows per item of the exact edge walks, ...): only raster.hip is recompiled, the other objects are the testing build's.
#   tools/build_variants.sh name1="-DTDS_FCHUNK=3" name2="-DTDS_FCHUNK=6 -DTDS_VCHUNK=8" ...      -> tools/_build/variants/libtdship_testing_<name>.so
# tools/profile_raster.py --lib <that> runs the raster launch through it.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/torchdrivesim_amd/csrc
O=$R/tools/_build/variants
mkdir -p $O
make -C $C -j8 ../lib/libtdship_testing.so > /dev/null
pids=()
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-function -DTDS_TESTING \
      -mllvm -amdgpu-sched-strategy=iterative-minreg $flags -c $C/raster.hip -o $O/raster_$name.o 2> $O/raster_$name.err && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libtdship_testing_$name.so $(ls $C/_objt/*.o | grep -v /raster.o) $O/raster_$name.o && \
    rm -f $O/raster_$name.o && echo "built $name ($flags)" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
