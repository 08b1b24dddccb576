#!/bin/bash
# Usage: bash tools/build_variant.sh <name> "<-D flags>" <file.hip> [more files]   -> focal_amd/lab/libfocal_hip_<name>.so
# A/B builds of single translation units for same-box comparisons (FOCAL_HIP_LIB=... selects the library at run time).
name=$1; flags=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/focal_amd/lab $root/build/lab_$name
objs=""
for f in $(ls $root/build/csrc/*.o); do
  b=$(basename $f .o); skip=0
  for v in "$@"; do [ "$b" == "$(basename $v .hip)" ] && skip=1; done
  [ $skip == 0 ] && objs="$objs $f"
done
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value $flags -c $root/focal_amd/csrc/$v -o $root/build/lab_$name/$(basename $v .hip).o || exit 1
  objs="$objs $root/build/lab_$name/$(basename $v .hip).o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/focal_amd/lab/libfocal_hip_$name.so $objs && echo built $name
