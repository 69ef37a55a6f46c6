#!/bin/bash
# A second build of the library that differs in csrc/sketchmap.hip only (extra compiler flags), for same-box A/B runs and stamp builds:
#   tools/build_sketchmap_variant.sh build_ab/libemx_in_stamp.so "-DIN_STAMP"
# Links the other sources' objects of the last __graft_entry__.build() as they stand.  EMX_LIB_PATH selects the variant.
set -e
out=$1; extra=$2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$(dirname "$out")"
obj=$(mktemp /tmp/emx_sketchmap.XXXXXX.o)
trap 'rm -f "$obj"' EXIT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=16 -I"$root/include" -I"$root/encodermap_amd/csrc" $extra \
  -c "$root/encodermap_amd/csrc/sketchmap.hip" -o "$obj"
others=$(ls "$root"/build/obj/*.o | grep -v '/sketchmap.o$')
hipcc --offload-arch=gfx950 -shared -fPIC $others "$obj" -o "$root/$out"
echo "built $out"
