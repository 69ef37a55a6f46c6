#!/bin/bash
# A second build of the library from a PATCHED copy of the sources, for same-box A/B runs (EMX_LIB_PATH selects it):
#   tools/build_variant.sh tools/experiments/wgrad_xrows.patch build_ab/libemx_xrows.so ["-DEXTRA_FLAG ..."]
# Runs on the build box (hipcc cross-compiles gfx950).  build_ab/ is git-ignored and travels to the GPU box with the snapshot.
set -e
patch_file=$1; out=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
work=$(mktemp -d /tmp/emx_variant.XXXXXX)
trap 'rm -rf "$work"' EXIT
mkdir -p "$work/encodermap_amd" "$root/$(dirname "$out")"
cp -r "$root/encodermap_amd/csrc" "$work/encodermap_amd/csrc"
cp -r "$root/include" "$work/include"
if [ -n "$patch_file" ] && [ "$patch_file" != "-" ]; then (cd "$work" && patch -p1 < "$root/$patch_file"); fi
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -mllvm -amdgpu-kernarg-preload-count=16 -I$work/include -I$work/encodermap_amd/csrc $extra"
objs=""
for src in "$work"/encodermap_amd/csrc/*.hip; do
  obj="$work/$(basename "${src%.hip}").o"
  per_file=""
  [ "$(basename "$src")" = infer.hip ] && per_file="-fno-slp-vectorize"   # (as __graft_entry__.EXTRA_FLAGS)
  hipcc $flags $per_file -c "$src" -o "$obj" &
  objs="$objs $obj"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$root/$out"
echo "built $out"
