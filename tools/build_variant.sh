#!/bin/bash
# usage: tools/build_variant.sh <name> "<extra hipcc flags>"   ->  gpurun_out/variants/librt_<name>.so
# Builds another copy of the library with extra -D flags for same-run comparisons (RT_AMD_LIB=<path> selects it).
set -e
name=$1; flags=$2
cd "$(dirname "$0")/.."
mkdir -p variants
csrc=raytracer_amd/csrc
tmp=$(mktemp -d)
common="-std=c++17 -O3 -fPIC -ffp-contract=off -fno-math-errno -Wno-unused-function $flags"
/opt/rocm/bin/hipcc $common --offload-arch=gfx950 -c -x hip $csrc/rt_kernels.hip -o $tmp/k.o
objs="$tmp/k.o"
for f in rt_flatten host_scene host_jpeg rt_common; do
  [ -f raytracer_amd/build/$f.o ] || { echo "build the main library first"; exit 1; }
  objs="$objs raytracer_amd/build/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -lz -o variants/librt_$name.so
rm -rf $tmp
echo variants/librt_$name.so
