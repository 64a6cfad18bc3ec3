#!/bin/bash
# build a compile-time variant of libngp_hip.so for a same-box A/B:  tools/build_variant.sh <name> <source.hip> -DMACRO=... [...]
# -> torch-ngp_amd/variants/<name>/libngp_hip.so (git-ignored, travels to the GPU box); run with NGP_HIP_LIBRARY=<that path>
set -e
name=$1; src=$2; shift 2
root=$(cd $(dirname $0)/.. && pwd)
csrc=$root/torch-ngp_amd/csrc
out=$root/torch-ngp_amd/variants/$name
mkdir -p $out
make -C $csrc -j8 > /dev/null
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $csrc/$src -o $out/$base.o
objs=""
for o in $csrc/_obj/*.o; do
  if [ $(basename $o) = $base.o ]; then objs="$objs $out/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libngp_hip.so $objs
echo $out/libngp_hip.so
