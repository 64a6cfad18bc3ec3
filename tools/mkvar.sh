#!/bin/bash
# build a variant of libngp_hip.so from an ARBITRARY source file standing in for one object:  tools/mkvar.sh <name> <object base, e.g. gridencoder> <source path> [-DMACRO=...]
# -> torch-ngp_amd/variants/<name>/libngp_hip.so (git-ignored, travels to the GPU box); run with NGP_HIP_LIBRARY=<that path>
set -e
name=$1; base=$2; src=$3; shift 3
root=$(cd $(dirname $0)/.. && pwd)
csrc=$root/torch-ngp_amd/csrc
out=$root/torch-ngp_amd/variants/$name
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I$csrc "$@" -x hip -c $src -o $out/$base.o
objs=""
for o in $csrc/_obj/*.o; do
  if [ $(basename $o) = $base.o ]; then objs="$objs $out/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libngp_hip.so $objs
echo built $name
