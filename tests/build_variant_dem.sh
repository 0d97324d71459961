#!/bin/bash
# development helper: libsedifoam_amd_<name>.so = the shipped objects with sf_dem.hip (the sub-step kernel) recompiled with
# extra flags (compiler experiments).  usage: tests/build_variant_dem.sh NAME -mllvm -some-flag ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
od=$root/sedifoam_amd/csrc/_obj/var_$name; mkdir -p $od
cp $root/sedifoam_amd/csrc/_obj/*.o $od/
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c $root/sedifoam_amd/csrc/sf_dem.hip -o $od/sf_dem.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/sedifoam_amd/libsedifoam_amd_$name.so $od/*.o
echo built libsedifoam_amd_$name.so
