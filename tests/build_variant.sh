#!/bin/bash
# development helper: build sedifoam_amd/libsedifoam_amd_<name>.so with extra -D flags on every source
# usage: tests/build_variant.sh NAME -DSF_FAST_MATH=0 ...   ; select with SF_LIB_PATH=<path> (see _lib.py)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
od=$root/sedifoam_amd/csrc/_obj/var_$name; mkdir -p $od
for f in $root/sedifoam_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  extra=""; [ "$b" == "sf_dem" ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"   # (like sedifoam_amd/build.py FILE_FLAGS)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DSF_VARIANT_BUILD $extra "$@" -c $f -o $od/$b.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/sedifoam_amd/libsedifoam_amd_$name.so $od/*.o
echo built libsedifoam_amd_$name.so
