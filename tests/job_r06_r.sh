#!/bin/bash
# k_build_neigh by kernel duration (rocprofv3 trace): whole / walk only / no walk / a third of the rows, shadow on and off
cd $GRAFT_REPO_ROOT
for v in default bc1 bc2 bc3; do for sh in 1 0; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  for args in "--bed fluidised --particles 100000" "--bed fluidised"; do
    echo -n "$v shadow=$sh [$args] : "
    SF_LIB_PATH=$p SF_BUILD_SHADOW=$sh SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_bc "$args --no-fluidised --no-parity" 2>&1 | grep "k_build_neigh" | awk '{print $4}' | tr '\n' ' '
    echo
    rm -rf gpurun_out/kt_r06_bc
  done
done; done > gpurun_out/r06_build_cut.txt 2>&1
cat gpurun_out/r06_build_cut.txt
