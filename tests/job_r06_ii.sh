#!/bin/bash
# k_build_neigh_quad by kernel duration: whole / ends behind the walk / ends behind the old tags + touch-first pass
cd $GRAFT_REPO_ROOT
for v in default qc1 qc2; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  for args in "--bed fluidised --particles 100000" "--bed fluidised"; do
    echo -n "$v [$args] : "
    SF_LIB_PATH=$p SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_qc "$args --no-fluidised --no-parity" 2>&1 | grep "k_build_neigh" | awk '{print $4}' | tr '\n' ' '
    echo
    rm -rf gpurun_out/kt_r06_qc
  done
done > gpurun_out/r06_quad_cut.txt 2>&1
cat gpurun_out/r06_quad_cut.txt
