#!/bin/bash
# k_build_neigh phase marks: 100 k loose bed with 1 / 4 lanes per atom, 1 M loose, 1 M packed
cd $GRAFT_REPO_ROOT
B="--steps 4 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile --no-fluidised --no-parity"
for lpa in 1 4; do for args in "--bed fluidised --particles 100000"; do
  echo "== $args LPA $lpa"
  SF_BUILD_LPA=$lpa SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_bph.so python bench.py $B $args 2>&1 >/dev/null | grep "k_build_neigh"
done; done > gpurun_out/r06_build_phase2.txt 2>&1
for args in "--bed fluidised" ""; do
  echo "== $args"
  SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_bph.so python bench.py $B $args 2>&1 >/dev/null | grep "k_build_neigh"
done >> gpurun_out/r06_build_phase2.txt 2>&1
cat gpurun_out/r06_build_phase2.txt
