#!/bin/bash
# lanes per atom, h-major lane mapping: bit-identity test, phase marks, whole-run A/B
cd $GRAFT_REPO_ROOT
(python -m pytest tests/test_dem_gpu.py -x -q -k "lanes_per_atom or variants_agree or periodic_images" 2>&1 | tail -5) > gpurun_out/r06_suite_n.log
B="--steps 4 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile --no-fluidised --no-parity"
for lpa in 1 2 4; do for args in "--bed fluidised --particles 100000" "--bed fluidised"; do
  echo "== $args LPA $lpa"
  SF_BUILD_LPA=$lpa SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_bph.so python bench.py $B $args 2>&1 >/dev/null | grep "k_build_neigh"
done; done > gpurun_out/r06_build_phase3.txt 2>&1
{
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2" "SF_BUILD_LPA=4"
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_BUILD_LPA=1" "SF_BUILD_LPA=2" "SF_BUILD_LPA=4"
done
} > gpurun_out/r06_build_lpa_ab2.txt 2>&1
tail -3 gpurun_out/r06_suite_n.log; cat gpurun_out/r06_build_phase3.txt gpurun_out/r06_build_lpa_ab2.txt
