#!/bin/bash
# pricing arms of the candidate walk (wrong lists; phase cycles only): one 16-byte load per record; every lane of a cell reads the same records
cd $GRAFT_REPO_ROOT
B="--steps 3 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile --no-fluidised --no-parity"
for v in bph bw1 bw2 bph; do for args in "--bed fluidised --particles 100000" "--bed fluidised"; do
  echo "== $v $args"
  SF_BUILD_LPA=1 SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so timeout 120 python bench.py $B $args 2>&1 >/dev/null | grep "k_build_neigh\|rror" | cut -c1-300
done; done > gpurun_out/r06_walk_arms.txt 2>&1
cat gpurun_out/r06_walk_arms.txt
