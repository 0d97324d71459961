#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  SF_PERSIST=0 tests/ab_lib.sh default
  SF_PERSIST=1 tests/ab_lib.sh default pnp pst pnn
done > gpurun_out/r06_persist_arms.txt 2>&1
cat gpurun_out/r06_persist_arms.txt
