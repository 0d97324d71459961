#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --particles 100000 --no-fluidised --no-parity" "SF_LPA=1" "SF_LPA=2" "SF_LPA=4"
tests/ab_env.sh "--bed fluidised --particles 300000 --no-fluidised --no-parity" "SF_LPA=1" "SF_LPA=2"
done
