#!/bin/bash
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
tests/ab_env.sh "--bed fluidised --no-fluidised --no-parity" "SF_LPA=1" "SF_LPA=2"
tests/ab_env.sh "--bed fluidised --particles 600000 --no-fluidised --no-parity" "SF_LPA=1" "SF_LPA=2"
tests/ab_env.sh "--bed fluidised --particles 200000 --no-fluidised --no-parity" "SF_LPA=1" "SF_LPA=2"
tests/ab_env.sh "--bed fluidised --particles 50000 --no-fluidised --no-parity" "SF_LPA=1" "SF_LPA=2" "SF_LPA=4"
done
