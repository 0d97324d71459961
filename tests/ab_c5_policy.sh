#!/bin/bash
# GPU box: the both-arms C5 kernel (500 k polydisperse, cohesive + lubricate/poly) under each non-temporal policy of the row
# streams x one / two history copies per contact, interleaved twice.  usage: tests/ab_c5_policy.sh [N]
n=${1:-500000}
for rep in 1 2; do
  for copies in 1 2; do
    for pol in 0 3 1 2; do
      echo -n "copies $copies policy $pol : "
      SF_HIST_COPIES=$copies SF_NT_POLICY=$pol python tests/micro/poly_bench.py $n all 6 2>/dev/null | tail -1
    done
  done
done
