#!/bin/bash
# GPU box: the brick driver's exchange on one rank that exchanges with itself, over RCCL and by direct ghost writes
n=${1:-126000}
export BENCH_EXTRA="--decomposition bricks"
echo "== RCCL =="; SF_HALO_DIRECT=0 bash tests/trace_selfcomm.sh selfbrick_rccl $n 2>&1 | tail -22
echo "== direct =="; SF_HALO_DIRECT=1 bash tests/trace_selfcomm.sh selfbrick_direct $n 2>&1 | tail -22
