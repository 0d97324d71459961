#!/bin/bash
# GPU box: the brick driver's exchange on one rank that exchanges with itself: over RCCL, by direct ghost writes into receive
# areas (one unpack kernel per exchange) and by ghost slots (no kernel between two sub-step kernels)
n=${1:-126000}
export BENCH_EXTRA="--decomposition bricks"
echo "== RCCL =="; SF_HALO_DIRECT=0 bash tests/trace_selfcomm.sh selfbrick_rccl $n 2>&1 | tail -22
echo "== direct =="; SF_HALO_DIRECT=1 bash tests/trace_selfcomm.sh selfbrick_direct $n 2>&1 | tail -22
echo "== ghost slots =="; SF_HALO_DIRECT=2 bash tests/trace_selfcomm.sh selfbrick_slots $n 2>&1 | tail -22
