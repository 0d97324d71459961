#!/bin/bash
# GPU box: the ghost-slot transport on one self-exchanging brick under each memory type of the areas x load flavour
n=${1:-126000}
export BENCH_EXTRA="--decomposition bricks"
for area in fine coarse; do for loads in system plain; do
  echo "== area $area loads $loads"
  SF_GS_AREA=$area SF_GS_LOADS=$loads SF_HALO_DIRECT=2 bash tests/trace_selfcomm.sh selfbrick_slots_${area}_$loads $n 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -4 | cut -c1-160
done; done
echo "== direct (receive areas + unpack kernel)"; SF_HALO_DIRECT=1 bash tests/trace_selfcomm.sh selfbrick_direct $n 2>&1 | tail -4 | cut -c1-160
