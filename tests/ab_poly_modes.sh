#!/bin/bash
# A/B of library builds on the 500 k polydisperse bed, every style mix: tests/ab_poly_modes.sh name1 name2 ...
for m in hertz cohesive lub all; do for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo -n "$v : "; SF_LIB_PATH=$p python tests/micro/poly_bench.py ${N:-500000} $m 2>/dev/null | tail -1
done; done
