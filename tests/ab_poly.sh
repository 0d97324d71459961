#!/bin/bash
# A/B of library builds on the 500k polydisperse cohesive+lubricate bed: tests/ab_poly.sh name1 name2 ...
for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo -n "$v : "; SF_LIB_PATH=$p python tests/micro/poly_bench.py 500000 all 2>/dev/null | tail -1
done
