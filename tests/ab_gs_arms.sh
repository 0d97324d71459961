#!/bin/bash
# GPU box: what each part of the ghost-slot hand-off costs -- variant libraries (tests/build_variant_dem.sh gs_<arm> -DSF_GS_EXP_...)
# on the self-exchanging 126 k brick.  The arms are NOT correct hand-offs (pricing only).
n=${1:-126000}
export BENCH_EXTRA="--decomposition bricks"
for arm in "" nodone nodone_plainst nogate bare; do
  lib=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd${arm:+_gs_$arm}.so
  [ -f $lib ] || continue
  echo "== arm ${arm:-shipped}"
  SF_LIB_PATH=$lib SF_HALO_DIRECT=2 bash tests/trace_selfcomm.sh selfbrick_arm_${arm:-shipped} $n 2>&1 | tail -1 | cut -c1-160
done
