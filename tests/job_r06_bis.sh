#!/bin/bash
# bisecting the one failure of the last full run: test_rccl_transport_self_images[True-rccl], five times per arm
cd $GRAFT_REPO_ROOT
for arm in "default" "SF_FLAG_SPIN=0" "nokr"; do
  for k in 1 2 3 4 5; do
    if [ "$arm" == "nokr" ]; then export SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_nokr.so; e=""; else unset SF_LIB_PATH; e=$arm; fi
    [ "$e" == "default" ] && e=""
    r=$(env $e python -m pytest "tests/test_halo_gpu.py::test_rccl_transport_self_images" -x -q 2>&1 | tail -1)
    echo "$arm run $k: $r"
  done
done
