#!/bin/bash
# GPU box: record the workgroup timeline of one k_substep launch (variant library `stamp`) and print its analysis
# usage: tests/exp_stamp.sh [bench args]
mkdir -p gpurun_out
SF_STAMP_FILE=gpurun_out/stamps.bin SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_stamp.so \
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity "$@" 2>&1 | tail -1 | cut -c1-400
python tests/micro/stamp_timeline.py gpurun_out/stamps.bin 2.0
