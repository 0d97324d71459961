#!/bin/bash
# GPU box: two libraries at several bed sizes / beds, interleaved (whole run ms per step)
# usage: tests/ab_lib_sizes.sh LIB "bench args" ["bench args" ...]
lib=$1; shift
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity $1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms/step %.4f  kernel_us %.2f'%(d['ms_per_step'],d['roofline']['mean_kernel_us']))"; }
for a in "$@"; do
  for rep in 1 2; do
    for l in default $lib; do
      p=""; [ "$l" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$l.so
      echo -n "[$a] $l : "; SF_LIB_PATH=$p bash -c "$(declare -f run); run '$a'"
    done
  done
done
