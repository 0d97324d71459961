#!/bin/bash
# GPU box: the default bench line (kernel profile on) for several libraries.  usage: tests/ab_kernel.sh "bench args" lib1 lib2 ...
args=$1; shift
for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo -n "[$args] $v : "
  SF_LIB_PATH=$p python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N %d value %.3e  ms/step %.3f  kernel_us %.1f  frac %.3f rebuilds %d k_half %.2f'%(d['config']['particles_per_gpu'],d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['config']['neighbor_rebuilds_in_run'],d['config']['k_half']))"
done
