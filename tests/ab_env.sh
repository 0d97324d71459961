#!/bin/bash
# A/B of engine env knobs with bench arguments (GPU box): tests/ab_env.sh "bench args" VAR=1 "VAR=2 OTHER=3" ...
args=$1; shift
for v in "$@"; do
  echo -n "[$args] $v : "
  env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N %d value %.3e  ms/step %.3f  kernel_us %.1f  frac %.3f rebuilds %d'%(d['config']['particles_per_gpu'],d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['config']['neighbor_rebuilds_in_run']))"
done
