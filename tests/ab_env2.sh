#!/bin/bash
# like ab_env.sh but without per-launch event sampling: whole-run throughput only
args=$1; shift
for v in "$@"; do
  echo -n "[$args] $v : "
  env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N %d value %.3e  ms/step %.3f  us/substep %.2f rebuilds %d'%(d['config']['particles_per_gpu'],d['value'],d['ms_per_step'],d['ms_per_step']*1e3/d['config']['substeps_per_step'],d['config']['neighbor_rebuilds_in_run']))"
done
