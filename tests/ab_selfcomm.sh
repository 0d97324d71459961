#!/bin/bash
# development helper (GPU box): whole-run period of the halo loop with self-communication over RCCL
# usage: tests/ab_selfcomm.sh PARTICLES "ENV=.. ENV2=.." ...
n=$1; shift
for v in "$@"; do
  echo -n "[$n] $v : "
  env SF_HALO_SELF_COMM=1 $v python bench.py --slab-driver --particles $n --steps 10 --warmup 3 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N %d value %.3e  us/substep %.2f rebuilds %d'%(d['config']['particles_per_gpu'],d['value'],d['ms_per_step']*1e3/d['config']['substeps_per_step'],d['config']['neighbor_rebuilds_in_run']))"
done
