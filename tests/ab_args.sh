#!/bin/bash
# bench sensitivity on the GPU box: tests/ab_args.sh LIBNAME "bench args" "bench args" ...   (LIBNAME default = shipped build)
v=$1; shift
p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
for a in "$@"; do
  echo -n "$v [$a] : "
  SF_LIB_PATH=$p python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N %d k_half %.2f value %.3e  ms/step %.2f  kernel_us %.1f  frac %.3f rebuilds %d'%(d['config']['particles_per_gpu'],d['config']['k_half'],d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['config']['neighbor_rebuilds_in_run']))"
done
