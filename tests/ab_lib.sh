#!/bin/bash
# A/B of differently compiled libraries on the 1M bed (GPU box): tests/ab_lib.sh name1 name2 ... ("" = default build)
for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo -n "$v : "
  SF_LIB_PATH=$p python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.3e  ms/step %.2f  kernel_us %.1f  frac %.3f rebuilds %d'%(d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['config']['neighbor_rebuilds_in_run']))"
done
