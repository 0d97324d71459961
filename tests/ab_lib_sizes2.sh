#!/bin/bash
# A/B of library builds over bed sizes and both bed kinds (headline kernel only): tests/ab_lib_sizes2.sh "SIZES" name1 name2 ...
sizes=$1; shift
for bed in packed fluidised; do for n in $sizes; do for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo -n "$bed $n $v : "
  SF_LIB_PATH=$p python bench.py --particles $n --bed $bed --steps 8 --warmup 3 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.3e  kernel_us %.2f  frac %.3f rebuilds %d'%(d['value'],d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['config']['neighbor_rebuilds_in_run']))"
done; done; done
