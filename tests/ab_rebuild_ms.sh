#!/bin/bash
# A/B of library builds on the rebuild: loose 1 M bed (many rebuilds) and the packed headline bed: tests/ab_rebuild_ms.sh name1 name2 ...
for bed in fluidised packed; do for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo -n "$bed ${N:-1000000} $v : "
  SF_LIB_PATH=$p python bench.py --particles ${N:-1000000} --bed $bed --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); c=d['config']; print('value %.3e  kernel_us %.1f  rebuild_ms %.3f  rebuilds %d'%(d['value'],d['roofline']['mean_kernel_us'],c['neighbor_rebuild_ms'],c['neighbor_rebuilds_in_run']))"
done; done
