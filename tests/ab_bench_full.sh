#!/bin/bash
# A/B of library builds on everything bench.py times (headline, fluidised side run, configs C2 / C3 / C5): tests/ab_bench_full.sh name1 name2 ...
for v in "$@"; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  echo "== $v"
  SF_LIB_PATH=$p python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('  headline kernel_us %.1f frac %.3f frac_2m %s value %.3e' % (r['mean_kernel_us'], r['frac'], r.get('frac_2m'), d['value']))
f = d.get('fluidised_bed') or {}
print('  fluidised:', {k: f.get(k) for k in ('value', 'kernel_us', 'roofline_frac', 'roofline_frac_whole_run', 'rebuilds') if k in f})
for k, c in (d.get('configs') or {}).items():
    print('  ', k, {q: c.get(q) for q in ('value', 'kernel_us', 'roofline_frac', 'coupled_steps_per_s') if q in c})
"
done
