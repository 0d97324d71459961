#!/bin/bash
# A/B of engine knobs on the 1M bed (run on the GPU box): prints ms/step and mean kernel us per variant
for v in "${@:-SF_TILE=4}"; do
  echo -n "$v : "
  env $v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.3e  ms/step %.2f  kernel_us %.1f  frac %.3f rebuilds %d'%(d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac'],d['config']['neighbor_rebuilds_in_run']))"
done
