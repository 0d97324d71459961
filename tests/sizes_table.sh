#!/bin/bash
# GPU box: the default bench at several bed sizes (whole run us per sub-step, kernel us, value)
for n in "$@"; do
  python bench.py --particles $n --steps 10 --warmup 3 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('N %8d  %7.2f us/substep  kernel %7.2f us  value %.3e  frac %.3f (whole run %.3f)  rebuilds %d' % (d['config']['particles_total'], d['ms_per_step']*1e3/50, d['roofline']['mean_kernel_us'], d['value'], d['roofline']['frac'], d['roofline']['frac']*d['roofline']['mean_kernel_us']/(d['ms_per_step']*1e3/50), d['config']['neighbor_rebuilds_in_run']))"
done
