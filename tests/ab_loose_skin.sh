#!/bin/bash
# GPU box: the loose 1 M bed with wider list skins -- what a SUPERSET list of cutoff + k skins would cost to build (rebuild
# ms) and to walk (kernel us at that list length).  usage: tests/ab_loose_skin.sh
for skin in 0.25 0.5 0.75 1.0; do
  python bench.py --bed fluidised --skin $skin --steps 3 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = d['config']
print('skin %s d: k_half %.2f  kernel %.1f us  rebuilds %d in %d sub-steps  rebuild %.3f ms  ms/step %.3f' % ('$skin', c.get('k_half', 0), d['roofline']['mean_kernel_us'], c['neighbor_rebuilds_in_run'], d['steps'] * c['substeps_per_step'], c['neighbor_rebuild_ms'], d['ms_per_step']))"
done
