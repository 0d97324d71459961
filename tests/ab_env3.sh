#!/bin/bash
# GPU box: the default bench kernel under several environments, interleaved REPS times (default 3); prints every run.
# usage: [REPS=n] tests/ab_env3.sh "ENV=V ENV=V" ...   ("-" = none)
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity $BENCH_EXTRA 2>gpurun_out/ab_env3.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('kernel_us %.2f ms/step %.3f'%(d['roofline']['mean_kernel_us'],d['ms_per_step']))"; grep "sedifoam_amd\]" gpurun_out/ab_env3.err | tail -${DBG_LINES:-0}; }
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-3}); do
  for e in "$@"; do
    echo -n "rep $rep [$e] "
    if [ "$e" == "-" ]; then run; else env $e bash -c "$(declare -f run); run"; fi
  done
done
