for bedargs in "--bed fluidised" "--jitter 0.15 --spacing 1.0"; do
  for tf in 0 1 ""; do
    echo "== $bedargs SF_TOUCH_FIRST=$tf"
    if [ -z "$tf" ]; then e="SF_DEBUG_HIST=1"; else e="SF_DEBUG_HIST=1 SF_TOUCH_FIRST=$tf"; fi
    env $e python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised $bedargs 2> gpurun_out/dbg.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.3e  ms/step %.3f  kernel_us %.1f rebuilds %d'%(d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['config']['neighbor_rebuilds_in_run']))"
    grep sedifoam_amd gpurun_out/dbg.err | sort | uniq -c | sort -rn | head -8
  done
done
