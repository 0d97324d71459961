#!/bin/bash
# GPU box: kernel time of the default bench for "lib[:ENV=V,...]" specs, interleaved twice
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity $BENCH_EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.4e  ms/step %.3f  kernel_us %.2f  frac %.4f'%(d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac']))"; }
for rep in 1 2; do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""
    [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
    p=""; [ "$lib" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$lib.so
    echo -n "[$spec] : "
    env SF_LIB_PATH=$p $envs bash -c "$(declare -f run); run"
  done
done
