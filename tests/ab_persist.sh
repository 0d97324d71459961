#!/bin/bash
# GPU box: whole-run time per sub-step (no per-launch events) of the shipped library against variant libraries / env
# knobs at several bed sizes.  usage: tests/ab_persist.sh "N1 N2 ..." "lib[:ENV=V,...]" ...
sizes=$1; shift
for n in $sizes; do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""
    [ "$spec" != "$lib" ] && envs=$(echo ${spec#*:} | tr ',' ' ')
    p=""; [ "$lib" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$lib.so
    echo -n "N=$n $spec : "
    env SF_LIB_PATH=$p $envs python bench.py --particles $n --steps 8 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-kernel-profile $BENCH_EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N %d value %.3e  us/substep %.2f  rebuilds %d'%(d['config']['particles_per_gpu'],d['value'],1e3*d['ms_per_step']/d['config']['substeps_per_step'],d['config']['neighbor_rebuilds_in_run']))"
  done
done
