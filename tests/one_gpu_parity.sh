#!/bin/bash
# development helper (GPU box): the self-proving N > 1 line of bench.py with N ranks sharing the one GPU over the stand-in
# wire: prints value, decomposition and the `parity` object (decomposed against single-domain).  usage: tests/one_gpu_parity.sh "2 4 8" PARTICLES
ranks=$1; n=${2:-200000}; shift; shift
root=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -shared -fPIC -O2 $root/tests/c_abi/standin_rccl.cpp -o /tmp/libstandin_rccl.so || exit 1
for w in $ranks; do
  SF_RCCL_LIB=/tmp/libstandin_rccl.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29700 + w)) \
    $root/bench.py --gpus $w --one-gpu --particles $n --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile "$@" 2>gpurun_out/ogp_$w.err | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line)
        print('N', d['n_gpus'], 'value %.3e' % d['value'], 'scaling', d['scaling'], '|', d['config']['decomposition'][:60])
        print('   halo:', d['config'].get('halo', '')[:60], '|', d['config'].get('halo_note'), '| exchange us', d['config'].get('halo_exchange_us_per_substep'))
        print('   parity:', json.dumps(d.get('parity')))
"
  echo "   rc=$? $(grep -c Traceback gpurun_out/ogp_$w.err) tracebacks"; grep -A8 Traceback gpurun_out/ogp_$w.err | tail -12
done
