#!/bin/bash
# development helper (GPU box): bench.py's N > 1 path (weak + strong mode, C++ slab driver) with N ranks sharing the one
# GPU, over the stand-in for librccl (tests/c_abi/standin_rccl.cpp).  Exercises the code of the multi-GPU run end to
# end; the numbers are those of N processes time-slicing one GPU with a host-staged wire, not a scaling measurement.
# usage: tests/one_gpu_ranks.sh "2 4 8" PARTICLES [extra bench args]
ranks=$1; n=${2:-200000}; shift; shift
root=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -shared -fPIC -O2 $root/tests/c_abi/standin_rccl.cpp -o /tmp/libstandin_rccl.so || exit 1
for w in $ranks; do
  echo "== $w ranks on one GPU, $n particles"
  SF_RCCL_LIB=/tmp/libstandin_rccl.so timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29700 + w)) \
    $root/bench.py --gpus $w --one-gpu --particles $n --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile "$@" 2>&1 | grep -v "^W0\|^\*\*\*\|OMP_NUM" | tail -3 | cut -c1-1500
done
