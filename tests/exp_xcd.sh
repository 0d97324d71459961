#!/bin/bash
# GPU box: the bench kernel with equal and with weighted XCD shares (same box, same library), then the timeline of the
# weighted launch.  usage: tests/exp_xcd.sh "w0,...,w7" ["w0,...,w7" ...]
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('value %.4e  ms/step %.3f  kernel_us %.2f  frac %.4f'%(d['value'],d['ms_per_step'],d['roofline']['mean_kernel_us'],d['roofline']['frac']))"; }
echo -n "equal        : "; run
for w in "$@"; do echo -n "$w : "; SF_XCD_WEIGHTS=$w run; done
echo -n "equal again  : "; run
w=$1
SF_XCD_WEIGHTS=$w SF_STAMP_FILE=gpurun_out/stamps_w.bin SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_stamp.so \
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity >/dev/null 2>&1
python tests/micro/stamp_timeline.py gpurun_out/stamps_w.bin 2.0 | grep -A12 "per XCD:"
