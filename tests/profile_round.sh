#!/bin/bash
# one GPU call: calibration, kernel trace + stats, PMC passes, plain bench line  (tag r03)
tag=$1
tests/calibrate_traffic.sh $tag | tail -20
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$tag.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fluidised --no-coupled --no-configs > $GRAFT_REPO_ROOT/gpurun_out/kt_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
tests/pmc.sh $tag "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_READ_sum" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VALU" | tail -30
tail -c 600 gpurun_out/bench_$tag.json
