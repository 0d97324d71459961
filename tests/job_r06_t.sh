#!/bin/bash
# traffic calibration (the micro-benchmark binaries travel again) + the driver's bench command on the round's final library
cd $GRAFT_REPO_ROOT
tests/calibrate_traffic.sh r06 > gpurun_out/r06_cal.log 2>&1
python bench.py --gpus 1 > gpurun_out/r06_bench_t.json 2> gpurun_out/r06_bench_t.err
tail -5 gpurun_out/r06_cal.log; tail -c 600 gpurun_out/r06_bench_t.json
