#!/bin/bash
# round 6, re-entry: full GPU suite on HEAD, the driver's bench command, rebuild traces of the two loose beds
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/r06_suite_g.log
python bench.py --gpus 1 > gpurun_out/r06_bench_g.json 2> gpurun_out/r06_bench_g.err
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3g "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3g.txt 2>&1
tests/trace_rebuild.sh r06_l1mg "--bed fluidised --no-fluidised --no-parity" > gpurun_out/r06_trace_l1mg.txt 2>&1
rm -rf gpurun_out/kt_r06_c3g gpurun_out/kt_r06_l1mg
tail -3 gpurun_out/r06_suite_g.log; tail -c 1500 gpurun_out/r06_bench_g.json
