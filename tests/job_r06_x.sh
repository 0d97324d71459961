#!/bin/bash
# final sources of the round: the whole GPU suite twice back to back, then the driver's bench command
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r06_suite_x1.log
(python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r06_suite_x2.log
python bench.py --gpus 1 > gpurun_out/r06_bench_x.json 2> gpurun_out/r06_bench_x.err
tail -2 gpurun_out/r06_suite_x1.log gpurun_out/r06_suite_x2.log; tail -c 300 gpurun_out/r06_bench_x.json
