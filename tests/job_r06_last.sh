#!/bin/bash
# the round's last sources: whole GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r06_suite_last.log
(python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3) > gpurun_out/r06_smoke_last.log
python bench.py --gpus 1 > gpurun_out/r06_bench_last.json 2> gpurun_out/r06_bench_last.err
tail -n 2 gpurun_out/r06_suite_last.log; cat gpurun_out/r06_smoke_last.log; tail -c 300 gpurun_out/r06_bench_last.json
