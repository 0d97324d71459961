#!/bin/bash
# state after the list-build work (in place, LDS parking, linear walk at 5 waves, filtered F_MAXNEIGH atomic, 16 k scan tiles):
# the whole GPU suite, traces, the one-XCD barrier probe
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r06_suite_p.log
SF_TRACE_MIN_NS=12000 tests/trace_rebuild.sh r06_c3p "--bed fluidised --particles 100000 --no-fluidised --no-parity" > gpurun_out/r06_trace_c3p.txt 2>&1
rm -rf gpurun_out/kt_r06_c3p
(cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o xcd_barrier $GRAFT_REPO_ROOT/tests/micro/xcd_barrier.hip && timeout 120 ./xcd_barrier 2000) > gpurun_out/r06_xcd_barrier.txt 2>&1
tail -3 gpurun_out/r06_suite_p.log; cat gpurun_out/r06_trace_c3p.txt gpurun_out/r06_xcd_barrier.txt
