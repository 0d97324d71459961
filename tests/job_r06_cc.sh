#!/bin/bash
# confidence: the whole GPU suite three times back to back on the final library
cd $GRAFT_REPO_ROOT
for k in 1 2 3; do (python -m pytest tests -m gpu -x -q 2>&1 | tail -2) ; done > gpurun_out/r06_suite_cc.log 2>&1
cat gpurun_out/r06_suite_cc.log
