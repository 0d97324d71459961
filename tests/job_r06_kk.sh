#!/bin/bash
cd $GRAFT_REPO_ROOT
(python -m pytest tests -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r06_suite_kk.log
tail -n 2 gpurun_out/r06_suite_kk.log
