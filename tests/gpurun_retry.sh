#!/bin/bash
# gpurun with retries while no box / slot is free (exit code 3: nothing charged).  usage: tests/gpurun_retry.sh <timeout s> '<command>'
t=$1; shift
for k in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
