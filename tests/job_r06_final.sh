#!/bin/bash
# profile round on the round's final sources (calibration, bench line, kernel trace + stats, PMC passes), then the driver's command
cd $GRAFT_REPO_ROOT
tests/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
python tests/pmc_summarize.py r06 "round-6 final sources (list build reworked; sub-step kernel unchanged since round 5's half-wave gather)" > gpurun_out/r06_pmc_summarize.log 2>&1
python bench.py --gpus 1 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
tail -5 gpurun_out/r06_profile_round.log | cut -c1-300; tail -c 400 gpurun_out/r06_bench_final.json
