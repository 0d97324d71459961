python -m pytest tests -m gpu -x -q 2>&1 | tail -4
( time python bench.py > gpurun_out/bench_r05_full.json 2> gpurun_out/bench_r05_full.err ) 2>&1 | tail -3
tail -c 3000 gpurun_out/bench_r05_full.json
tests/profile_round.sh r05 2>&1 | tail -60
