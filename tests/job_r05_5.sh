timeout 1500 python -m pytest tests/test_halo_gpu.py -x -q -m gpu -k "processor_grid" 2>&1 | tail -15
bash tests/trace_selfcomm_brick.sh 126000 2>&1 | tail -80
