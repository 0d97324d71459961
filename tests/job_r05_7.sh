export AMD_LOG_LEVEL=0
( SF_HALO_DIRECT_TIMEOUT=20 timeout -k 10 1200 python -m pytest tests/test_halo_gpu.py -x -q -m gpu -k "processor_grid and (1] or 2])" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -30 | cut -c1-300 )
bash tests/trace_selfcomm_brick.sh 126000 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -75 | cut -c1-200
bash tests/ab_c5_policy.sh 2>&1 | tail -20
bash tests/ab_loose_skin.sh 2>&1 | tail -8
