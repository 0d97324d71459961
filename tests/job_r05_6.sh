export SF_HALO_DIRECT_TIMEOUT=10
export AMD_LOG_LEVEL=0
( SF_HALO_SELF_COMM=1 SF_HALO_DIRECT=2 SF_DEBUG_HALO=1 timeout -s KILL 150 python bench.py --slab-driver --particles 126000 --steps 4 --warmup 1 --no-cpu-baseline --no-coupled --no-configs --no-kernel-profile --decomposition bricks 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -25 | cut -c1-400 )
echo "== rc $?"
( timeout -s KILL 400 python -m pytest tests/test_halo_gpu.py -x -q -m gpu -k "processor_grid and hertz-True-2" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -60 | cut -c1-300 )
