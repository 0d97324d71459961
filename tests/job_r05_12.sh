export AMD_LOG_LEVEL=0
export BENCH_EXTRA="--decomposition bricks"
echo "== ghost slots"; SF_HALO_DIRECT=2 bash tests/trace_selfcomm.sh selfbrick_slots 126000 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -3 | cut -c1-160
echo "== direct"; SF_HALO_DIRECT=1 bash tests/trace_selfcomm.sh selfbrick_direct 126000 2>&1 | tail -1
echo "== walled test"
( SF_DEBUG_HALO=1 timeout -k 10 300 python -m pytest tests/test_halo_gpu.py -q -m gpu -x -k "processor_grid and hertz-False-2" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -45 | cut -c1-250 )
echo "== tests mode 2"
( SF_HALO_DIRECT_TIMEOUT=20 timeout -k 10 900 python -m pytest tests/test_halo_gpu.py -q -m gpu -k "processor_grid and 2]" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -6 | cut -c1-250 )
