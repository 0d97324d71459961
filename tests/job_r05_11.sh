export AMD_LOG_LEVEL=0
bash tests/ab_gs_arms.sh 126000
echo "== direct"; BENCH_EXTRA="--decomposition bricks" SF_HALO_DIRECT=1 bash tests/trace_selfcomm.sh selfbrick_direct 126000 2>&1 | tail -1
for rep in 1 2 3; do
echo "== walled test, rep $rep"
( SF_DEBUG_HALO=1 SF_HALO_DIRECT_TIMEOUT=20 timeout -k 10 300 python -m pytest tests/test_halo_gpu.py -q -m gpu -k "processor_grid and hertz-False-2" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | grep -i "error\|passed\|failed\|sedifoam_amd\]" | head -12 | cut -c1-300 )
done
