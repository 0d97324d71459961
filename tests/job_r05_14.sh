export AMD_LOG_LEVEL=0
for rep in 1 2 3 4 5 6 7 8; do
echo "== walled test rep $rep"
( SF_DEBUG_HALO=1 SF_TEST_TIMEOUT=15 timeout -k 10 200 python -m pytest tests/test_halo_gpu.py -q -m gpu -x -k "processor_grid and hertz-False-2" 2>&1 | grep "ran out\|passed\|failed" | cut -c1-500 )
done
echo "== tests mode 2 (twice)"
for rep in 1 2; do
( SF_TEST_TIMEOUT=20 timeout -k 10 900 python -m pytest tests/test_halo_gpu.py -q -m gpu -k "processor_grid and 2]" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -4 | cut -c1-250 )
done
