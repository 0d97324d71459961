export AMD_LOG_LEVEL=0
for rep in 1 2 3 4 5 6; do
echo "== walled test rep $rep"
( SF_DEBUG_HALO=1 SF_TEST_TIMEOUT=15 timeout -k 10 200 python -m pytest tests/test_halo_gpu.py -q -m gpu -x -k "processor_grid and hertz-False-2" 2>&1 | grep "ran out\|passed\|failed" | cut -c1-400 )
done
