export AMD_LOG_LEVEL=0
for rep in 1 2 3 4 5 6; do
rm -f /tmp/gstr.*
echo "== walled test rep $rep"
( SF_DEBUG_HALO_TRACE=/tmp/gstr SF_DEBUG_HALO=1 SF_TEST_TIMEOUT=12 timeout -k 10 200 python -m pytest tests/test_halo_gpu.py -q -m gpu -x -k "processor_grid and hertz-False-2" 2>&1 | grep "ran out\|passed\|failed" | cut -c1-300 )
for r in 0 1 2 3; do echo "-- rank $r"; tail -n 4 /tmp/gstr.$r; done
done
