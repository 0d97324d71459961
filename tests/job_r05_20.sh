export AMD_LOG_LEVEL=0
python -m pytest tests/test_reference_pins_gpu.py -m gpu -q -k "cloud" 2>&1 | grep -v "amdgpu.ids" | tail -25 | cut -c1-220
SF_TEST_TIMEOUT=30 FUZZ_DIRECT=2 timeout -k 10 1700 python tests/fuzz_bricks.py 7 40 2>&1 | grep "^ok\|^FAILED\|failed" | cut -c1-260
