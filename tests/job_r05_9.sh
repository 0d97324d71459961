export AMD_LOG_LEVEL=0
bash tests/ab_gs_arms.sh 126000
echo "== walled, ghost slots"; SF_HALO_DIRECT=2 timeout -k 10 600 python tests/micro/debug_gs_walled.py 2>&1 | grep "^steps"
echo "== periodic, ghost slots"; SF_HALO_DIRECT=2 timeout -k 10 400 python tests/micro/debug_gs_walled.py periodic 2>&1 | grep "^steps" | tail -3
