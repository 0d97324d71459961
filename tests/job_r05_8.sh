export AMD_LOG_LEVEL=0
bash tests/ab_gs_variants.sh 126000
for v in "fine system" "coarse plain"; do set -- $v
  echo "== tests: area $1 loads $2"
  ( SF_GS_AREA=$1 SF_GS_LOADS=$2 SF_HALO_DIRECT_TIMEOUT=20 timeout -k 10 900 python -m pytest tests/test_halo_gpu.py -q -m gpu -k "processor_grid and 2]" 2>&1 | grep -v "Gloo\|amdgpu.ids\|socket.cpp" | tail -12 | cut -c1-250 )
done
