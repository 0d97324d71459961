#!/bin/bash
# GPU box: the LDS-DMA variants (tests/build_variant.sh dma -DSF_EXP_DMA=1 / dma2 -DSF_EXP_DMA=2): parity against the
# oracle on the headline bed, then kernel time against the shipped library, interleaved
for v in "$@"; do
SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so timeout 300 python bench.py --steps 2 --warmup 1 --no-coupled --no-fluidised 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d=json.loads(l); print('$v', {k:d['parity'][k] for k in ('ok','max_abs_dx_over_d','max_rel_v','max_rel_f','gpu_rebuilds')})
"
done
bash tests/ab_lib_env.sh default "$@" 2>&1 | tail -8
