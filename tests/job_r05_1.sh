tests/ab_tail.sh tail > gpurun_out/r05_ab_tail.txt 2>&1
tests/ab_lib_env.sh default tail:SF_TAIL_FRAC=0 >> gpurun_out/r05_ab_tail.txt 2>&1
tests/ab_poly.sh default tail c5w3u c5w3 default tail c5w3u c5w3 > gpurun_out/r05_ab_poly.txt 2>&1
for m in cohesive lub; do for v in tail c5w3u c5w3; do echo -n "$v $m: "; SF_LIB_PATH=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so python tests/micro/poly_bench.py 500000 $m 2>/dev/null | tail -1; done; done >> gpurun_out/r05_ab_poly.txt 2>&1
tests/exp_stamp.sh > gpurun_out/r05_stamp_plain.txt 2>&1
SF_TAIL_FRAC=0.08 tests/exp_stamp.sh > gpurun_out/r05_stamp_tail08.txt 2>&1
cat gpurun_out/r05_ab_tail.txt gpurun_out/r05_ab_poly.txt; tail -30 gpurun_out/r05_stamp_tail08.txt
