cd /tmp && export TMPDIR=/tmp
for v in firstonly default; do
  p=""; [ "$v" != "default" ] && p=$GRAFT_REPO_ROOT/sedifoam_amd/libsedifoam_amd_$v.so
  SF_LIB_PATH=$p timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_build_$v -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fluidised --no-coupled --no-configs --no-parity --no-kernel-profile > $GRAFT_REPO_ROOT/gpurun_out/kt_build_$v.log 2>&1
  echo "== $v"; grep "k_build_neigh\|k_back_slots\|k_partner_tags" $GRAFT_REPO_ROOT/gpurun_out/kt_build_$v/p_kernel_stats.csv | cut -c1-60,200-400 | head -4
  grep "k_build_neigh" $GRAFT_REPO_ROOT/gpurun_out/kt_build_$v/p_kernel_stats.csv | awk -F'","|",|,' '{print "calls",$(NF-6),"avg ns",$(NF-4)}'
done
