for bed in fluidised packed; do for sub in 2 1 3 2 1; do
  echo -n "$bed SF_SUB=$sub : "
  SF_SUB=$sub python bench.py --bed $bed --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-configs --no-fluidised --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); c=d['config']; print('value %.3e  kernel_us %.1f  rebuild_ms %.3f  rebuilds %d'%(d['value'],d['roofline']['mean_kernel_us'],c['neighbor_rebuild_ms'],c['neighbor_rebuilds_in_run']))"
done; done
