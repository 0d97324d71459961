export AMD_LOG_LEVEL=0
for rep in 1 2; do
SF_DEBUG_XCD=1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-coupled --no-parity --no-fluidised --no-configs 2>gpurun_out/xcd_$rep.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('value %.4e kernel %.1f us frac %.4f' % (d['value'], r['mean_kernel_us'], r['frac']))"
grep "XCD times" gpurun_out/xcd_$rep.err | cut -c1-220
done
