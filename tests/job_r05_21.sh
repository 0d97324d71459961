export AMD_LOG_LEVEL=0
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do for d in 0 1; do
echo -n "SF_GHOST_DEFER=$d : "
SF_GHOST_DEFER=$d python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-coupled --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d['fluidised_bed']; c=d['configs']
print('headline %.4e rebuild_ms %.3f | fluidised %.4e (%.3f ms/step, rebuild %.3f ms) | C2 %.4e C3 %.4e (%.3f ms/step) C5 %.4e' % (d['value'], d['config']['neighbor_rebuild_ms'], f['value'], f['ms_per_step'], f['neighbor_rebuild_ms'], c['C2']['value'], c['C3']['value'], c['C3']['ms_per_step'], c['C5']['value']))"
done; done
