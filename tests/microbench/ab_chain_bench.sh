# A/B of the persistent chain engine at bench level: headline (run-ahead loop), serial loop and the Depth phase, chains off / on at several grid sizes
for cfg in "off 0 256" "on 1 128" "on 1 256" "off 0 256" "on 1 256"; do set -- $cfg
  MI355X_CHAIN=$2 MI355X_CHAIN_GRID=$3 python bench.py --steps 60 --warmup 8 --no-extras --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('chain $1 grid $3: headline', d['value'], 'serial', d['serial_loop']['value'], 'phase_us', d.get('phase_us'), 'chained', d.get('chained_matvecs_in_last_plan'))"
done
