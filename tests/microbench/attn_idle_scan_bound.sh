for i in 1 2 3; do for A in 0 1; do
  MI355X_ATTN_ASSUME_SHORT=$A timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | A=$A python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('assume_short', os.environ['A'], d['value'], d['phase_us']['temporal'])"
done; done
