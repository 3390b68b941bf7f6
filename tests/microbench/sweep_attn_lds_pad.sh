for r in 1 2; do for P in 0 8192 16384 32768 65536; do
  MI355X_ATTN_LDS_PAD=$P timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | P=$P python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('pad', os.environ['P'], d['value'], d['phase_us'])"
done; done
