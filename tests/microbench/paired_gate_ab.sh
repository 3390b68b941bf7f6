for i in 1 2 3; do for P in 0 1; do
  if [ $P = 1 ]; then export MI355X_PAIRED_GATE=1; else unset MI355X_PAIRED_GATE; fi
  timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras | P=$P python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('paired_gate', os.environ['P'], d['value'])"
done; done
