#!/bin/bash
# sweep_quant_env.sh <quant> VAR "values" [rounds]
Q=$1; V=$2; L=$3; N=${4:-2}
for i in $(seq $N); do for a in $L; do
  export $V=$a
  timeout 300 python bench.py --quant $Q --steps 60 --warmup 8 --no-cpu-baseline --no-roofline | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('$Q $V', os.environ['$V'], d['value'], d['phase_us'])"
done; done
