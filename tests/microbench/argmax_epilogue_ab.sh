for i in 1 2; do for A in 0 1; do
  if [ $A = 1 ]; then export MI355X_NO_ARGMAX_EPILOGUE=1; else unset MI355X_NO_ARGMAX_EPILOGUE; fi
  timeout 300 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-roofline --no-extras 2>/dev/null | A=$A python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('no_argmax_epilogue', os.environ['A'], d['value'])"
done; done
