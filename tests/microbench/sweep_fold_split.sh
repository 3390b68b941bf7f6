for s in 128 192 256 320; do
  for b in 1024; do
    MI355X_FOLD_SLOTS=$s MI355X_ATTN_BIG_MIN=$b python bench.py --context-fill 2800 --steps 60 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fill2800 slots $s big_min $b', d['value'], d['value_serial'], d['phase_us']['temporal'])"
    MI355X_FOLD_SLOTS=$s MI355X_ATTN_BIG_MIN=$b python bench.py --model personaplex --context 2000 --context-fill 1900 --steps 60 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pp1900   slots $s big_min $b', d['value'], d['value_serial'], d['phase_us']['temporal'], d['phase_us']['depth'])"
  done
done
for b in 512 2048; do
    MI355X_FOLD_SLOTS=192 MI355X_ATTN_BIG_MIN=$b python bench.py --context-fill 2800 --steps 60 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fill2800 slots 192 big_min $b', d['value'], d['value_serial'], d['phase_us']['temporal'])"
    MI355X_FOLD_SLOTS=192 MI355X_ATTN_BIG_MIN=$b python bench.py --model personaplex --context 2000 --context-fill 1900 --steps 60 --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pp1900   slots 192 big_min $b', d['value'], d['value_serial'], d['phase_us']['temporal'], d['phase_us']['depth'])"
done
