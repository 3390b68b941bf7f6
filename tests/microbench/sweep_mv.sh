for tpw in 2 4 8; do for gm in 128 256 512 1024; do
  echo "TPW=$tpw GRID_MIN=$gm: $(MI355X_MV_TPW=$tpw MI355X_MV_GRID_MIN=$gm timeout 120 python bench.py --steps 30 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phase_us'], d['roofline']['achieved'])")"
done; done
