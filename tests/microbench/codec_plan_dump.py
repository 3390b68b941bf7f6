"""MI355X_DUMP_PLAN=1 python tests/microbench/codec_plan_dump.py [flags] : plan listing of the codec's encode / decode graphs at moshika's Mimi (one frame each)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L)
cfg.enable_lm = 0
m = hu.Model("hip", cfg, seed=0, flags=int(sys.argv[1]) if len(sys.argv) > 1 else 32)
rng = np.random.default_rng(2)
for _ in range(2):
    sys.stderr.write("==== encode\n"); sys.stderr.flush()
    codes = m.mimi_encode((rng.standard_normal(1920) * 0.1).astype(np.float32))
    sys.stderr.write("==== decode\n"); sys.stderr.flush()
    m.mimi_decode(codes)
m.free()
