"""MI355X_DUMP_PLAN=1 python tests/microbench/plan_dump_tts.py : plan listing of the LM graphs at the tts-1.6b-shaped config (BASELINE.json configs[1])"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.tts_like(L); cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
cfg.num_layers = 2
if len(sys.argv) > 1: cfg.dep_q = cfg.n_q = int(sys.argv[1])
m = hu.Model("hip", cfg)
for _ in range(20):
    m.lm_step([])
print(m.stats().kernels_in_last_plan, "kernels in the last plan;", m.stats().chained_matvecs_in_last_plan, "chained")
m.free()
