"""MI355X_DUMP_PLAN=1 python tests/microbench/plan_dump.py [sampled] : plan listing (GENERIC = one kernel per node) of the LM graphs at moshika widths"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L); cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
cfg.num_layers = 2
if len(sys.argv) > 1 and sys.argv[1] == "sampled":
    cfg.temp, cfg.temp_text = 0.8, 0.7
m = hu.Model("hip", cfg)
m.lm_step([0] * 8)
m.lm_step([0] * 8)
m.free()
