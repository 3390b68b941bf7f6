import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
from ggml_util import F32
cfg = hu.hot.tiny_tts(hu.L, linear_type=F32, layers=1)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
m = hu.Model("hip", cfg, seed=0)
hu.set_conditions(m, cfg)
hu.set_text_hook(m, lambda offset, sampled: 9)
m.lm_step_n([])
