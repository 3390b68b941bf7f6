import os, sys
sys.path.insert(0, "tests")
import numpy as np
import hot_util as hu
cfg = hu.hot.moshika(hu.L)
cfg.num_layers, cfg.context = 2, 64
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
m = hu.Model("hip", cfg, seed=0, flags=32)
rng = np.random.default_rng(3)
for _ in range(3):
    print(m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist()))
st = m.stats()
print("chained", st.chained_matvecs_in_last_plan, "programs", st.chain_step_programs_in_last_plan)
