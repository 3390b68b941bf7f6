import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
cfg = hu.hot.tiny_personaplex(hu.L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
rec = {}
for kind, flags in (("oracle", 0), ("hip", 0), ("hip", 7)):
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    r = []
    for i in range(2):
        m.lm_step(list(range(8)))
        r.append((m.last_raw(), [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)]))
    rec[(kind, flags)] = r
    m.free()
for key in (("hip", 0), ("hip", 7)):
    for i in range(2):
        a, b = rec[("oracle", 0)][i], rec[key][i]
        print(key, "step", i, "tokens equal", a[0] == b[0], " ".join(f"{hu.rel_err(x, y):.0e}" for x, y in zip(a[1], b[1])))
