import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
from ggml_util import F32
cfg = hu.hot.tiny_personaplex(hu.L, linear_type=F32)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
rng = np.random.default_rng(9)
embs = [rng.standard_normal(cfg.dim).astype(np.float32) for _ in range(3)]
for kind, flags in (("oracle", 0), ("hip", 0), ("hip", 7)):
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    for e in embs:
        m.lm_step_embedding(e)
    m.system_prompts([5, 6, 7, 8])
    for i in range(3):
        m.lm_step(list(range(i, i + 8)))
        raw = m.last_raw()
        lg = [m.read(f"dep_logits{k}", cfg.card) for k in range(cfg.dep_q)]
        am = [int(np.argmax(l)) for l in lg]
        top = [np.sort(l)[-2:] for l in lg]
        print(kind, flags, i, "text", raw[0], "raw", raw[1], flush=True)
        print("      argmax(host)", am, "margins", " ".join(f"{(t[1]-t[0])/np.abs(l).max():.0e}" for t, l in zip(top, lg)))
    m.free()
