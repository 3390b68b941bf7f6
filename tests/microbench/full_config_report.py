"""Full moshika q4_k config, HIP vs oracle: per-frame, per-Depth-step logit errors and top-2 margins (is a token mismatch a near-tie?)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
cfg = hu.hot.moshika(hu.L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(5)
inputs = [rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist() for _ in range(steps)]
rec = {}
for kind in ("oracle", "hip"):
    m = hu.Model(kind, cfg, seed=0)
    r = []
    for ia in inputs:
        m.lm_step(ia)
        raw = m.last_raw()
        r.append((raw, m.read("text_logits", cfg.text_card).copy(), [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)]))
        if kind == "hip":
            pass
    rec[kind] = r
    m.free()
for i in range(steps):
    a, b = rec["oracle"][i], rec["hip"][i]
    print(f"step {i}: text tok {a[0][0]} / {b[0][0]} err {hu.rel_err(a[1], b[1]):.1e}")
    for k in range(cfg.dep_q):
        la, lb = a[2][k], b[2][k]
        top = np.sort(la)[-2:]
        print(f"   dep {k}: tok {a[0][1][k]} / {b[0][1][k]}  logit err {hu.rel_err(la, lb):.1e}  oracle top-2 margin {(top[1]-top[0])/np.abs(la).max():.1e}")
        if a[0][1][k] != b[0][1][k]:
            break
