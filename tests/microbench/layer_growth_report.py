"""How the HIP-vs-oracle logit difference grows with depth at moshika's real widths (is it one wrong kernel or rounding-flip chaos?)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
for layers in (1, 2, 4, 8, 16, 32):
    cfg = hu.hot.moshika(hu.L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.num_layers = layers
    rng = np.random.default_rng(5)
    inputs = [rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist() for _ in range(2)]
    rec = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        r = []
        for ia in inputs:
            m.lm_step(ia)
            r.append((m.read("stack_out", cfg.dim).copy(), m.read("text_logits", cfg.text_card).copy()))
        rec[kind] = r
        m.free()
    e = [(hu.rel_err(a[0], b[0]), hu.rel_err(a[1], b[1])) for a, b in zip(rec["oracle"], rec["hip"])]
    print(f"layers {layers:2d}: stack_out err {e[0][0]:.1e} {e[1][0]:.1e}   text logits err {e[0][1]:.1e} {e[1][1]:.1e}", flush=True)
