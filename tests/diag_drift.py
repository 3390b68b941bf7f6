"""Diagnostic (not a test): per-step logit error of the HIP backend vs the oracle, to separate rounding-flip
noise (random steps, ~1e-3) from a systematic bug (persistent / growing)."""
import sys
import numpy as np
import hot_util as hu

def run(kind, cfg, steps):
    m = hu.Model(kind, cfg, seed=0)
    rng = np.random.default_rng(3)
    rec = []
    for _ in range(steps):
        ia = rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist()
        r, txt, aud = m.lm_step(ia)
        rec.append((txt, aud, m.read("text_logits", cfg.text_card), m.read("transformer_out", cfg.dim)))
    m.free()
    return rec

for ctx in (12, 64):
    cfg = hu.hot.tiny(hu.L, context=ctx)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    a, b = run("oracle", cfg, 40), run("hip", cfg, 40)
    print("context", ctx)
    for i, (x, y) in enumerate(zip(a, b)):
        print(i, "tok_eq", x[0] == y[0] and x[1] == y[1], "logits %.2e" % hu.rel_err(x[2], y[2]), "tout %.2e" % hu.rel_err(x[3], y[3]))
