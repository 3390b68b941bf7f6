"""Per-step logit errors of the real-width Depth transformer test model: fused vs no-attention-prologue vs per-node."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
import test_hip_frame as t
cfg = hu.hot.tiny(hu.L, dep_q=4, n_q=8)
cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden = 1024, 16, 2, 2816
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
steps = 12
ref, _ = t.run_lm("oracle", cfg, steps)
for name, flags in [(os.environ.get("LABEL", "fused"), 0), ("per-node", 7)]:
    got, _ = t.run_lm("hip", cfg, steps, flags=flags)
    e_txt = [hu.rel_err(a[3], b[3]) for a, b in zip(ref, got)]
    e_dep = [hu.rel_err(a[4], b[4]) for a, b in zip(ref, got)]
    print(name, "text", " ".join(f"{e:.1e}" for e in e_txt))
    print(name, "dep ", " ".join(f"{e:.1e}" for e in e_dep), "tokens equal:", all(a[:3] == b[:3] for a, b in zip(ref, got)))
