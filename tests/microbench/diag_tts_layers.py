"""Node-by-node comparison of the first Temporal layers of the tts-shaped model (cross-attention over the cached condition) at full width: oracle vs device."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu, parity_probe as pp
L = hu.L
cfg = hu.hot.tts_like(L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
cfg.num_layers = 3
ref, dev = hu.Model("oracle", cfg, seed=0), hu.Model("hip", cfg, seed=0)
hu.set_conditions(ref, cfg); hu.set_conditions(dev, cfg)
rng = np.random.default_rng(3)
x = (rng.standard_normal(cfg.dim) * 4).astype(np.float32)
for layer in range(cfg.num_layers):
    a, ya = pp.probe(ref, 0, layer, 0, x, 0)
    L.ggml_backend_mi355x_set_flags(dev.be, 1 | 2 | 4)
    b, yb = pp.probe(dev, 0, layer, 0, x, 0)
    worst = []
    for na, nb in zip(a, b):
        if na.values is None or nb.values is None or na.view_src:
            continue
        if np.isnan(nb.values).all():
            continue
        worst.append((hu.rel_err(na.values, nb.values), na.idx, na.op, na.ne))
    worst.sort(reverse=True)
    print("layer", layer, "output rel err", hu.rel_err(ya, yb), "nodes", len(a), "worst:", [(f"{e:.1e}", i, o, n) for e, i, o, n in worst[:6]])
    first = next(((e, i, o, n) for e, i, o, n in sorted(worst, key=lambda t: t[1]) if e > 1e-5), None)
    print("   first node beyond 1e-5:", first)
    x = ya
