"""Per layer, teacher-forced (tests/parity_probe.py): which rounding sites flip between the oracle and the device, and how far the layer output moves.
argv: update_scale (default 1), layers to show (default 8)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
import parity_probe as pp
L = hu.L
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
nl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = hu.hot.moshika(L); cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0; cfg.update_scale = scale; cfg.num_layers = nl
ref, dev = hu.Model("oracle", cfg, seed=0), hu.Model("hip", cfg, seed=0)
x = (np.random.default_rng(11).standard_normal(cfg.dim) * 4).astype(np.float32)
for layer in range(nl):
    a, ya = pp.probe(ref, 0, layer, 0, x, 0)
    L.ggml_backend_mi355x_set_flags(dev.be, 7)
    b, yb = pp.probe(dev, 0, layer, 0, x, 0)
    L.ggml_backend_mi355x_set_flags(dev.be, 0)
    c, yc = pp.probe(dev, 0, layer, 0, x, 0)
    st = pp.compare_layer(a, b, f"layer {layer}", taint_tol=1.0, max_flip_frac=1.0)
    errs = [(n.idx, n.op, f"{hu.rel_err(n.values, m.values):.1e}") for n, m in zip(a, b) if n.values is not None and m.values is not None and not n.view_src and np.isfinite(n.values).all() and hu.rel_err(n.values, m.values) > 2e-6]
    print(f"layer {layer}: y err per-node {hu.rel_err(ya, yb):.2e} fused {hu.rel_err(ya, yc):.2e} | clean-site flips {[s for s in st['site_flips'] if s[3]]} | nodes beyond 2e-6: {errs[:14]}", flush=True)
    x = ya
