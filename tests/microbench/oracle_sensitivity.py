"""How sensitive is the reference arithmetic itself? Oracle vs oracle with ONE layer-0 norm scale vector moved by one float ulp, at moshika's
widths. The quantisers on the path (Q8_K activations, BF16 cache rows) turn 1e-7 perturbations into rare full-step flips; this measures what
they do to the logits, i.e. the agreement any two correct implementations with different float summation orders can expect. CPU only."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
layers_list = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 4, 32]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for layers in layers_list:
    cfg = hu.hot.moshika(L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.num_layers = layers
    rng = np.random.default_rng(5)
    inputs = [rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist() for _ in range(steps)]
    rec = {}
    for variant in ("base", "ulp"):
        m = hu.Model("oracle", cfg, seed=0)
        if variant == "ulp":
            t = C.cast(L.moshi_hot_weight(m.m, b"lm.transformer.layers.0.norm1.alpha"), hu.pkg.TP)
            assert t
            w = np.zeros(cfg.dim, np.float32)
            L.ggml_backend_tensor_get(t, w.ctypes.data, 0, w.nbytes)
            w = np.nextafter(w, np.float32(np.inf), dtype=np.float32)
            L.ggml_backend_tensor_set(t, w.ctypes.data, 0, w.nbytes)
        r = []
        for ia in inputs:
            m.lm_step(ia)
            raw = m.last_raw()
            r.append((raw, m.read("stack_out", cfg.dim).copy(), m.read("text_logits", cfg.text_card).copy(),
                      [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)]))
        rec[variant] = r
        m.free()
    for i in range(steps):
        a, b = rec["base"][i], rec["ulp"][i]
        deps = [hu.rel_err(a[3][k], b[3][k]) for k in range(cfg.dep_q)]
        same = a[0] == b[0]
        print(f"layers {layers:2d} step {i}: stack_out {hu.rel_err(a[1], b[1]):.1e}  text logits {hu.rel_err(a[2], b[2]):.1e}  dep logits max {max(deps):.1e}  tokens equal {same}", flush=True)
