"""Where does the device leave the oracle on the contractive full-width model? Prints rel err of the Temporal stack's checkpoints for a few frames
and the per-layer growth through the teacher-forced probe."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
import parity_probe as pp
L = hu.L
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0 / 16
cfg = hu.hot.moshika(L); cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0; cfg.update_scale = scale
rng = np.random.default_rng(21)
inputs = [rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist() for _ in range(4)]
ms = {k: hu.Model(k, cfg, seed=0) for k in ("oracle", "hip")}
for i, ia in enumerate(inputs):
    outs = {}
    for k, m in ms.items():
        r = m.lm_step(ia)
        outs[k] = (r, m.last_raw(), {w: m.read(w, n).copy() for w, n in (("transformer_in", cfg.dim), ("stack_out", cfg.dim), ("transformer_out", cfg.dim), ("text_logits", cfg.text_card))},
                   [m.read(f"dep_logits{q}", cfg.card).copy() for q in range(cfg.dep_q)])
    a, b = outs["oracle"], outs["hip"]
    print(f"frame {i}: tokens equal {a[1] == b[1]}", {w: f"{hu.rel_err(a[2][w], b[2][w]):.2e}" for w in a[2]}, "depth", [f"{hu.rel_err(x, y):.1e}" for x, y in zip(a[3], b[3])], flush=True)
    x = a[2]["stack_out"]; print("   stack_out rms", float(np.sqrt((x * x).mean())), "max", float(np.abs(x).max()), "logit max", float(np.abs(a[2]["text_logits"]).max()))
x = (np.random.default_rng(11).standard_normal(cfg.dim) * 4).astype(np.float32)
for layer in range(cfg.num_layers):
    a, ya = pp.probe(ms["oracle"], 0, layer, 0, x, 40)
    c, yc = pp.probe(ms["hip"], 0, layer, 0, x, 40)
    print(f"layer {layer}: out rel err {hu.rel_err(ya, yc):.2e}  |x| rms {float(np.sqrt((ya*ya).mean())):.3f}", flush=True)
    x = ya
