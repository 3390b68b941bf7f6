"""Node-by-node oracle vs device (one kernel per node) comparison of one cached graph of a tiny PersonaPlex-shaped model."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
which = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = hu.hot.tiny_personaplex(L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
models = {}
for kind in ("oracle", "hip"):
    m = hu.Model(kind, cfg, seed=0, flags=7 if kind == "hip" else 0)
    m.lm_step(list(range(8)))
    models[kind] = m
ga, gb = L.moshi_hot_graph(models["oracle"].m, which), L.moshi_hot_graph(models["hip"].m, which)
n = L.ggml_graph_n_nodes(ga)
shown = 0
for i in range(n):
    ta, tb = L.ggml_graph_node(ga, i), L.ggml_graph_node(gb, i)
    t = ta.contents
    if t.type not in (0, 26) or not L.ggml_is_contiguous(ta):
        continue
    nb = L.ggml_nbytes(ta)
    dt = np.float32 if t.type == 0 else np.int32
    a = np.zeros(nb // 4, dt); b = np.zeros(nb // 4, dt)
    L.ggml_backend_tensor_get(ta, a.ctypes.data, 0, nb); L.ggml_backend_tensor_get(tb, b.ctypes.data, 0, nb)
    if dt == np.int32:
        if not np.array_equal(a, b):
            print(f"node {i:4d} {L.ggml_op_name(t.op).decode():14s} I32 {a[:8]} vs {b[:8]}"); shown += 1
        continue
    if not np.isfinite(a).all():
        continue
    e = hu.rel_err(a, b)
    if e > 1e-5:
        srcs = [(L.ggml_op_name(t.src[k].contents.op).decode(), t.src[k].contents.type, list(t.src[k].contents.ne)) for k in range(3) if t.src[k]]
        print(f"node {i:4d} {L.ggml_op_name(t.op).decode():14s} [{t.ne[0]} {t.ne[1]} {t.ne[2]}] rel err {e:.2e} srcs {srcs}")
        shown += 1
    if shown >= 8:
        break
print("done", n)
