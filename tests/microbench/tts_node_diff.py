import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
from ggml_util import F32
L = hu.L
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = hu.hot.tiny_tts(L, linear_type=int(sys.argv[2]) if len(sys.argv) > 2 else 0, layers=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
ms = {}
for kind in ("oracle", "hip"):
    m = hu.Model(kind, cfg, seed=0, flags=flags if kind == "hip" else 0)
    hu.set_conditions(m, cfg)
    n = cfg.text_card + 1
    hu.set_text_hook(m, lambda offset, sampled: ((offset * 7) % 5) * n + (offset * 13) % cfg.text_card)
    m.lm_step_n([])
    m.lm_step_n([])
    ms[kind] = m
ga, gb = L.moshi_hot_graph(ms["oracle"].m, 0), L.moshi_hot_graph(ms["hip"].m, 0)
for i in range(L.ggml_graph_n_nodes(ga)):
    ta, tb = L.ggml_graph_node(ga, i), L.ggml_graph_node(gb, i)
    t = ta.contents
    if t.type != 0 or not L.ggml_is_contiguous(ta):
        continue
    nb = L.ggml_nbytes(ta)
    a = np.zeros(nb // 4, np.float32); b = np.zeros(nb // 4, np.float32)
    L.ggml_backend_tensor_get(ta, a.ctypes.data, 0, nb); L.ggml_backend_tensor_get(tb, b.ctypes.data, 0, nb)
    if not np.isfinite(a).all():
        continue
    e = hu.rel_err(a, b)
    if e > 1e-5:
        print(f"node {i:4d} {L.ggml_op_name(t.op).decode():14s} [{t.ne[0]} {t.ne[1]} {t.ne[2]}] rel err {e:.2e}")
