"""Node-by-node comparison of the Temporal graph between the oracle and the HIP backend at moshika's widths (1 layer):
prints the first nodes whose values differ by more than a threshold. flags=1|2|4 runs one kernel per node on the device."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
import ggml_util as gu
L = hu.L
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 7
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = hu.hot.moshika(L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
cfg.num_layers = layers
rng = np.random.default_rng(5)
ia = rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist()
models = {}
for kind in ("oracle", "hip"):
    m = hu.Model(kind, cfg, seed=0, flags=flags if kind == "hip" else 0)
    m.lm_step(ia)
    models[kind] = m
ga = L.moshi_hot_graph(models["oracle"].m, 0)
gb = L.moshi_hot_graph(models["hip"].m, 0)
n = L.ggml_graph_n_nodes(ga)
assert n == L.ggml_graph_n_nodes(gb)
shown = 0
for i in range(n):
    ta, tb = L.ggml_graph_node(ga, i), L.ggml_graph_node(gb, i)
    t = ta.contents
    if t.type not in (0, 1) or not L.ggml_is_contiguous(ta):   # F32 / F16 contiguous only
        continue
    nb = L.ggml_nbytes(ta)
    dt = np.float32 if t.type == 0 else np.float16
    a = np.zeros(nb // dt().itemsize, dt); b = np.zeros(nb // dt().itemsize, dt)
    L.ggml_backend_tensor_get(ta, a.ctypes.data, 0, nb); L.ggml_backend_tensor_get(tb, b.ctypes.data, 0, nb)
    a = a.astype(np.float32); b = b.astype(np.float32)
    if i in (103, 105) or t.type == 1:
        d = np.abs(a - b); j = int(d.argmax())
        print(f"   node {i} type {t.type} op {L.ggml_op_name(t.op).decode()} n={a.size} nonzero a={np.count_nonzero(a)} b={np.count_nonzero(b)} ndiff={np.count_nonzero(d)} maxdiff at {j}: {a[j]} vs {b[j]}")
    if not np.isfinite(a).all():
        continue
    e = hu.rel_err(a, b)
    if e > 1e-5:
        print(f"node {i:4d} {L.ggml_op_name(t.op).decode():14s} [{t.ne[0]} {t.ne[1]} {t.ne[2]}] name={t.name.decode()[:40]:40s} rel err {e:.2e}")
        shown += 1
        if shown >= 12:
            break
print("done", n, "nodes")
# are the BF16 cache flips near-ties? the V projection (last third of the in_proj output) against what attention returned at step 0 (p == 1)
def fetch(g, i, dt=np.float32):
    t = L.ggml_graph_node(g, i); nb = L.ggml_nbytes(t); a = np.zeros(nb // 4, dt); L.ggml_backend_tensor_get(t, a.ctypes.data, 0, nb); return a
for i in range(n):
    t = L.ggml_graph_node(ga, i).contents
    if t.op == L.ggml_graph_node(ga, 103).contents.op and t.ne[0] == 3 * cfg.dim and t.ne[1] == 1:
        va, vb = fetch(ga, i)[2 * cfg.dim:], fetch(gb, i)[2 * cfg.dim:]
        oa, ob = fetch(ga, 105), fetch(gb, 105)
        print(f"in_proj node {i}: V f32 oracle vs hip rel err {hu.rel_err(va, vb):.2e}")
        for j in np.nonzero(oa != ob)[0]:
            ulp = 2.0 ** (np.floor(np.log2(abs(va[j]))) - 7)
            frac = (abs(va[j]) / ulp) % 1.0
            print(f"   elem {j}: V oracle {va[j]:.9g} hip {vb[j]:.9g} (diff {abs(va[j]-vb[j])/abs(va[j]):.1e} rel) -> cache {oa[j]} vs {ob[j]}; position between bf16 neighbours {frac:.6f}")
        break
