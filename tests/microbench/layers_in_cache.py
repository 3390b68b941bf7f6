"""What would the Temporal layer cost if its weights did not come from HBM? moshika's widths without a Depth transformer (dep_q = 0) and without the
codec, L layers: at L = 1, 2 the whole weight set of a step (116 MB per layer + 74 MB of text linear) stays in the 256 MB Infinity Cache from one step to
the next, at L = 8, 32 every byte crosses HBM. us per step / per added layer."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hot_util as hu
L = hu.L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
res = {}
for layers in (1, 2, 4, 8, 32):
    cfg = hu.hot.moshika(L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.num_layers = layers
    cfg.n_q, cfg.dep_q = 8, 0
    for i in range(hu.hot.MAX_CB):
        cfg.delays[i] = 0
    cfg.mimi_n_q = 8
    m = hu.Model("hip", cfg, seed=0)
    codes = (C.c_int32 * 8)(*range(8)); txt = C.c_int32(); aud = (C.c_int32 * 32)()
    for _ in range(8):
        L.moshi_hot_lm_step(m.m, codes, C.byref(txt), aud)
    L.ggml_backend_synchronize(m.be); t0 = time.perf_counter()
    for _ in range(n):
        L.moshi_hot_lm_step(m.m, codes, C.byref(txt), aud)
    L.ggml_backend_synchronize(m.be); dt = time.perf_counter() - t0
    res[layers] = 1e6 * dt / n
    print(f"{layers:2d} layers: {res[layers]:8.1f} us per step", flush=True)
    m.free()
ks = sorted(res)
for a, b in zip(ks, ks[1:]):
    print(f"  {a:2d} -> {b:2d} layers: {(res[b] - res[a]) / (b - a):6.2f} us per added layer")
