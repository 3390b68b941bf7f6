"""LM steps only (no codec) at the moshika configuration: blocking steps vs run-ahead (moshi_hot_lm_step_run_ahead). The gap between the run-ahead rate here and
bench.py's default line is what the codec stream running beside the LM costs it."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import hot_util as hu
L = hu.L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for chain in (0, 2):
    cfg = hu.hot.moshika(L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.chain_depth = chain
    m = hu.Model("hip", cfg, seed=0)
    codes = (C.c_int32 * 8)(*range(8)); txt = C.c_int32(); aud = (C.c_int32 * 32)()
    step = (lambda: L.moshi_hot_lm_step_run_ahead(m.m, codes, C.byref(txt), aud)) if chain == 2 else (lambda: L.moshi_hot_lm_step(m.m, codes, C.byref(txt), aud))
    for _ in range(8):
        step()
    L.ggml_backend_synchronize(m.be); t0 = time.perf_counter()
    for _ in range(n):
        step()
    L.ggml_backend_synchronize(m.be); dt = time.perf_counter() - t0
    print(f"chain_depth {chain}: {n / dt:7.1f} LM steps/s ({1e6 * dt / n:.0f} us per step)", flush=True)
    if chain == 2:
        L.moshi_hot_lm_step_run_ahead(m.m, None, C.byref(txt), aud)
    m.free()
