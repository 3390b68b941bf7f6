"""Voice-prompt embedding frames (moshi_hot_lm_step_embedding: Temporal stack built on the scratch context every frame, lm.h:694-709, 1004-1037)
at PersonaPlex widths: frames/s."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
cfg = hu.hot.personaplex(hu.L)
cfg.context = 2000
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
m = hu.Model("hip", cfg, seed=0)
rng = np.random.default_rng(0)
embs = [rng.standard_normal(cfg.dim).astype(np.float32) for _ in range(40)]
for e in embs[:4]:
    m.lm_step_embedding(e)
hu.L.ggml_backend_synchronize(m.be)
t0 = time.perf_counter()
for e in embs:
    m.lm_step_embedding(e)
hu.L.ggml_backend_synchronize(m.be)
dt = time.perf_counter() - t0
print(f"embedding frames: {len(embs) / dt:.1f} frames/s ({1e3 * dt / len(embs):.2f} ms per frame)")
