"""Prompt prefill at the full PersonaPlex / moshika widths: n provided frames stepped one by one (the reference's way) vs moshi_hot_prefill
in chunks of T. Prints frames/s of each."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chunks = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 16, 32]
cfg = hu.hot.personaplex(L)
cfg.context = 2000
if os.environ.get("PREFILL_QUANT"):          # q8_0 / q4_0 linears at the same widths (the int8-MFMA mat-mul of those weight types)
    cfg.linear_type = {"q8_0": 8, "q4_0": 2, "q4_k": 12}[os.environ["PREFILL_QUANT"]]
if os.environ.get("PREFILL_DIM"):            # e.g. 2048: the tts / stt width (16 layers, feed-forward 5632)
    cfg.dim, cfg.num_heads, cfg.num_layers, cfg.ffn_hidden = int(os.environ["PREFILL_DIM"]), 16, 16, 5632
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
rng = np.random.default_rng(1)
frames = [[int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist() for _ in range(n)]
m = hu.Model("hip", cfg, seed=0, flags=int(os.environ.get("PREFILL_FLAGS", "0")))   # 2 = no hipGraph capture (needed under rocprofv3)
def sync():
    L.ggml_backend_synchronize(m.be)
for f in frames[:4]:
    m.lm_step_n(f)
sync(); t0 = time.perf_counter()
for f in frames:
    m.lm_step_n(f)
sync(); dt = time.perf_counter() - t0
print(f"frame by frame: {n / dt:8.1f} frames/s ({1e3 * dt / n:.2f} ms per frame)")
for T in chunks:
    m.prefill(frames[:T], T)          # plan + table warm-up
    sync(); t0 = time.perf_counter()
    m.prefill(frames, T)
    sync(); dt = time.perf_counter() - t0
    print(f"prefill T={T:3d}: {n / dt:8.1f} frames/s ({1e3 * dt / n:.2f} ms per frame)", flush=True)
