"""In-kernel timeline of the sampler tail of the Depth step program's linears[k] phases (sampling mode; needs the -DCH_LOG build: tests/microbench/build_stamped_lib.sh,
then MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/nest_sampler_stamps.py). Workgroup 0, wave 0: 8 = rows published, 11 = own rows / noise
staged, 12 = all 2 048 logits gathered, 13 = soft-max statistics (two barriers), 14 = ranks of the own rows counted (one barrier), 15 = candidate published, then the next
phase's 10 (start) and 2 (token merged)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
cfg.temp, cfg.temp_text, cfg.top_k, cfg.top_k_text = 0.8, 0.7, 250, 25
m = hu.Model("hip", cfg, seed=0)
rng = np.random.default_rng(0)
for _ in range(8):
    m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
L.ggml_backend_synchronize(m.be)
buf = (C.c_ulonglong * (2 * 512 * 16))()
assert L.ggml.mi355x_chain_log_read(buf) == 0
rec = np.frombuffer(buf, np.uint64).reshape(2, 512, 16).astype(np.int64)
per = 25
heads = [p for p in range(per - 1, 200, per) if rec[0, p, 15] > 0]
print(f"{len(heads)} sampled linears[k] phases; launch span {(rec[0, heads[-1], 15] - rec[0, 0, 0]) / 100.0:.1f} us")
names = ["publish->staged", "gather 2048 logits", "soft-max stats", "ranks", "candidate", "next phase: token merged"]
rows = []
for p in heads:
    r = rec[0, p]
    nxt = rec[0, p + 1] if p + 1 < 512 and rec[0, p + 1, 2] > 0 else None
    rows.append([(r[11] - r[8]) / 100.0, (r[12] - r[11]) / 100.0, (r[13] - r[12]) / 100.0, (r[14] - r[13]) / 100.0, (r[15] - r[14]) / 100.0, ((nxt[2] - r[15]) / 100.0) if nxt is not None else float("nan")])
a = np.array(rows)
print(" | ".join(f"{n} {np.nanmean(a[:, i]):.2f}" for i, n in enumerate(names)), "us")
print("whole linears[k] phase (start -> candidate):", f"{np.mean([(rec[0, p, 15] - rec[0, p, 10]) / 100.0 for p in heads]):.2f} us")
