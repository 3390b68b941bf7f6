"""In-kernel timeline of the persistent stream engine (needs the -DST_LOG build: tests/microbench/build_stamped_lib.sh, then
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/stream_stamps.py). Wave 0 of the first and last workgroup stamp
s_memrealtime (10 ns ticks) at the stages of every phase of the LAST stream launch of a frame (the log is overwritten by every launch)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
cfg.num_layers = int(os.environ.get("LAYERS", "3"))   # the log shows the last launch: with 3 layers that is layer 1 -> 2 (four phases) unless LAST=1
m = hu.Model("hip", cfg, seed=0)
rng = np.random.default_rng(0)
for _ in range(6):
    m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
L.ggml_backend_synchronize(m.be)
lib = L.ggml
buf = (C.c_ulonglong * (2 * 64 * 8))()
lib.mi355x_stream_log_read(buf)
rec = np.frombuffer(buf, np.uint64).reshape(2, 64, 8).astype(np.int64)
for wgi, label in ((0, "workgroup 0"), (1, "last workgroup")):
    r = rec[wgi]
    n = int((r[:, 0] > 0).sum())
    t0 = r[0, 0]
    print(f"{label}: {n} phases (times in us from the first streamer stamp)")
    print("phase | streamer: start  blocks-ready  dots-done  published | gatherer: start  own-wg-pub  gathered  blocks-written")
    for p in range(n):
        t = [(r[p, i] - t0) / 100.0 for i in range(8)]
        print(f"{p:5d} | {t[0]:15.2f} {t[1]:13.2f} {t[2]:10.2f} {t[3]:10.2f} | {t[4]:15.2f} {t[5]:10.2f} {t[6]:9.2f} {t[7]:15.2f}")
