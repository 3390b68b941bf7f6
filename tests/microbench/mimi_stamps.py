"""In-kernel timeline of the Mimi transformer program (needs the -DCH_LOG build: tests/microbench/build_stamped_lib.sh, then
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/mimi_stamps.py). Workgroup 0 (owner of attention part 0) and workgroup 255 (no part)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L)
cfg.enable_lm = 0
cfg.enable_mimi_encoder = 0
m = hu.Model("hip", cfg, seed=0, flags=32)
rng = np.random.default_rng(2)
for i in range(12):
    m.mimi_decode(rng.integers(0, cfg.mimi_codebook_size, cfg.mimi_n_q).tolist())
L.ggml_backend_synchronize(m.be)
buf = (C.c_ulonglong * (2 * 512 * 16))()
assert L.ggml.mi355x_chain_log_read(buf) == 0
rec = np.frombuffer(buf, np.uint64).reshape(2, 512, 16).astype(np.int64)
kinds = ["in_proj", "attention", "out_proj", "linear1", "linear2"]
n = int((rec[0, :, 10] > 0).sum())
first = int(np.argmax(rec[0, :, 10] > 0))
print(f"{n} phases logged from phase {first}; {(rec[0, first + n - 1, 8] - rec[0, first, 10]) / 100.0:.1f} us")
for who, name in ((0, "workgroup 0 (owner)"), (1, "workgroup 255")):
    print(name)
    r = rec[who]
    for ki, kind in enumerate(kinds):
        sel = [p for p in range(5, 40) if p % 5 == ki and r[p, 10] and r[p + 1 if p + 1 < 40 else p, 10]]
        def seg(a, b):
            v = [(r[p, b] - r[p, a]) for p in sel if r[p, a] and r[p, b]]
            return np.mean(v) / 100.0 if v else 0.0
        dur = np.mean([(r[p + 1, 10] - r[p, 10]) for p in sel if p + 1 < 40 and r[p + 1, 10]]) / 100.0
        print(f"  {kind:10s} {dur:6.2f} us | start -> input {seg(10, 2):5.2f} | -> xs ready {seg(2, 5):5.2f} | -> end {seg(5, 8):5.2f} | whole {seg(10, 8):5.2f}")
# the attention phase on its owner (attn_ring256_body's stage stamps 11 .. 15 between the phase's start 10 and end 8)
r = rec[0]
sel = [p for p in range(5, 40) if p % 5 == 1 and r[p, 10] and r[p, 11]]
names = [(10, 11, "q / k / v granules polled, rotated, ring rows stored"), (11, 12, "barrier"), (12, 0, "scores of the four passes"), (0, 13, "wave maxima, barrier"),
         (13, 3, "exp, wave sums"), (3, 4, "barrier"), (4, 6, "total, P x V partials"), (6, 7, "partials to LDS"), (7, 14, "barrier"),
         (14, 15, "64-group reduction by one wave, output granules"), (15, 8, "exit barriers")]
for a, b, what in names:
    v = [(r[p, b] - r[p, a]) / 100.0 for p in sel if r[p, a] and r[p, b]]
    print(f"  attention {what:60s} {np.mean(v) if v else 0.0:5.2f} us")
