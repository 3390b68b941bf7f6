"""In-kernel timeline of the Q8_0 step program (tts-shaped Depth transformer; needs the -DCH_LOG build: tests/microbench/build_stamped_lib.sh, then
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/nest80_stamps.py). Wave 0 of workgroup 0 (owner of head 0) and of workgroup 255
(not an owner) stamp s_memrealtime at: phase start (10), hand-off taken (2), blocks ready (5), dots done (6), published (8); the first 512 phases are kept."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.tts_like(L)
cfg.num_layers = 2
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
m = hu.Model("hip", cfg, seed=0)
hu.set_conditions(m, cfg)
hu.set_text_hook(m, lambda offset, sampled: int((offset * 13) % cfg.text_card))
for _ in range(26):
    m.lm_step([])
L.ggml_backend_synchronize(m.be)
buf = (C.c_ulonglong * (2 * 512 * 16))()
assert L.ggml.mi355x_chain_log_read(buf) == 0
rec = np.frombuffer(buf, np.uint64).reshape(2, 512, 16).astype(np.int64)
per, Lc = 5 * cfg.dep_layers + 1, cfg.dep_layers
kinds = []
for l in range(Lc):
    kinds += ["in_proj0" if l == 0 else "in_proj", "attention", "out_proj", "linear_in", "linear_out"]
kinds.append("head")
n = int((rec[0, :, 10] > 0).sum())
print(f"{n} phases logged; {(rec[0, n - 1, 8] - rec[0, 0, 10]) / 100.0:.1f} us for them ({(rec[0, n - 1, 8] - rec[0, 0, 10]) / 100.0 / n:.2f} us per phase)")
for who, name in ((0, "workgroup 0 (owner of head 0)"), (1, "workgroup 255 (no head)")):
    print(name)
    for kind in dict.fromkeys(kinds):
        sel = [p for p in range(per, n - 1) if kinds[p % per] == kind]     # (skip the first step: its layer 0 takes its embedding from memory)
        r = rec[who]
        dur = np.mean([(r[p + 1, 10] - r[p, 10]) for p in sel]) / 100.0
        def seg(a, b):
            v = [(r[p, b] - r[p, a]) for p in sel if r[p, a] and r[p, b]]
            return np.mean(v) / 100.0 if v else 0.0
        print(f"  {kind:10s} {dur:6.2f} us | start -> hand-off taken {seg(10, 2):5.2f} | -> blocks {seg(2, 5):5.2f} | -> dots {seg(5, 6):5.2f} | -> published {seg(6, 8):5.2f}" if kind != "attention"
              else f"  {kind:10s} {dur:6.2f} us | start -> body {seg(10, 2):5.2f} | body {seg(2, 8):5.2f}")
