"""In-kernel timeline of the persistent chain engine (needs the -DCH_LOG build: tests/microbench/build_stamped_lib.sh, then
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/chain_stamps.py). Wave 0 of the first and last workgroup stamp
s_memrealtime (10 ns ticks) at the stages of every phase of the last Depth launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L) if os.environ.get("MODEL", "moshika") == "moshika" else hu.hot.personaplex(L)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
m = hu.Model("hip", cfg, seed=0)
rng = np.random.default_rng(0)
for _ in range(8):
    m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
L.ggml_backend_synchronize(m.be)
lib = L.ggml
buf = (C.c_ulonglong * (2 * 512 * 16))()
assert lib.mi355x_chain_log_read(buf) == 0
rec = np.frombuffer(buf, np.uint64).reshape(2, 512, 16).astype(np.int64)
names = ["start", "pre-loads", "gathered", "attention", "blocks", "barrier", "dots+req", "dot sync", "published"]
# The step programs stamp 3 and 5 around norm + Q8_K quantiser and have no stamp 4 (the generic chain kernel splits the stage there): the missing stamp is set to
# stamp 5, so "blocks" carries the whole norm + quantiser stage (rounds 5's tables showed 0.00 / 0.00 there and left ~0.75 us of every phase unattributed).
fix = (rec[:, :, 4] == 0) & (rec[:, :, 3] > 0) & (rec[:, :, 5] > 0)
rec[:, :, 4] = np.where(fix, rec[:, :, 5], rec[:, :, 4])
if fix.any(): names[4] = "norm+q8k"
hoist = rec[0, 511].copy()   # the step program logs its hoisted depformer_in phase into the last record: 10 = entry, 0 = products published, 9 = everybody's gathered
rec[:, 511] = 0
n = int((rec[0, :, 0] > 0).sum())
t0 = rec[0, 0, 0]
if hoist[10]:
    print(f"step program: hoisted depformer_in phase {(hoist[0] - hoist[10]) / 100.0:.2f} us to publication + {(hoist[9] - hoist[0]) / 100.0:.2f} us gather; first phase starts {(rec[0, 0, 10] - hoist[9]) / 100.0:.2f} us later")
print(f"{n} phases; launch span (workgroup 0) {(rec[0, n - 1, 8] - t0) / 100.0:.1f} us")
per = int(os.environ.get("PER", "25" if hoist[10] else "26"))
show = int(sys.argv[1]) if len(sys.argv) > 1 else 2 * per
print("phase |  start us | " + " | ".join(f"{x:>9s}" for x in names[1:]) + " | last wg start-lag")
for p in range(min(n, show)):
    r = rec[0, p]
    d = [(r[i] - r[i - 1]) / 100.0 if r[i] and r[i - 1] else 0.0 for i in range(1, 9)]
    print(f"{p:5d} | {(r[0] - t0) / 100.0:9.2f} | " + " | ".join(f"{x:9.2f}" for x in d) + f" | {(rec[1, p, 0] - r[0]) / 100.0:8.2f}")
dur = np.array([(rec[0, p + 1, 0] - rec[0, p, 0]) / 100.0 for p in range(n - 1)])
stage = np.array([[(rec[0, p, i] - rec[0, p, i - 1]) / 100.0 if rec[0, p, i] and rec[0, p, i - 1] else 0.0 for i in range(1, 9)] for p in range(n)])
head = np.array([(rec[0, p, 0] - rec[0, p, 10]) / 100.0 for p in range(n)])
tail = np.array([(rec[0, p, 9] - rec[0, p, 8]) / 100.0 for p in range(n)])
between = np.array([(rec[0, p + 1, 10] - rec[0, p, 9]) / 100.0 for p in range(n - 1)])
print(f"descriptor reads at the head of a phase {head.mean():.2f} us, arg-max tail {tail.mean():.2f} us, between phases (kind dispatch) {between.mean():.2f} us")
print("mean per phase us:", f"{dur.mean():.2f}", "| by stage:", " ".join(f"{names[i + 1]}={stage[:, i].mean():.2f}" for i in range(8)))
for k in range(per):
    sel = np.arange(k, n - 1, per)
    print(f"  phase {k:2d} of a step: {dur[sel].mean():6.2f} us  stages " + " ".join(f"{stage[sel, i].mean():5.2f}" for i in range(8)))


# inside the attention of an attention phase (wave 0 of workgroup 0): gathered -> [barrier] -> entry, RoPE + staging, both heads' chains to their products (interleaved
# since round 4; one after the other they were 0.61 + 0.58 us), 8-slot sums -> [barrier] -> attention done
att = [p for p in range(n) if rec[0, p, 11] > 0]
if att:
    a = np.array([[(rec[0, p, 11] - rec[0, p, 2]), (rec[0, p, 12] - rec[0, p, 11]), (rec[0, p, 13] - rec[0, p, 12]), (rec[0, p, 14] - rec[0, p, 13]), (rec[0, p, 15] - rec[0, p, 14]), (rec[0, p, 3] - rec[0, p, 15])] for p in att]) / 100.0
    print(f"attention phases ({len(att)}), us: barrier in {a[:, 0].mean():.2f} | RoPE + staging {a[:, 1].mean():.2f} | both heads -> products {a[:, 2].mean():.2f} | (-) {a[:, 3].mean():.2f} | slot sums {a[:, 4].mean():.2f} | barrier out {a[:, 5].mean():.2f}")
