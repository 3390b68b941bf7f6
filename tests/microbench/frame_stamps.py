"""In-situ timing of every block mat-vec of ONE graph-replayed frame (needs the -DMV_LOG build: tests/microbench/build_stamped_lib.sh, then
MI355X_LIB=tests/microbench/ab/libggml-mi355x-log.so python tests/microbench/frame_stamps.py). Workgroup 0 / wave 0 of every launch logs its phase
stamps (s_memtime cycles) and its start / end in s_memrealtime ticks (10 ns): prints per shape the in-kernel phases and the gap since the
previous mat-vec ended (which contains whatever other kernels ran in between)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
cfg = hu.hot.moshika(L)
m = hu.Model("hip", cfg, seed=0)
if os.environ.get("FILL"):
    L.moshi_hot_set_context_fill(m.m, int(os.environ["FILL"]))     # start from a ring that is already this full
pcm = np.zeros(1920, np.float32)
for _ in range(12):
    m.sts_frame(pcm)
lib = L.ggml
lib.mi355x_mv_log_read.restype = C.c_int
buf = (C.c_ulonglong * (8192 * 24))()
lib.mi355x_mv_log_read(buf, 8192)       # reset
m.sts_frame(pcm)
L.ggml_backend_synchronize(m.be)
n = lib.mi355x_mv_log_read(buf, 8192)
rec = np.frombuffer(buf, np.uint64).reshape(8192, 24)[:n].astype(np.int64)
rec = rec[np.argsort(rec[:, 16])]
t0 = rec[0, 16]
names = {0: "plain", 1: "rms", 2: "gate", 5: "preq", 6: "attn", 0xA7: "ATTN"}
rows = []
prev_end = None
for r in rec:
    K, M = int(r[18] & 0xffffffff), int(r[18] >> 32)
    pro, ws, grid = int(r[19] & 0xff), int((r[19] >> 8) & 0xff), int(r[19] >> 32)
    start, end = (r[16] - t0) / 100.0, (r[17] - t0) / 100.0
    ph = [(int(r[i]) - int(r[0])) if r[i] else 0 for i in range(1, 16)]
    rows.append((K, M, pro, ws, grid, start, end - start, (start - prev_end) if prev_end is not None else 0.0, ph))
    prev_end = end
print(f"{n} mat-vec launches in the frame; span {rows[-1][5] + rows[-1][6]:.0f} us")
from collections import defaultdict
agg = defaultdict(list)
for r in rows:
    agg[r[:5]].append(r)
print("   K      M  pro ws grid    n | in-kernel us (block 0) | gap before us | phase cycles: loads-issued, prologue, sync, staged, tiles, sync, end")
for k, v in sorted(agg.items(), key=lambda kv: -len(kv[1]) * np.mean([x[6] + x[7] for x in kv[1]])):
    dur = np.median([x[6] for x in v]); gap = np.median([x[7] for x in v]); ph = np.median(np.array([x[8] for x in v]), axis=0)
    print(f"{k[0]:5d} {k[1]:6d} {names.get(k[2], k[2]):>5} {k[3]:2d} {k[4]:4d} {len(v):4d} | {dur:8.2f}              | {gap:8.2f}      | " + " ".join(f"{int(p):6d}" for p in ph[:7]) + (" | extra " + " ".join(f"{int(p):6d}" for p in ph[7:11]) if k[2] == 0xA7 else " | x arrived, sumsq in, quantiser starts " + " ".join(f"{int(p):6d}" for p in ph[7:10])))
tot_k = sum(x[6] for x in rows); tot_g = sum(x[7] for x in rows)
print(f"sum of in-kernel {tot_k:.0f} us, sum of gaps {tot_g:.0f} us")
if len(sys.argv) > 1:
    for r in rows[:int(sys.argv[1])]:
        print(r)
# the folded attention + out_proj launches (attn_outproj_kernel): workgroup 0's s_memrealtime stamps - start, attention stage done, K values gathered,
# quantised, dots done, end
if hasattr(lib, "mi355x_fold_log_read"):
    lib.mi355x_fold_log_read.restype = C.c_int
    fb = (C.c_ulonglong * (4096 * 8))()
    nf = lib.mi355x_fold_log_read(fb, 4096)
    if nf > 0:
        fr = np.frombuffer(fb, np.uint64).reshape(4096, 8)[:min(nf, 4096)].astype(np.int64)
        fr = fr[-32:]                                    # the last frame's launches
        d = np.diff(fr[:, :6], axis=1) / 100.0
        print(f"{nf} merged launches logged; last 32: in-kernel us (workgroup 0) median of the five stage durations (attn_outproj: attention, gather, quantise, dots, row sums; inproj_attn: prologue, tiles, row sums + publish, attention, -): {np.round(np.median(d, axis=0), 2)}  total {np.median(fr[:,5]-fr[:,0])/100.0:.2f}")
        print("   launch-to-launch period of the folded kernel (us):", np.round(np.median(np.diff(fr[:, 0])) / 100.0, 2))
