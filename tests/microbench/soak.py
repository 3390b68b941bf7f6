"""Determinism / stability soak at the full moshika configuration: N frames of the sts loop twice (fresh model each time, optionally starting
near the ring wrap); token streams and a PCM checksum must be identical, and no bounded device wait may have fired."""
import os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
fill = int(sys.argv[2]) if len(sys.argv) > 2 else 2800
cfg = hu.hot.personaplex(hu.L) if len(sys.argv) > 3 and sys.argv[3] == "personaplex" else hu.hot.moshika(hu.L)
if len(sys.argv) > 4:
    cfg.linear_type = {"q8_0": 8, "q4_0": 2, "q4_k": 12}[sys.argv[4]]
runs = []
for rep in range(2):
    m = hu.Model("hip", cfg, seed=0)
    hu.L.moshi_hot_set_context_fill(m.m, fill)
    rng = np.random.default_rng(3)
    toks, crc, crcs = [], 0, []
    t0 = time.perf_counter()
    for i in range(n):
        r, txt, aud, pcm = m.sts_frame((rng.standard_normal(1920) * 0.05).astype(np.float32))
        toks.append((r, txt, tuple(aud)))
        crc = zlib.crc32(pcm.tobytes(), crc)
        crcs.append(zlib.crc32(pcm.tobytes()))
    dt = time.perf_counter() - t0
    runs.append((toks, crc, crcs))
    print(f"run {rep}: {n} frames from ring offset {fill} (capacity {cfg.context}) in {dt:.2f} s = {n / dt:.1f} frames/s incl. host-side noise generation; pcm crc {crc:08x}", flush=True)
    m.free()
if runs[0] != runs[1]:
    ft = next((i for i, (a, b) in enumerate(zip(runs[0][0], runs[1][0])) if a != b), None)
    fp = next((i for i, (a, b) in enumerate(zip(runs[0][2], runs[1][2])) if a != b), None)
    print(f"first differing tokens at frame {ft}: {runs[0][0][ft] if ft is not None else None} vs {runs[1][0][ft] if ft is not None else None}; first differing pcm at frame {fp}")
assert runs[0] == runs[1], "two identical runs differ"
print("identical token streams and PCM; no device error raised")
