"""Soak of the two-stream run-ahead frame loop at the full moshika configuration: N frames (default 3 300: across the Temporal ring's wrap at 3 000 and the
codec rings' at 250) of random input through the serial loop and through moshi_hot_sts_pipeline_*; every token and every PCM sample must be identical.
A second pipelined run checks run-to-run determinism."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3300
rng = np.random.default_rng(5)
frames = [(rng.standard_normal(1920) * 0.05).astype(np.float32) for _ in range(n)]
def digest(res):
    h = hashlib.sha256()
    for r in res:
        h.update(np.array([r[0], r[1]] + list(r[2]), np.int64).tobytes()); h.update(np.ascontiguousarray(r[3]).tobytes())
    return h.hexdigest()[:16]
out = []
for mode in ("serial", "run-ahead", "run-ahead"):
    cfg = hu.hot.personaplex(L) if os.environ.get("MODEL") == "personaplex" else hu.hot.moshika(L)
    if os.environ.get("CONTEXT"):
        cfg.context = int(os.environ["CONTEXT"])
    if os.environ.get("SAMPLED"):          # the reference's --bench temperatures
        cfg.temp, cfg.temp_text = 0.8, 0.7
    cfg.codec_stream, cfg.chain_depth = (0, 0) if mode == "serial" else (1, 2)
    m = hu.Model("hip", cfg, seed=0)
    if os.environ.get("SAMPLED"):          # host rand() reseeded per run, AFTER the model (and with it the HIP runtime, which draws from rand() while it starts) is up
        import ctypes
        ctypes.CDLL(None).srand(4321)
    t0 = time.perf_counter()
    res = m.sts_pipeline(frames) if mode != "serial" else [m.sts_frame(f) for f in frames]
    dt = time.perf_counter() - t0
    m.free()
    out.append(res)
    print(f"{mode:10s}: {n / dt:6.1f} frames/s, digest {digest(res)}", flush=True)
bad = [i for i, (a, b) in enumerate(zip(out[0], out[1])) if a[:3] != b[:3] or not np.array_equal(a[3], b[3])]
bad2 = [i for i, (a, b) in enumerate(zip(out[1], out[2])) if a[:3] != b[:3] or not np.array_equal(a[3], b[3])]
print("serial vs run-ahead: first differing frames", bad[:5], "| run-ahead vs run-ahead:", bad2[:5])
assert not bad and not bad2
print("identical")
