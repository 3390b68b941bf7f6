"""__graft_entry__.smoke(): one tiny decode frame of the hot path on the MI355X device, checked against the oracle."""
import numpy as np

import hot_util as hu


def run():
    cfg = hu.hot.tiny(hu.L, layers=1)
    frames = [np.zeros(1920, np.float32), np.full(1920, 0.01, np.float32), np.zeros(1920, np.float32)]
    out = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        out[kind] = [m.sts_frame(f) for f in frames]
        if kind == "hip":
            st = m.stats()
            assert st.graph_replays > 0 and st.graphs_computed > 0, "the HIP backend did not execute the graphs"
        m.free()
    for a, b in zip(out["oracle"], out["hip"]):
        assert a[:3] == b[:3], f"token mismatch: oracle {a[:3]} vs MI355X {b[:3]}"
        if a[0]:
            assert hu.rel_err(a[3], b[3]) < 1e-2
    # the same frames through bench.py's default loop (two command streams, run-ahead): bit-identical to the serial loop
    cfg.codec_stream, cfg.chain_depth = 1, 2
    m = hu.Model("hip", cfg, seed=0)
    piped = m.sts_pipeline(frames)
    m.free()
    for a, b in zip(out["hip"], piped):
        assert a[:3] == b[:3] and (not a[0] or np.array_equal(a[3], b[3])), f"pipelined loop differs: {a[:3]} vs {b[:3]}"
    print("smoke ok:", [o[:3] for o in out["hip"]])
