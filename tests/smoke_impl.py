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
    print("smoke ok:", [o[:3] for o in out["hip"]])
