"""the Mimi transformer program against one launch per node group: encoder codes and decoder PCM over N frames, MI355X_CHAIN_VERBOSE=1 shows the plan decision"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
L = hu.L
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = hu.hot.moshika(L)
cfg.enable_lm = 0
out = {}
for flags in (32, 16):
    m = hu.Model("hip", cfg, seed=0, flags=flags)
    rng = np.random.default_rng(2)
    rec = []
    for i in range(frames):
        pcm = (rng.standard_normal(1920) * 0.1).astype(np.float32)
        codes = m.mimi_encode(pcm)
        rec.append((codes, m.mimi_decode(codes).copy()))
    st = m.stats()
    print("flags", flags, "programs in last plan", st.chain_step_programs_in_last_plan, "chained", st.chained_matvecs_in_last_plan)
    out[flags] = rec
    m.free()
same_codes = all(a[0] == b[0] for a, b in zip(out[32], out[16]))
same_pcm = all(np.array_equal(a[1], b[1]) for a, b in zip(out[32], out[16]))
print("codes identical", same_codes, "pcm identical", same_pcm, "max pcm diff", max(float(np.abs(a[1] - b[1]).max()) for a, b in zip(out[32], out[16])))
