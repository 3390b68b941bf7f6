import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
import test_hip_frame as T
from ggml_util import F32
cfg = hu.hot.tiny_tts(hu.L, linear_type=F32)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
ref, _ = T.run_tts("oracle", cfg, 4)
for name, flags in (("fused", 0), ("per-node", 7)):
    got, _ = T.run_tts("hip", cfg, 4, flags=flags, forced=ref)
    for i, (a, b) in enumerate(zip(ref, got)):
        print(name, i, "text logits", f"{hu.rel_err(a[2], b[2]):.1e}", "transformer_out", f"{hu.rel_err(a[3], b[3]):.1e}", "raw", a[1] == b[1],
              "dep", " ".join(f"{hu.rel_err(x, y):.0e}" for x, y in zip(a[4], b[4])) if a[4] else "-")
