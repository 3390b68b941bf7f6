"""Print the logit / sample error statistics of the whole-frame parity runs (HIP vs oracle) per weight type."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import hot_util as hu
import test_hip_frame as t
from ggml_util import BF16, F32, Q4_0, Q4_K, Q8_0
for name, lt, et in [("q4_k", Q4_K, Q4_0), ("q8_0", Q8_0, Q8_0), ("q4_0", Q4_0, Q4_0), ("bf16", BF16, BF16), ("f32", F32, F32)]:
    cfg = hu.hot.tiny(hu.L, linear_type=lt, embed_type=et)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = t.run_lm("oracle", cfg, 16)
    got, _ = t.run_lm("hip", cfg, 16)
    errs = np.array([max(hu.rel_err(a[3], b[3]), hu.rel_err(a[4], b[4])) for a, b in zip(ref, got)])
    tok = all(a[:3] == b[:3] for a, b in zip(ref, got))
    print(f"{name}: tokens_exact={tok} logit rel err max {errs.max():.2e} median {np.median(errs):.2e} p90 {np.quantile(errs, 0.9):.2e}")
