"""-m gpu: whole-frame parity of the streaming-decode hot path, MI355X backend vs the CPU oracle, through
the C-ABI driver (include/moshi_hot.h): same synthetic weights (same seed), same inputs.

Bars (BASELINE.json north_star): greedy token ids bit-exact; logits within 1e-3 (relative to max |logit|);
codec samples within 1e-3 of max |sample| (conv activations go through F16 im2col on both sides)."""
import numpy as np
import pytest

import hot_util as hu
from ggml_util import BF16, F32, Q4_0, Q4_K, Q8_0

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3
PCM_TOL = 1e-3


def run_lm(kind, cfg, steps, seed=3, flags=0):
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    rng = np.random.default_rng(seed)
    rec = []
    n_in = cfg.n_q - cfg.dep_q
    for _ in range(steps):
        ia = rng.integers(0, cfg.card, n_in).tolist()
        r, txt, aud = m.lm_step(ia)
        logits = m.read("text_logits", cfg.text_card)
        dl = m.read(f"dep_logits{cfg.dep_q - 1}", cfg.card)
        rec.append((r, txt, aud, logits, dl))
    st = m.stats() if kind == "hip" else None
    m.free()
    return rec, st


def check_lm(ref, got):
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], f"step {i}: tokens differ: oracle {a[:3]} vs hip {b[:3]}"
        assert hu.rel_err(a[3], b[3]) < LOGIT_TOL, f"step {i}: text logits rel err {hu.rel_err(a[3], b[3]):.2e}"
        assert hu.rel_err(a[4], b[4]) < LOGIT_TOL, f"step {i}: depth logits rel err {hu.rel_err(a[4], b[4]):.2e}"


@pytest.mark.parametrize("lt,et", [(Q4_K, Q4_0), (BF16, BF16), (F32, F32), (Q8_0, Q8_0)])
def test_lm_steps_match_oracle(lt, et):
    cfg = hu.hot.tiny(hu.L, linear_type=lt, embed_type=et)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 6)
    got, st = run_lm("hip", cfg, 6)
    check_lm(ref, got)
    assert st.graph_replays > 0, "cached graphs must replay as hipGraphs"
    if lt in (Q4_K, BF16, F32):
        assert st.fused_nodes_in_last_plan > 0, "fusion matchers did not fire on the Depth graph"


def test_lm_ring_wrap_and_unfused_agree():
    # context 12 < steps 30: the Temporal ring wraps (mask branch offset > capacity, torch.h:211-214)
    cfg = hu.hot.tiny(hu.L, context=12)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 30)
    got, _ = run_lm("hip", cfg, 30)
    check_lm(ref, got)
    plain, st = run_lm("hip", cfg, 30, flags=1 | 2 | 4)   # one kernel per node, no hipGraph, no upload batching
    check_lm(ref, plain)
    assert st.fused_nodes_in_last_plan == 0 and st.graph_replays == 0


def test_sts_frames_match_oracle():
    cfg = hu.hot.tiny(hu.L)
    rng = np.random.default_rng(5)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.1 for _ in range(4)]
    out = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        out[kind] = [m.sts_frame(f) for f in frames]
        m.free()
    for i, (a, b) in enumerate(zip(out["oracle"], out["hip"])):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], f"frame {i}: oracle {a[:3]} vs hip {b[:3]}"
        if a[0]:
            assert hu.rel_err(a[3], b[3]) < PCM_TOL, f"frame {i}: pcm rel err {hu.rel_err(a[3], b[3]):.2e}"


def test_mimi_codec_crosses_t2_mask_quirk():
    # Mimi transformers have T = 2, capacity 250: after 125 frames bias_pattern_index takes its second branch
    # (SURVEY.md §5 quirk). Codes in -> pcm out, 130 frames, decoder only.
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_lm = cfg.enable_mimi_encoder = 0
    rng = np.random.default_rng(9)
    codes = [rng.integers(0, cfg.mimi_codebook_size, cfg.mimi_n_q).tolist() for _ in range(130)]
    pcm = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        pcm[kind] = [m.mimi_decode(c) for c in codes]
        m.free()
    worst = max(hu.rel_err(a, b) for a, b in zip(pcm["oracle"], pcm["hip"]))
    assert worst < PCM_TOL, f"worst pcm rel err {worst:.2e}"


def test_mimi_encoder_codes_exact():
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_lm = cfg.enable_mimi_decoder = 0
    rng = np.random.default_rng(11)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.2 for _ in range(5)]
    codes = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        codes[kind] = [m.mimi_encode(f) for f in frames]
        m.free()
    assert codes["oracle"] == codes["hip"]
