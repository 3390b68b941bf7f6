"""CPU: the host-side hot-path driver (include/moshi_hot.h) on the host device with the oracle attached —
frame protocol, delay ring, determinism, and the bias-mask lookup semantics incl. the T = 2 quirk."""
import ctypes as C

import numpy as np
import pytest

import ggml_util as gu
import hot_util as hu
from ggml_util import F32


def test_delay_ring_first_frame_produces_nothing_then_frames_flow():
    cfg = hu.hot.tiny(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg)
    rng = np.random.default_rng(0)
    outs = [m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist()) for _ in range(4)]
    m.free()
    assert outs[0][0] == 0                      # offset <= max_delay (lm.h:950)
    assert all(o[0] == 1 for o in outs[1:])
    assert all(0 <= t < cfg.card for o in outs[1:] for t in o[2]) and all(0 <= o[1] < cfg.text_card for o in outs[1:])


def test_driver_is_deterministic_in_seed():
    cfg = hu.hot.tiny(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    runs = []
    for seed in (0, 0, 1):
        m = hu.Model("oracle", cfg, seed=seed)
        runs.append([m.lm_step([1, 2, 3]) for _ in range(3)])
        m.free()
    assert runs[0] == runs[1] and runs[0] != runs[2]


def test_codec_round_trip_runs_and_is_finite():
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_lm = 0
    m = hu.Model("oracle", cfg)
    rng = np.random.default_rng(1)
    for _ in range(2):
        codes = m.mimi_encode(rng.standard_normal(1920).astype(np.float32) * 0.1)
        assert len(codes) == cfg.mimi_n_q and all(0 <= c < cfg.mimi_codebook_size for c in codes)
        pcm = m.mimi_decode(codes)
        assert pcm.shape == (1920,) and np.isfinite(pcm).all() and np.abs(pcm).max() > 0
    m.free()


def visible_slots(C, T, offset):
    """Which ring slots each of the T query rows may attend at stream position `offset`, read from the graph the
    driver/reference build: cont(view_2d(pattern, C, T, nb1, col)) (torch.h:170-223), executed by the oracle."""
    start, width = 2 * C - T, 3 * C - T
    pat = np.zeros((T, width), np.float32)
    for j in range(T):
        pat[j, start + 1 + j:] = -np.inf
        for i in range(T - j - 1):
            pat[j, C - 1 - i] = -np.inf
    col = start - offset if offset <= C else C - (offset % C)

    def build(g):
        p = g.input(pat)
        v = g.view_2d(p, C, T, p.contents.nb[1], col * 4)
        return [g.cont(v)]
    (m,), _ = gu.run_graph("oracle", build)
    return [set(np.nonzero(np.isfinite(m.reshape(T, C)[j]))[0].tolist()) for j in range(T)]


def test_mask_t1_is_filled_prefix_then_everything():
    C = 12
    for off in range(0, 40):
        vis = visible_slots(C, 1, off)[0]
        assert vis == (set(range(off + 1)) if off < C else set(range(C))), off


def test_mask_t2_causal_before_wrap_and_reference_quirk_after():
    C = 10
    for off in range(0, C - 1, 2):              # chunk occupies slots off, off+1
        v0, v1 = visible_slots(C, 2, off)
        assert v0 == set(range(off + 1)) and v1 == set(range(off + 2)), off
    # once offset > C the intra-chunk mask is shifted two columns left: query row 0 masks the slot
    # before the chunk and sees the slot after it. Parity means reproducing this, not fixing it.
    off = C + 4
    s = off % C
    v0, v1 = visible_slots(C, 2, off)
    assert v1 == set(range(C))
    assert v0 == set(range(C)) - {(s - 1) % C}


# ---- the frame protocol in full (moshi_lmgen_step, lm.h:778-979) against a pure-Python restatement of its delay ring -------------
class RingModel:
    """Host-side integer logic of moshi_lmgen_step, restated from lm.h:778-979 / 722-743: what goes into the model and what comes out,
    given the raw samples of every frame. The device / oracle supplies only the samples."""

    def __init__(self, cfg):
        self.ncb = cfg.n_q + 1
        self.delays = [cfg.delays[i] for i in range(self.ncb)]
        self.max_delay = max(self.delays)
        self.lm_dep_q = cfg.dep_q
        self.dep_q = 8 if cfg.personaplex else cfg.dep_q
        self.CT = self.max_delay + 2 + (1 if cfg.personaplex else 0)
        self.cache = [[-2] * self.ncb for _ in range(self.CT)]
        self.initial = [cfg.text_card] + [cfg.card] * (self.ncb - 1)
        self.offset = 0

    def inputs(self, tokens):
        needed = self.ncb - self.dep_q - 1
        self.provided = False
        if needed > 0:
            if len(tokens) == self.ncb:
                for i in range(self.ncb):
                    self.cache[(self.offset + self.delays[i]) % self.CT][i] = tokens[i]
                self.provided = True
            else:
                for i in range(needed):
                    k = self.dep_q + 1 + i
                    self.cache[(self.offset + self.delays[k]) % self.CT][k] = tokens[i]
        pos = self.offset % self.CT
        return [self.initial[i] if self.offset <= self.delays[i] else self.cache[pos][i] for i in range(self.ncb)]

    def outputs(self, text, audio):
        audio = list(audio)
        self.offset += 1
        if not self.provided:
            pos = self.offset % self.CT
            self.cache[pos][0] = text
            for q in range(self.lm_dep_q):
                self.cache[pos][q + 1] = audio[q]
        if self.offset <= self.max_delay:
            return 0, None, None
        t = self.cache[(self.offset - self.max_delay + self.delays[0]) % self.CT][0]
        for i in range(1, self.dep_q + 1):
            audio[i - 1] = self.cache[(self.offset - self.max_delay + self.delays[i]) % self.CT][i]
        if any(x == -1 for x in audio):
            return 0, t, audio[:self.dep_q]
        return 1, t, audio[:self.dep_q]


def embedding_ids(m, cfg):
    """what the Temporal graph was fed: ids / scales uploaded by moshi_lmmodel_text_token_embed_step"""
    g = hu.L.moshi_hot_graph(m.m, 0)
    ids = []
    for i in range(hu.L.ggml_graph_n_nodes(g)):
        t = hu.L.ggml_graph_node(g, i)
        if hu.L.ggml_op_name(t.contents.op) == b"GET_ROWS" and len(ids) < cfg.n_q + 1:
            idx = np.zeros(1, np.int32)
            hu.L.ggml_backend_tensor_get(t.contents.src[1], idx.ctypes.data, 0, 4)
            ids.append(int(idx[0]))
    return ids


@pytest.mark.parametrize("variant", ["moshika", "personaplex"])
def test_frame_protocol_matches_ring_restatement(variant):
    cfg = hu.hot.tiny(hu.L, layers=1) if variant == "moshika" else hu.hot.tiny_personaplex(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg)
    ring = RingModel(cfg)
    rng = np.random.default_rng(3)
    n_in = cfg.n_q - cfg.io_dep_q
    for step in range(9):
        full = variant == "personaplex" and step in (2, 3, 6)        # "provided" prompt frames between ordinary ones
        tokens = rng.integers(0, cfg.card, cfg.n_q + 1 if full else n_in).tolist()
        want_in = ring.inputs(tokens)
        r, txt, aud = m.lm_step_n(tokens)
        got_in = embedding_ids(m, cfg)
        assert got_in == [max(t, 0) for t in want_in], f"step {step}: model inputs {got_in} vs {want_in}"
        raw_t, raw_a = m.last_raw()
        assert len(raw_a) == cfg.dep_q
        wr, wt, wa = ring.outputs(raw_t, raw_a)
        assert r == wr, f"step {step}: produced {r} vs {wr}"
        if r:
            assert (txt, aud) == (wt, wa), f"step {step}"
    m.free()


def test_personaplex_depth_chain_is_16_steps_over_a_ring_of_8():
    cfg = hu.hot.tiny_personaplex(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    assert (cfg.dep_q, cfg.dep_context, cfg.io_dep_q, cfg.n_q) == (16, 8, 8, 16)      # tools/personaplex-config.json, lm.h:803-805
    m = hu.Model("oracle", cfg)
    r = [m.lm_step(list(range(8))) for _ in range(3)]
    raw = m.last_raw()
    assert len(raw[1]) == 16 and all(0 <= t < cfg.card for t in raw[1])
    assert [x[0] for x in r] == [0, 1, 1] and all(len(x[2]) == 8 for x in r)
    m.free()


def test_personaplex_system_prompt_frames_step_the_model_but_leave_predictions_out_of_the_ring():
    cfg = hu.hot.tiny_personaplex(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    prompt = [int(x) for x in hu.L.moshi_hot_personaplex_prompt_tokens()[0:17]]
    assert prompt == [3, 948, 243, 1178, 546, 1736, 1030, 1978, 2008, 430, 1268, 381, 1611, 1095, 1495, 56, 472]   # PROMPT_TOKENS (lm.h:983-987)
    m = hu.Model("oracle", cfg)
    m.system_prompts([11, 12, 13])
    assert hu.L.moshi_hot_offset(m.m) == 6 + 3 + 6                     # lm.h:1118-1134
    # the frame after the prompts reads the prompt's tokens from the ring (delay-0 streams) — not the model's own samples
    ring = RingModel(cfg)
    for text in [3] * 6 + [11, 12, 13] + [3] * 6:
        ring.inputs([text] + prompt[1:])
        ring.outputs(-5, [-5] * 16)                                     # predictions must never show up
    want = ring.inputs(list(range(8)))
    m.lm_step(list(range(8)))
    assert embedding_ids(m, cfg) == [max(t, 0) for t in want] and -5 not in want
    m.free()


def test_voice_prompt_embedding_frame_equals_a_token_frame_with_the_same_embedding_sum():
    cfg = hu.hot.tiny_personaplex(hu.L, layers=2)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    a, b = hu.Model("oracle", cfg), hu.Model("oracle", cfg)
    for step in range(3):                                               # moshi_lmmodel_forward_embedding on the scratch context (lm.h:694-709, 1004-1037)
        a.lm_step(list(range(step, step + 8)))
        emb = a.read("transformer_in", cfg.dim)
        b.lm_step_embedding(emb)
        assert np.array_equal(a.read("transformer_out", cfg.dim), b.read("transformer_out", cfg.dim)), f"step {step}"
    assert hu.L.moshi_hot_offset(b.m) == 3
    a.free(); b.free()


def test_vad_head_is_softmax_of_extra_head_2():
    cfg = hu.hot.tiny(hu.L, layers=1, linear_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.extra_heads, cfg.extra_heads_dim = 3, 6                         # stt: lm_default.h:211-213, config.h:158
    m = hu.Model("oracle", cfg)
    rng = np.random.default_rng(0)
    n_in = cfg.n_q - cfg.dep_q
    r = m.lm_step_n(rng.integers(0, cfg.card, n_in).tolist(), vad=True)
    assert r[0] == 0 and r[3] == -1.0                                    # nothing produced yet: vad untouched (lm.h:950, 966)
    r = m.lm_step_n(rng.integers(0, cfg.card, n_in).tolist(), vad=True)
    assert r[0] == 1
    w = np.zeros((6, cfg.dim), np.float32)
    t = C.cast(hu.L.moshi_hot_weight(m.m, b"lm.extra_heads.2.weight"), hu.pkg.TP)
    hu.L.ggml_backend_tensor_get(t, w.ctypes.data, 0, w.nbytes)
    z = w.astype(np.float64) @ m.read("transformer_out", cfg.dim).astype(np.float64)
    p = np.exp(z - z.max()); p /= p.sum()
    assert abs(r[3] - p[0]) < 1e-5
    m.free()
    cfg.extra_heads = 0
    m = hu.Model("oracle", cfg)
    m.lm_step_n([1] * n_in)
    assert m.lm_step_n([1] * n_in, vad=True)[3] == 0.0                   # no heads: *vad = 0 (lm.h:973-975)
    m.free()
