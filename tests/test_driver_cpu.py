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


@pytest.mark.parametrize("chain,model", [(0, "tiny"), (1, "tiny"), (2, "tiny"), (2, "personaplex")])
def test_pipelined_frame_loop_equals_the_serial_loop_on_the_host_device(chain, model):
    # moshi_hot_sts_pipeline_* (include/moshi_hot.h): call k steps the LM on frame k, decodes frame k - 1 and encodes frame k + 1. On the host device
    # there is one stream (codec_stream is ignored), which pins the protocol itself: outputs arrive one call later, nothing else changes.
    # chain 1: the Depth graph reads the text token from the device-side token state, the next step's inputs are staged early.
    # chain 2: run-ahead - the Temporal graph takes the previous step's samples from that state too, step k is queued before step k - 1 is read.
    rng = np.random.default_rng(3)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.1 for _ in range(8)]
    # PersonaPlex: the Depth chain samples the other speaker's codebooks too and those samples, not the received codes, are what the next step embeds
    # for every delayed column (lm.h:819-824, 935-943 in that order): under run-ahead exactly those columns stay on the device
    cfg = hu.hot.tiny(hu.L, layers=1) if model == "tiny" else hu.hot.tiny_personaplex(hu.L, layers=1)
    m = hu.Model("oracle", cfg)
    serial = [m.sts_frame(f) for f in frames]
    m.free()
    cfg.codec_stream = 1
    cfg.chain_depth = chain
    m = hu.Model("oracle", cfg)
    piped = m.sts_pipeline(frames)
    m.free()
    assert serial[0][0] == 0 and sum(a[0] for a in serial) >= len(serial) - 3
    for a, b in zip(serial, piped):
        assert a[:3] == b[:3]
        if a[0]:
            assert np.array_equal(a[3], b[3])
    # a staged step is taken back by whatever else moves the Temporal stream position: chained steps, then a prefill, then steps again
    cfg3 = hu.hot.tiny_personaplex(hu.L, layers=1)
    prompt = [[int(rng.integers(0, cfg3.text_card))] + rng.integers(0, cfg3.card, cfg3.n_q).tolist() for _ in range(3)]
    seqs = []
    for ch in (0, 1, 2):
        cfg3.chain_depth = ch
        m = hu.Model("oracle", cfg3)
        o = [m.lm_step([1] * 8) for _ in range(2)]
        m.prefill(prompt, 2)
        o += [m.lm_step([2] * 8) for _ in range(3)]
        seqs.append(o)
        m.free()
    assert seqs[0] == seqs[1] == seqs[2]


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
        self.delay_steps = cfg.delay_steps

    def replace(self):
        return self.offset < self.delay_steps          # depformer_replace_tokens (src/moshi.cpp:905)

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
        replaced = self.replace()
        if replaced:
            audio = [-1] * self.lm_dep_q                # lm.h:910-913
        if self.delay_steps:                            # lm.h:915-921
            audio = [-1 if self.offset < self.delays[q + 1] + self.delay_steps else a for q, a in enumerate(audio)]
        self.offset += 1
        if not self.provided:
            pos = self.offset % self.CT
            self.cache[pos][0] = text
            for q in range(self.lm_dep_q):
                self.cache[pos][q + 1] = audio[q]
        if self.offset <= self.max_delay or replaced:
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


# ---- tts branches (BASELINE.json configs[1]) ----------------------------------------------------------------
def demux_token(cfg, first, second):
    return (second + 1) * (cfg.text_card + 1) + first      # lm.h:176-191


def weight_f32(m, name, shape):
    w = np.zeros(shape, np.float32)
    t = C.cast(hu.L.moshi_hot_weight(m.m, name.encode()), hu.pkg.TP)
    assert t, name
    hu.L.ggml_backend_tensor_get(t, w.ctypes.data, 0, w.nbytes)
    return w


def test_tts_frame_protocol_delay_steps_and_text_hook():
    cfg = hu.hot.tiny_tts(hu.L, layers=1, linear_type=F32, embed_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg)
    hu.set_conditions(m, cfg)
    forced = [demux_token(cfg, 7 + i, i - 1) for i in range(12)]           # second = -1 on the first frame: right half scaled by 0
    seen = []
    hu.set_text_hook(m, lambda offset, sampled: (seen.append((offset, sampled)), forced[offset])[1])
    ring = RingModel(cfg)
    produced = []
    for step in range(12):
        want_in = ring.inputs([])
        r, txt, aud = m.lm_step_n([])
        raw_t, raw_a = m.last_raw()
        assert raw_t == forced[step]                                       # the hook's token is what goes on (lm.h:880-900)
        if step < cfg.delay_steps:
            assert raw_a == [-1] * cfg.dep_q                               # Depth skipped (src/moshi.cpp:905, lm.h:910-913)
        wr, wt, wa = ring.outputs(raw_t, raw_a)
        assert r == wr, f"step {step}"
        if r:
            assert (txt, aud) == (wt, wa)
        produced.append(r)
        # what the model was fed: text id split into (left, right) by the demux step
        g = hu.L.moshi_hot_graph(m.m, 0)
        ids = []
        for i in range(hu.L.ggml_graph_n_nodes(g)):
            t = hu.L.ggml_graph_node(g, i)
            if hu.L.ggml_op_name(t.contents.op) == b"GET_ROWS" and len(ids) < 2:
                idx = np.zeros(1, np.int32)
                hu.L.ggml_backend_tensor_get(t.contents.src[1], idx.ctypes.data, 0, 4)
                ids.append(int(idx[0]))
        tok = max(want_in[0], 0)
        n = cfg.text_card + 1
        assert ids == [tok % n, max(tok // n - 1, 0)], f"step {step}: demux ids {ids} for token {tok}"
    assert [o for o, _ in seen] == list(range(12))
    # nothing comes out until every stream is past delays[q + 1] + delay_steps and the delay ring has turned over
    first = produced.index(1)
    assert first >= cfg.delay_steps + max(cfg.delays[i] for i in range(cfg.n_q + 1)) and all(produced[first:])
    m.free()


def test_tts_embedding_sum_is_demux_plus_audio_rows_plus_condition():
    cfg = hu.hot.tiny_tts(hu.L, layers=1, linear_type=F32, embed_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg)
    cond_sum, _ = hu.set_conditions(m, cfg)
    first, second = 11, 5
    toks = [demux_token(cfg, first, second), demux_token(cfg, 3, -1)]
    hu.set_text_hook(m, lambda offset, sampled: toks[offset])
    m.lm_step_n([])
    m.lm_step_n([])          # frame 1 reads frame 0's text token from the ring (delay 0); audio streams still at their initial / -1 tokens
    E = weight_f32(m, "lm.text_emb.weight", (cfg.text_card + 1, cfg.dim)).astype(np.float64)
    o1 = weight_f32(m, "lm.text_emb.out1.weight", (cfg.dim, cfg.dim)).astype(np.float64)
    o2 = weight_f32(m, "lm.text_emb.out2.weight", (cfg.dim, cfg.dim)).astype(np.float64)
    want = o1 @ E[first] + o2 @ E[second] + cond_sum
    for k in range(cfg.n_q):           # offset 1: streams with delay >= 1 feed their initial token (row `card`), delay 0 the ring (-1 -> scale 0)
        if 1 <= cfg.delays[k + 1]:
            want += weight_f32(m, f"lm.emb.{k}.weight", (cfg.card + 1, cfg.dim))[cfg.card]
    got = m.read("transformer_in", cfg.dim)
    assert np.max(np.abs(got - want)) < 1e-4 * np.max(np.abs(want))
    m.free()


def test_tts_cross_attention_reads_the_condition():
    cfg = hu.hot.tiny_tts(hu.L, layers=1, linear_type=F32, embed_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    outs = []
    for seed in (4, 4, 5):
        m = hu.Model("oracle", cfg)
        hu.set_conditions(m, cfg, seed=seed)
        hu.set_text_hook(m, lambda offset, sampled: 9)
        m.lm_step_n([])
        outs.append(m.read("transformer_out", cfg.dim).copy())
        m.free()
    assert np.array_equal(outs[0], outs[1]) and not np.allclose(outs[0], outs[2])


def test_tts_cross_attention_update_matches_numpy_restatement():
    # moshi_streaming_multihead_cross_attention + init() (transformer.h:343-396, 714-762) restated in numpy from the module's definition:
    # q = W[:dim]·LN(x); (k | v) = W[dim:]·cond, "b t (p h d) -> p b h t d"; softmax(q·k/sqrt(D)) over the Tc condition rows; out_proj
    cfg = hu.hot.tiny_tts(hu.L, layers=1, linear_type=F32, embed_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg)
    _, cond = hu.set_conditions(m, cfg)
    hu.set_text_hook(m, lambda offset, sampled: 9)
    m.lm_step_n([])
    dim, H = cfg.dim, cfg.num_heads
    D = dim // H
    W = weight_f32(m, "lm.transformer.layers.0.cross_attention.in_projs.0.weight", (3 * dim, dim)).astype(np.float64)
    Wo = weight_f32(m, "lm.transformer.layers.0.cross_attention.out_projs.0.weight", (dim, dim)).astype(np.float64)
    lw = weight_f32(m, "lm.transformer.layers.0.norm_cross.weight", (dim,)).astype(np.float64)
    lb = weight_f32(m, "lm.transformer.layers.0.norm_cross.bias", (dim,)).astype(np.float64)
    wo_ptr = hu.L.moshi_hot_weight(m.m, b"lm.transformer.layers.0.cross_attention.out_projs.0.weight")
    g = hu.L.moshi_hot_graph(m.m, 0)

    def value(t):
        a = np.zeros(hu.L.ggml_nelements(t), np.float32)
        hu.L.ggml_backend_tensor_get(t, a.ctypes.data, 0, a.nbytes)
        return a.astype(np.float64)

    out_node = norm_node = None
    for i in range(hu.L.ggml_graph_n_nodes(g)):
        t = hu.L.ggml_graph_node(g, i)
        if hu.L.ggml_op_name(t.contents.op) == b"NORM":
            norm_node = t
        if hu.L.ggml_op_name(t.contents.op) == b"MUL_MAT" and C.cast(t.contents.src[0], C.c_void_p).value == wo_ptr:
            out_node = t
    assert out_node and norm_node
    x = value(norm_node.contents.src[0])                      # the residual stream entering the cross-attention block
    mu, var = x.mean(), x.var()
    n = (x - mu) / np.sqrt(var + 0.0) * lw + lb               # LayerNorm, eps 0.0 (lm_default.h:34)
    q = (W[:dim] @ n).reshape(H, D)
    kv = cond.astype(np.float64) @ W[dim:].T                  # [Tc, 2 dim]
    k = kv[:, :dim].reshape(-1, H, D)
    v = kv[:, dim:].reshape(-1, H, D)
    o = np.zeros((H, D))
    for h in range(H):
        s = k[:, h] @ q[h] / np.sqrt(D)
        p = np.exp(s - s.max()); p /= p.sum()
        o[h] = p @ v[:, h]
    want = Wo @ o.reshape(dim)
    got = value(out_node)
    assert np.max(np.abs(got - want)) < 1e-4 * np.max(np.abs(want)), np.max(np.abs(got - want)) / np.max(np.abs(want))
    m.free()


# ---- batched prompt prefill -----------------------------------------------------------------------------------
@pytest.mark.parametrize("variant,chunk", [("moshika", 0), ("moshika", 3), ("personaplex", 4)])
def test_batched_prefill_leaves_the_state_of_frame_by_frame_provided_steps(variant, chunk):
    cfg = hu.hot.tiny(hu.L, layers=2) if variant == "moshika" else hu.hot.tiny_personaplex(hu.L, layers=2)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    rng = np.random.default_rng(21)
    n = 10
    frames = [[int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist() for _ in range(n)]
    after = [rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist() for _ in range(4)]
    a, b = hu.Model("oracle", cfg), hu.Model("oracle", cfg)
    a.lm_step(after[0]); b.lm_step(after[0])                    # not from a cold start: one ordinary frame first
    for f in frames:
        a.lm_step_n(f)                                          # frame by frame, as the reference steps its prompts (lm.h:1063-1134)
    b.prefill(frames, chunk)
    assert hu.L.moshi_hot_offset(a.m) == hu.L.moshi_hot_offset(b.m) == n + 1
    assert np.array_equal(a.read("transformer_out", cfg.dim), b.read("transformer_out", cfg.dim))
    for ia in after:                                            # everything the model does afterwards is identical, bit for bit
        ra, rb = a.lm_step(ia), b.lm_step(ia)
        assert ra == rb and a.last_raw() == b.last_raw()
        assert np.array_equal(a.read("text_logits", cfg.text_card), b.read("text_logits", cfg.text_card))
        assert np.array_equal(a.read(f"dep_logits{cfg.dep_q - 1}", cfg.card), b.read(f"dep_logits{cfg.dep_q - 1}", cfg.card))
    a.free(); b.free()


def test_batched_prefill_falls_back_to_single_frames_at_the_ring_wrap():
    cfg = hu.hot.tiny(hu.L, layers=1, context=12)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    rng = np.random.default_rng(22)
    frames = [[int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist() for _ in range(20)]   # crosses capacity 12
    a, b = hu.Model("oracle", cfg), hu.Model("oracle", cfg)
    for f in frames:
        a.lm_step_n(f)
    b.prefill(frames, 8)
    ia = list(range(cfg.n_q - cfg.dep_q))
    for _ in range(3):
        assert a.lm_step(ia) == b.lm_step(ia)
        assert np.array_equal(a.read("text_logits", cfg.text_card), b.read("text_logits", cfg.text_card))
    a.free(); b.free()


def test_personaplex_system_prompts_batched_equals_frame_by_frame():
    cfg = hu.hot.tiny_personaplex(hu.L, layers=2)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    a, b = hu.Model("oracle", cfg), hu.Model("oracle", cfg)
    a.system_prompts([21, 22, 23, 24, 25])
    b.system_prompts([21, 22, 23, 24, 25], batched=True, chunk=8)
    assert hu.L.moshi_hot_offset(a.m) == hu.L.moshi_hot_offset(b.m) == 17
    for i in range(3):
        ia = list(range(i, i + 8))
        assert a.lm_step(ia) == b.lm_step(ia) and a.last_raw() == b.last_raw()
        assert np.array_equal(a.read("text_logits", cfg.text_card), b.read("text_logits", cfg.text_card))
    a.free(); b.free()


# ---- BASELINE.json configs[0]: mimi-decode on a 1-second .mimi file (12.5 frames of codes), CPU -------------------------------------------
def mimi_decoder_cfg(n_q):
    cfg = hu.hot.moshika(hu.L)                     # the real codec: 2048-entry codebooks, SEANet + 8-layer transformer
    cfg.enable_lm = cfg.enable_mimi_encoder = 0
    cfg.mimi_n_q = n_q                             # mimi_alloc(..., n_q) sizes rvq_rest to n_q - 1 levels (lm_default.h:229-241)
    return cfg


@pytest.mark.parametrize("n_q", [1, 8, 32])
def test_mimi_file_one_second_decodes_on_the_cpu_device(tmp_path, n_q):
    rng = np.random.default_rng(n_q)
    frames = rng.integers(0, 2048, (13, n_q)).tolist()
    path = str(tmp_path / "one_second.mimi")
    hu.write_mimi(path, frames)
    with open(path, "ab") as f:
        f.write(b"\x01")                           # a ragged tail (less than one frame) is ignored, like the tool's fread loop
    got_q, codes = hu.read_mimi(path)
    assert got_q == n_q and codes.tolist() == frames
    _, pcm = hu.decode_mimi_file("oracle", path, mimi_decoder_cfg)
    assert pcm.size == 13 * 1920 and np.isfinite(pcm).all() and np.abs(pcm).max() > 0      # 13 frames x 80 ms at 24 kHz
    _, again = hu.decode_mimi_file("oracle", path, mimi_decoder_cfg)
    assert np.array_equal(pcm, again)


def test_mimi_file_rejects_bad_headers(tmp_path):
    p = str(tmp_path / "bad.mimi")
    open(p, "wb").write(b"MIMO" + b"\x08\x00\x00\x00")
    with pytest.raises(ValueError):
        hu.read_mimi(p)
    open(p, "wb").write(b"MIMI" + b"\x21\x00\x00\x00")     # n_q = 33 > 32 (tools/mimi-decode.cpp:146-149)
    with pytest.raises(ValueError):
        hu.read_mimi(p)


def test_batched_prefill_edge_cases_zero_one_and_chunk_one():
    cfg = hu.hot.tiny(hu.L, layers=1)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    rng = np.random.default_rng(5)
    frames = [[int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist() for _ in range(5)]
    ref = hu.Model("oracle", cfg)
    for f in frames:
        ref.lm_step_n(f)
    want = ref.lm_step([1] * (cfg.n_q - cfg.dep_q))
    for split in ([0, 5], [1, 4], [5, 0], [2, 1, 2]):          # empty calls, single frames (stepped one by one), mixed
        m = hu.Model("oracle", cfg)
        at = 0
        for n in split:
            m.prefill(frames[at:at + n], 1 if n == 1 else 0) if n else hu.L.moshi_hot_prefill(m.m, None, 0, 0)
            at += n
        assert hu.L.moshi_hot_offset(m.m) == 5
        assert m.lm_step([1] * (cfg.n_q - cfg.dep_q)) == want, split
        m.free()
    ref.free()
