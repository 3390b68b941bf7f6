"""-m gpu: whole-frame parity of the streaming-decode hot path, MI355X backend vs the CPU oracle, through
the C-ABI driver (include/moshi_hot.h): same synthetic weights (same seed), same inputs.

Bars (BASELINE.json north_star):
  * greedy token ids bit-exact on every free-running parity sequence below;
  * logits: typical agreement is float-summation noise (median < 1e-5 of max |logit|). ggml's activation
    quantisers (Q8_K / Q8_0 / BF16 / F16 stores) are discontinuous, so a 1e-7 input difference occasionally flips
    one rounded value and moves that step's logits by ~1e-3..1e-2 on this 512-wide test model. Hard bound asserted:
    1e-2 of max (the reference's own accepted backend tolerance, src/replay.h:333-341) on the short free-running
    sequences; teacher-forced long runs assert the distribution (median < 1e-4, 80 % of steps < 1e-2);
  * codec samples within 1e-2 of max |sample| (median far lower)."""
import ctypes as C

import numpy as np
import pytest

import hot_util as hu
from ggml_util import BF16, F32, Q4_0, Q4_K, Q8_0

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-2
PCM_TOL = 1e-2


def run_lm(kind, cfg, steps, seed=3, flags=0, forced=None, srand=None, context_fill=0):
    """Free-running when forced is None; otherwise after every step the ring is overwritten with forced[i] (teacher forcing).
    srand: re-seed libc rand() before every step (the sampler's exponential noise is drawn from it on the host)."""
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    if context_fill:
        hu.L.moshi_hot_set_context_fill(m.m, context_fill)   # Temporal ring position jumps ahead; skipped slots hold their zero init
    rng = np.random.default_rng(seed)
    rec = []
    n_in = cfg.n_q - cfg.io_dep_q
    for i in range(steps):
        if srand is not None:
            srand(1000 + i)
        ia = rng.integers(0, cfg.card, n_in).tolist()
        r, txt, aud = m.lm_step(ia)
        raw = m.last_raw()
        logits = m.read("text_logits", cfg.text_card)
        dl = m.read(f"dep_logits{cfg.dep_q - 1}", cfg.card)
        rec.append((r, txt, aud, logits, dl, raw))
        if forced is not None:
            m.force_last(*forced[i][5])
    st = m.stats() if kind == "hip" else None
    m.free()
    return rec, st


def check_lm(ref, got, tol=LOGIT_TOL):
    errs = []
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], f"step {i}: tokens differ: oracle {a[:3]} vs hip {b[:3]}"
        e = max(hu.rel_err(a[3], b[3]), hu.rel_err(a[4], b[4]))
        assert e < tol, f"step {i}: logits rel err {e:.2e}"
        errs.append(e)
    assert np.median(errs) < 1e-5, f"median logit error {np.median(errs):.2e}: more than summation noise"


# north_star bars: bf16 logits within 1e-3 (measured: bit-identical), quantised within a quantiser step. Observed maxima over 16
# steps (tests/microbench/parity_report.py): bf16 0, f32 1.2e-7, q8_0 2.0e-7 of max |logit|; q4_k / q4_0 sit at ~5e-7 except on the
# rare step where one Q8_K / Q8_0 activation value rounds the other way (see the module docstring), hence the looser hard bound.
TYPE_TOL = {BF16: 1e-6, F32: 1e-5, Q8_0: 1e-5, Q4_K: LOGIT_TOL, Q4_0: LOGIT_TOL}


@pytest.mark.parametrize("lt,et", [(Q4_K, Q4_0), (BF16, BF16), (F32, F32), (Q8_0, Q8_0), (Q4_0, Q4_0)])
def test_lm_steps_match_oracle(lt, et):
    cfg = hu.hot.tiny(hu.L, linear_type=lt, embed_type=et)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 8)
    got, st = run_lm("hip", cfg, 8)
    check_lm(ref, got, tol=TYPE_TOL[lt])
    assert st.graph_replays > 0, "cached graphs must replay as hipGraphs"
    if lt in (Q4_K, BF16, F32, Q8_0, Q4_0):
        assert st.fused_nodes_in_last_plan > 0, "fusion matchers did not fire on the Depth graph"


def test_sampled_decoding_matches_oracle_with_the_same_host_noise():
    # temp > 0: softmax(l / T) -> top-k -> argmax(p / Exp(1)) (sampling.h:4-64); the exponential noise comes from libc rand() on the
    # host and is uploaded per compute (context.h:456-480), so re-seeding libc before every step gives both backends the same draws
    import ctypes
    libc = ctypes.CDLL(None)
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.temp, cfg.temp_text, cfg.top_k, cfg.top_k_text = 0.8, 0.7, 20, 10
    ref, _ = run_lm("oracle", cfg, 10, srand=libc.srand)
    got, _ = run_lm("hip", cfg, 10, srand=libc.srand)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert a[:3] == b[:3], f"step {i}: sampled tokens differ: oracle {a[:3]} vs hip {b[:3]}"
    assert len({a[1] for a in ref}) > 1, "sampling produced a constant text token: the test is not exercising the sampler"


def test_fused_and_unfused_paths_agree_with_oracle():
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 8)
    plain, st = run_lm("hip", cfg, 8, flags=1 | 2 | 4)   # one kernel per node, no hipGraph, no upload batching
    check_lm(ref, plain)
    assert st.fused_nodes_in_last_plan == 0 and st.graph_replays == 0


def test_ring_wrap_teacher_forced():
    # context 12 < 40 steps: the Temporal ring wraps (mask branch offset > capacity, torch.h:211-214). The HIP run is
    # fed the oracle's sampled tokens after every step, so one rounding flip cannot decouple the two sequences.
    cfg = hu.hot.tiny(hu.L, context=12)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 40)
    got, _ = run_lm("hip", cfg, 40, forced=ref)
    # text logits only: the Depth steps of one frame chain through tokens sampled on the device inside the same graph,
    # which teacher forcing (applied between frames) cannot pin
    errs = np.array([hu.rel_err(a[3], b[3]) for a, b in zip(ref, got)])
    agree = np.mean([a[5][0] == b[5][0] for a, b in zip(ref, got)])
    # isolated flip steps reach a few 1e-2 on this random-weight 512-wide model; everything else is summation noise
    assert errs.max() < 0.2, f"max logit err {errs.max():.2e}"
    assert np.quantile(errs, 0.8) < 1e-2, f"80th percentile logit err {np.quantile(errs, 0.8):.2e}"
    assert np.median(errs) < 1e-4, f"median logit err {np.median(errs):.2e}"
    assert agree >= 0.9, f"greedy tokens agree on only {agree:.0%} of teacher-forced steps"


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


@pytest.mark.parametrize("streams,chain,model", [(1, 0, "tiny"), (2, 0, "tiny"), (2, 1, "tiny"), (1, 1, "tiny"), (2, 2, "tiny"), (1, 2, "tiny"), (2, 2, "personaplex")])
def test_pipelined_frame_loop_is_bit_identical_to_the_serial_loop(streams, chain, model):
    # moshi_hot_sts_pipeline_*: LM of frame k beside decode of k - 1 and encode of k + 1 (on a second command stream when codec_stream = 1).
    # Every graph consumes the same inputs and states in the same order, so tokens and PCM are the serial loop's, bit for bit.
    rng = np.random.default_rng(21)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.1 for _ in range(30 if chain == 2 else 14)]   # 30 frames cross the Temporal ring's wrap (context 24)
    make = (lambda: hu.hot.tiny(hu.L)) if model == "tiny" else (lambda: hu.hot.tiny_personaplex(hu.L))
    cfg = make()
    m = hu.Model("hip", cfg, seed=0)
    serial = [m.sts_frame(f) for f in frames]
    ring_serial = m.host_ring()
    m.free()
    cfg2 = make()
    cfg2.codec_stream = int(streams == 2)
    cfg2.chain_depth = chain      # 1: text token handed from the Temporal to the Depth graph on the device, next step's inputs staged behind the Depth graph;
                                  # 2: run-ahead - the samples reach the next Temporal graph through device memory, step k is queued before step k - 1 is read
    m = hu.Model("hip", cfg2, seed=0)
    piped = m.sts_pipeline(frames)
    ring_piped = m.host_ring()
    m.free()
    assert any(a[0] for a in serial)
    # the host-side delay ring too (PersonaPlex: the other speaker's delay-0 codes share a row with the previous step's samples; run-ahead writes them in
    # the opposite order and must still leave the serial loop's ring)
    assert np.array_equal(ring_serial, ring_piped), f"host delay ring differs at {np.argwhere(ring_serial != ring_piped)[:8].tolist()}"
    for i, (a, b) in enumerate(zip(serial, piped)):
        assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], f"frame {i}: serial {a[:3]} vs pipelined {b[:3]}"
        if a[0]:
            assert np.array_equal(a[3], b[3]), f"frame {i}: pcm differs, max {np.abs(a[3] - b[3]).max():.3e}"


@pytest.mark.parametrize("shape", ["stt", "tts"])
def test_pipelined_loop_with_half_a_codec_equals_the_serial_one(shape):
    # stt-shaped (no Depth transformer, VAD head, encoder only) and tts-shaped (text hook, conditions, decoder only) models through the same pipelined calls:
    # the codec half they have runs on the second stream beside the LM step of the neighbouring frame; tokens, VAD probabilities and PCM as in the serial order
    rng = np.random.default_rng(31)
    n = 10
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.1 for _ in range(n)]

    def make():
        if shape == "stt":
            cfg = hu.hot.tiny(hu.L, dep_q=0, n_q=8)
            cfg.extra_heads, cfg.extra_heads_dim = 3, 6
            cfg.enable_mimi_decoder = 0
            cfg.mimi_n_q = 8
        else:
            cfg = hu.hot.tiny_tts(hu.L)
            cfg.enable_mimi_encoder = 0
            cfg.mimi_n_q = cfg.dep_q
        return cfg
    res = []
    for piped in (False, True):
        cfg = make()
        cfg.codec_stream = int(piped)
        m = hu.Model("hip", cfg, seed=0)
        if shape == "tts":
            hu.set_conditions(m, cfg)
            hu.set_text_hook(m, lambda offset, sampled: int((offset * 13) % cfg.text_card))
        txt, aud, out = C.c_int32(-7), (C.c_int32 * 64)(), np.zeros(1920, np.float32)
        got = []
        if not piped:
            codes, vad = (C.c_int32 * 64)(), C.c_float(-1.0)
            for f in frames:
                out[:] = 0
                if shape == "stt":
                    hu.L.moshi_hot_mimi_encode(m.m, f.ctypes.data, codes)
                    r = hu.L.moshi_hot_lm_step_n(m.m, codes, cfg.n_q, C.byref(txt), aud, C.byref(vad))
                    got.append((r, txt.value if r else None, round(vad.value, 7) if r else None, None))
                else:
                    r = hu.L.moshi_hot_lm_step_n(m.m, codes, 0, C.byref(txt), aud, None)
                    if r:
                        hu.L.moshi_hot_mimi_decode(m.m, aud, out.ctypes.data)
                    got.append((r, (txt.value, list(aud)[:cfg.dep_q]) if r else None, None, out.copy() if r else None))
        else:
            hu.L.moshi_hot_sts_pipeline_begin(m.m, frames[0].ctypes.data if shape == "stt" else None)
            pcm_of = {}
            for k in range(n):
                prev = np.zeros(1920, np.float32)
                nxt = frames[k + 1].ctypes.data if (shape == "stt" and k + 1 < n) else None
                r = hu.L.moshi_hot_sts_pipeline_frame(m.m, nxt, C.byref(txt), aud, prev.ctypes.data)
                if r & 2:
                    pcm_of[k - 1] = prev
                if shape == "stt":
                    got.append((r & 1, txt.value if r & 1 else None, round(hu.L.moshi_hot_sts_pipeline_vad(m.m), 7) if r & 1 else None, None))
                else:
                    got.append([r & 1, (txt.value, list(aud)[:cfg.dep_q]) if r & 1 else None, None, None])
            last = np.zeros(1920, np.float32)
            if hu.L.moshi_hot_sts_pipeline_end(m.m, C.byref(txt), aud, last.ctypes.data) & 2:
                pcm_of[n - 1] = last
            if shape == "tts":
                got = [(g[0], g[1], None, pcm_of.get(k) if g[0] else None) for k, g in enumerate(got)]
        m.free()
        res.append(got)
    assert sum(g[0] for g in res[0]) >= n - 6
    for k, (a, b) in enumerate(zip(*res)):
        assert a[:3] == b[:3], f"frame {k}: serial {a[:3]} vs pipelined {b[:3]}"
        if a[3] is not None:
            assert b[3] is not None and np.array_equal(a[3], b[3]), f"frame {k}: pcm differs"


def test_temporal_stack_on_the_device_matches_the_pytorch_restatement():
    # tests/golden/temporal_stack.npz was computed by PyTorch (tests/golden/make_module_golden.py), not by the oracle: the MI355X kernels (fused and one
    # per node) against an independent implementation of the architecture, F32 weights, 9 frames across the ring wrap
    import test_oracle_golden as tg
    assert tg._run_temporal_stack_fixture("hip", 2e-6) < 1e-6
    ev = []
    assert tg._run_temporal_stack_fixture("hip", 5e-6, model="personaplex", events=ev) < 5e-6      # 144 chained Depth steps: worst 2.0e-6 ...
    # ... but for BF16 rounding-tie events: one was seen (5.6e-5 at step 5 of a frame), and the later steps of that frame attend to the row it moved (7 steps up to 1e-4)
    assert len(ev) <= 16 and all(e < 1e-3 for e in ev), ev


def test_streaming_mimi_encoder_on_the_device_matches_the_offline_pytorch_restatement():
    # the device's streaming encoder against the PyTorch fixture (computed offline over the whole signal, not by the oracle)
    import test_oracle_golden as tg
    same, worst = tg._run_mimi_encoder_fixture("hip")
    assert same >= 0.9 and worst < 1e-2, (same, worst)      # (a code may sit on a centroid tie; the CPU oracle reproduces all of them)
    same, worst = tg._run_mimi_encoder_fixture("hip", full=True)
    assert same >= 0.85 and worst < 1e-2, (same, worst)     # 8 levels of 2048 centroids: a first-level tie would move that frame's later levels too


def test_quantised_stack_on_the_device_matches_the_block_arithmetic_restatement():
    # Q4_K linears, Q8_K activation rounding: the device against the numpy / PyTorch restatement (not the oracle); rounding-tie events may fall elsewhere
    import test_oracle_golden as tg
    errs, _ = tg._run_quantised_stack_fixture("hip")
    assert errs.size >= 20 and errs.max() < 3e-2 and np.median(errs) < 2e-6 and (errs > 1e-5).sum() <= max(4, errs.size // 8), (errs.max(), np.median(errs), (errs > 1e-5).sum())


def test_streaming_mimi_decoder_on_the_device_matches_the_offline_pytorch_restatement():
    import test_oracle_golden as tg
    assert tg._run_mimi_decoder_fixture("hip") < 3e-3


def test_lm_step_on_the_device_matches_hugging_face_moshi():
    # tests/golden/hf_moshi.npz: Hugging Face transformers' Moshi decoder layers + Depth decoder (independent implementation, float32) over the same synthetic weights
    import test_oracle_golden as tg
    # (round 5: ggml's BF16 rounding sites emulated inside HF's attention - the fixture and the oracle agree to 3e-7, the device follows within its F32 kernels' noise)
    e_out, e_txt, e_dep, agree = tg._run_hf_moshi_fixture("hip")
    assert e_out < 2e-5 and e_txt < 2e-5 and e_dep < 2e-5 and agree == 1.0, (e_out, e_txt, e_dep, agree)


def test_lm_step_at_moshika_width_on_the_device_matches_hugging_face_moshi():
    # tests/golden/hf_moshi_wide.npz: the independent pin at moshika's widths (dim 4096, 32 heads, FFN 11264, 32 000 text logits, the full Depth transformer), F32 weights
    import test_oracle_golden as tg
    e_out, e_txt, e_dep, agree = tg._run_hf_moshi_fixture("hip", wide=True)
    assert e_out < 1e-3 and e_txt < 1e-3 and e_dep < 5e-3 and agree == 1.0, (e_out, e_txt, e_dep, agree)


def test_streaming_codec_on_the_device_matches_hugging_face_mimi():
    # tests/golden/hf_mimi.npz: Hugging Face transformers' Mimi (independent implementation, float32, offline) over the same synthetic weights
    import test_oracle_golden as tg
    same, lat, pcm = tg._run_hf_mimi_fixture("hip")
    assert same >= 46 / 48 and lat < 1e-2 and pcm < 5e-3, (same, lat, pcm)


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
    errs = np.array([hu.rel_err(a, b) for a, b in zip(pcm["oracle"], pcm["hip"])])
    assert errs.max() < PCM_TOL and np.median(errs) < 1e-3, f"pcm rel err max {errs.max():.2e} median {np.median(errs):.2e}"


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


@pytest.mark.parametrize("full", [False, True])
def test_codec_convolutions_without_im2col_launches_are_bit_identical(full):
    # default: the launch that produces a codec convolution's input (a conv product or a transposed conv's finishing launch) also writes that convolution's
    # F16 im2col panel - f16(elu(.)) of every output element at its panel places, the carried-tail columns by the launch's extra workgroup - so no im2col
    # launch runs between the two; flag 64 keeps one im2col launch per convolution. Same operand values at the same places, same products in the same order:
    # codes and PCM equal bit for bit over 12 streaming frames, at the tiny and at moshika's codec widths.
    cfg = hu.hot.moshika(hu.L) if full else hu.hot.tiny(hu.L)
    cfg.enable_lm = 0
    rng = np.random.default_rng(23)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.2 for _ in range(12)]
    res = {}
    for flags in (0, 64):
        m = hu.Model("hip", cfg, seed=0, flags=flags)
        codes = [m.mimi_encode(f) for f in frames]
        pcm = [m.mimi_decode(c).copy() for c in codes]
        m.free()
        res[flags] = (codes, pcm)
    assert res[0][0] == res[64][0], "codes differ"
    for i, (a, b) in enumerate(zip(res[0][1], res[64][1])):
        assert np.array_equal(a, b), f"frame {i}: PCM differs by {np.abs(a - b).max():.3e}"


def test_transposing_copies_in_front_of_the_codec_transformers_are_written_by_their_producers():
    # the encoder's last conv and the decoder's upsampler feed the codec transformers through cont(transpose(.)): the producing launch stores through the
    # copy's strides instead (one launch less per direction). MI355X_NO_CONV_TRANSPOSE_FOLD keeps the copy: same codes, same PCM, bit for bit, fewer kernels.
    import os
    cfg = hu.hot.moshika(hu.L)
    cfg.enable_lm = 0
    rng = np.random.default_rng(29)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.2 for _ in range(6)]
    res = {}
    for keep_copy in (False, True):
        if keep_copy:
            os.environ["MI355X_NO_CONV_TRANSPOSE_FOLD"] = "1"
        try:
            m = hu.Model("hip", cfg, seed=0)
            codes = [m.mimi_encode(f) for f in frames]
            pcm = [m.mimi_decode(c).copy() for c in codes]
            kernels = m.stats().kernels_in_last_plan
            m.free()
        finally:
            os.environ.pop("MI355X_NO_CONV_TRANSPOSE_FOLD", None)
        res[keep_copy] = (codes, pcm, kernels)
    assert res[False][0] == res[True][0], "codes differ"
    for i, (a, b) in enumerate(zip(res[False][1], res[True][1])):
        assert np.array_equal(a, b), f"frame {i}: PCM differs by {np.abs(a - b).max():.3e}"
    assert res[False][2] == res[True][2] - 1, (res[False][2], res[True][2])   # (the last plan is the decoder's)


def test_mimi_fused_equals_unfused():
    # the codec fusions (streaming conv / conv-transpose groups, RVQ levels and sums, scalar gather, fused transformer layers)
    # against one generic kernel per node on the same device: same codes, same samples up to summation order
    cfg = hu.hot.tiny(hu.L)
    cfg.enable_lm = 0
    rng = np.random.default_rng(21)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.2 for _ in range(6)]
    res = {}
    for flags in (0, 1 | 2 | 4):
        m = hu.Model("hip", cfg, seed=0, flags=flags)
        codes = [m.mimi_encode(f) for f in frames]
        pcm = [m.mimi_decode(c) for c in codes]
        st = m.stats()
        m.free()
        res[flags] = (codes, pcm, st)
    assert res[0][0] == res[7][0], "RVQ codes differ between the fused and the per-node path"
    errs = [hu.rel_err(a, b) for a, b in zip(res[7][1], res[0][1])]
    assert max(errs) < 1e-5, f"pcm fused vs unfused rel err {max(errs):.2e}"
    assert res[0][2].fused_nodes_in_last_plan > 0 and res[7][2].fused_nodes_in_last_plan == 0


def test_long_ring_split_attention_teacher_forced():
    # Ring capacity 1280 >= ATTN_SPLIT_MIN_C: the Temporal attention runs split over 5 workgroups per head once more than
    # 256 slots are live (hip_kernels_fused.hip, attn_decode_kernel<true>). 560 teacher-forced steps cross P = 1 -> 2 -> 3
    # participating workgroups; the oracle evaluates the reference's full-capacity soft_max.
    cfg = hu.hot.tiny(hu.L, context=1280, layers=1, dep_q=1, n_q=2)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    steps = 560
    ref, _ = run_lm("oracle", cfg, steps)
    got, _ = run_lm("hip", cfg, steps, forced=ref)
    errs = np.array([hu.rel_err(a[3], b[3]) for a, b in zip(ref, got)])
    agree = np.mean([a[5][0] == b[5][0] for a, b in zip(ref, got)])
    late = errs[300:]   # steps that ran with two or three workgroups per head
    assert errs.max() < 0.2, f"max logit err {errs.max():.2e}"
    assert np.quantile(late, 0.8) < 1e-2, f"80th percentile logit err (split steps) {np.quantile(late, 0.8):.2e}"
    assert np.median(late) < 1e-4, f"median logit err (split steps) {np.median(late):.2e}"
    assert agree >= 0.9, f"greedy tokens agree on only {agree:.0%} of teacher-forced steps"


def test_long_ring_split_attention_across_the_wrap():
    # the same split path with every slot live and the write position wrapping (offset 1270 .. 1310 over a ring of 1280): five
    # workgroups per head, the freshly written slot owned by the last and then by the first of them
    cfg = hu.hot.tiny(hu.L, context=1280, layers=1, dep_q=1, n_q=2)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 40, context_fill=1270)
    got, _ = run_lm("hip", cfg, 40, forced=ref, context_fill=1270)
    errs = np.array([hu.rel_err(a[3], b[3]) for a, b in zip(ref, got)])
    agree = np.mean([a[5][0] == b[5][0] for a, b in zip(ref, got)])
    assert errs.max() < 0.2 and np.quantile(errs, 0.8) < 1e-2 and np.median(errs) < 1e-4, f"logit err max {errs.max():.2e} p80 {np.quantile(errs, 0.8):.2e} median {np.median(errs):.2e}"
    assert agree >= 0.9, f"greedy tokens agree on only {agree:.0%} of teacher-forced steps"


def test_depth_transformer_at_moshika_width_uses_the_attention_prologue():
    # The Depth transformer at its real width (1024 = 16 heads x 64, gated FFN 2816, ring = dep_q): only this shape takes the
    # out_proj mat-vec whose prologue recomputes the short-ring attention (heads spread over 8 waves) - the 256-wide test model
    # does not. Temporal stays small so the oracle finishes quickly.
    cfg = hu.hot.tiny(hu.L, dep_q=4, n_q=8)
    cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden = 1024, 16, 2, 2816
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    steps = 12
    ref, _ = run_lm("oracle", cfg, steps)
    got, _ = run_lm("hip", cfg, steps)
    plain, _ = run_lm("hip", cfg, steps, flags=1 | 2 | 4)
    for name, run in (("fused", got), ("per-node", plain)):
        assert all(a[:3] == b[:3] for a, b in zip(ref, run)), f"{name}: greedy tokens differ from the oracle"
        errs = np.array([max(hu.rel_err(a[3], b[3]), hu.rel_err(a[4], b[4])) for a, b in zip(ref, run)])
        # summation-order noise everywhere except the occasional activation-quantiser flip step (module docstring), which shows up
        # identically in the per-node run
        assert np.median(errs) < 1e-5 and errs.max() < 0.1 and np.mean(errs < 1e-3) >= 0.6, f"{name}: logit errors {errs}"
    # the prologue-fused attention is the same arithmetic as the stand-alone kernel chain: the two device runs agree closely
    dd = np.array([hu.rel_err(a[4], b[4]) for a, b in zip(plain, got)])
    assert np.median(dd) < 1e-6 and dd.max() < 1e-2, f"fused vs per-node Depth logits: median {np.median(dd):.2e} max {dd.max():.2e}"


def near_tie(la, tok_ref, tok_got, err):
    """tok_got is an acceptable greedy pick iff, in the ORACLE's logits, it sits within the observed logit disagreement of the oracle's pick."""
    return tok_ref == tok_got or float(la[tok_ref] - la[tok_got]) <= 2.0 * err * float(np.abs(la).max()) + 1e-6


def snapshot(m, cfg):
    return (m.last_raw(), m.read("text_logits", cfg.text_card).copy(), [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)])


def compare_frame(cfg, ref, got, text_tol, dep_tol, where):
    """One frame's samples, oracle vs device: logits within the bars; a sample may differ only where the ORACLE's logits make it a tie within
    the observed disagreement, and the Depth chain is compared up to its first such divergence (later steps are conditioned on it).
    Returns (tokens compared, tokens equal)."""
    (ta, da), la, dla = ref
    (tb, db), lb, dlb = got
    e = hu.rel_err(la, lb)
    assert e < text_tol, f"{where}: text logits rel err {e:.2e}"
    assert near_tie(la, ta, tb, e), f"{where}: text token {tb} vs oracle {ta} is not a near-tie (err {e:.2e})"
    if ta != tb:
        return 1, 0
    compared = equal = 1
    for k in range(cfg.dep_q):
        e = hu.rel_err(dla[k], dlb[k])
        assert e < dep_tol, f"{where} depth {k}: logits rel err {e:.2e}"
        assert near_tie(dla[k], da[k], db[k], e), f"{where} depth {k}: token {db[k]} vs oracle {da[k]} is not a near-tie (err {e:.2e})"
        compared += 1; equal += da[k] == db[k]
        if da[k] != db[k]:
            break
    return compared, equal


def test_full_moshika_config_mimi_codec_matches_oracle():
    # the codec at the benchmark configuration (mimi_n_q 8 of 32 codebooks, 2048-entry tables): codes bit-exact, samples within PCM_TOL
    cfg = hu.hot.moshika(hu.L)
    cfg.num_layers = 1; cfg.dep_layers = 1             # the transformers are covered above; keep the oracle build quick
    rng = np.random.default_rng(77)
    frames = [rng.standard_normal(1920).astype(np.float32) * 0.1 for _ in range(3)]
    codes, pcm = {}, {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        codes[kind] = [m.mimi_encode(f) for f in frames]
        pcm[kind] = [m.mimi_decode(c) for c in codes["oracle"]]
        m.free()
    assert codes["oracle"] == codes["hip"]
    for i, (a, b) in enumerate(zip(pcm["oracle"], pcm["hip"])):
        assert hu.rel_err(a, b) < PCM_TOL, f"frame {i}: pcm rel err {hu.rel_err(a, b):.2e}"


# ---- model variants of BASELINE.json configs[4] (PersonaPlex) and configs[2] (stt: VAD head) ------------------------------------------
# Q8_0 linears: 16 chained greedy picks over 2048-way random-weight logits are decided by margins of ~1e-2, which a single Q8_K activation
# flip (Q4_K's dot type, module docstring) would overturn; Q8_0 keeps the comparison at summation noise so token ids can be asserted exactly.
def test_personaplex_lm_steps_match_oracle():
    # 17 codebooks, 16 chained Depth steps over a ring of 8 (steps 8..15 wrap inside ONE graph: bias_pattern_index's second branch,
    # torch.h:211-214), 8 of them exposed by the frame protocol (lm.h:802-805)
    cfg = hu.hot.tiny_personaplex(hu.L, linear_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 8)
    got, st = run_lm("hip", cfg, 8)
    plain, _ = run_lm("hip", cfg, 8, flags=1 | 2 | 4)
    check_lm(ref, got, tol=TYPE_TOL[F32])
    check_lm(ref, plain, tol=TYPE_TOL[F32])
    assert all(a[5] == b[5] for a, b in zip(ref, got)), "raw samples of all 16 Depth steps"
    assert st.graph_replays > 0


def test_personaplex_depth_at_real_width_wraps_the_ring_inside_the_attention_prologue():
    cfg = hu.hot.tiny_personaplex(hu.L, linear_type=F32)
    cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden = 1024, 16, 2, 2816
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    steps = 8
    ref, _ = run_lm("oracle", cfg, steps)
    got, _ = run_lm("hip", cfg, steps, forced=ref)
    plain, _ = run_lm("hip", cfg, steps, flags=1 | 2 | 4, forced=ref)
    for name, run in (("fused", got), ("per-node", plain)):
        errs = np.array([max(hu.rel_err(a[3], b[3]), hu.rel_err(a[4], b[4])) for a, b in zip(ref, run)])
        agree = np.mean([a[5] == b[5] for a, b in zip(ref, run)])
        assert np.median(errs) < 1e-4 and errs.max() < 0.2 and agree >= 0.75, f"{name}: logit errors {errs}, frames with identical samples {agree:.0%}"
    dd = np.array([hu.rel_err(a[4], b[4]) for a, b in zip(plain, got)])
    assert np.median(dd) < 1e-5, f"fused vs per-node last-step Depth logits: {dd}"


def test_personaplex_prompt_frames_and_voice_embedding_frames_match_oracle():
    # system prompts = "provided" frames through the same cached graphs (lm.h:1079-1134); voice-prompt frames feed a precomputed
    # embedding through the Temporal stack built on the scratch context every frame (lm.h:694-709, 1004-1037): an uncached device plan.
    # 22 frames x 16 chained picks: the ordinary frames are teacher-forced and a pick may differ only at an oracle near-tie (BF16 cache rows
    # round either way even with F32 weights; one such 1e-4 tie shows up in this very sequence).
    cfg = hu.hot.tiny_personaplex(hu.L, linear_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    rng = np.random.default_rng(9)
    embs = [rng.standard_normal(cfg.dim).astype(np.float32) for _ in range(3)]
    rec = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        r = []
        for e in embs:
            m.lm_step_embedding(e)
            r.append((m.last_raw(), None, None, m.read("transformer_out", cfg.dim).copy()))
        m.system_prompts([5, 6, 7, 8])
        r.append(snapshot(m, cfg) + (m.read("transformer_out", cfg.dim).copy(),))
        for i in range(3):
            out = m.lm_step(list(range(i, i + 8)))
            r.append(snapshot(m, cfg) + (m.read("transformer_out", cfg.dim).copy(), out))
            if kind == "hip":
                m.force_last(*rec["oracle"][len(r) - 1][0])
        assert hu.L.moshi_hot_offset(m.m) == 3 + 16 + 3
        rec[kind] = r
        m.free()
    compared = equal = 0
    for i, (a, b) in enumerate(zip(rec["oracle"], rec["hip"])):
        assert hu.rel_err(a[3], b[3]) < 1e-3, f"stage {i}: transformer_out rel err {hu.rel_err(a[3], b[3]):.2e}"
        if i >= 3:      # embedding frames leave no text logits behind (scratch context)
            c, e = compare_frame(cfg, a[:3], b[:3], 1e-3, 1e-2, f"stage {i}")
            compared += c; equal += e
    assert equal >= compared - 2, f"{equal} of {compared} picks equal"
    assert rec["oracle"][4][4] == rec["hip"][4][4]       # first ordinary frame after the prompts: same protocol output


def test_stt_shape_no_depth_graph_and_vad_head():
    # configs[2] (moshi-stt): dep_q = 0, every audio codebook is an input (lm.h:807-825), the VAD head runs on the scratch context (lm.h:966-976)
    cfg = hu.hot.tiny(hu.L, dep_q=0, n_q=8)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.extra_heads, cfg.extra_heads_dim = 3, 6
    rng = np.random.default_rng(2)
    inputs = [rng.integers(0, cfg.card, cfg.n_q).tolist() for _ in range(6)]
    rec = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        rec[kind] = [m.lm_step_n(ia, vad=True) + (m.read("text_logits", cfg.text_card).copy(),) for ia in inputs]
        m.free()
    for i, (a, b) in enumerate(zip(rec["oracle"], rec["hip"])):
        assert a[:3] == b[:3], f"step {i}: {a[:3]} vs {b[:3]}"
        assert abs(a[3] - b[3]) < 1e-4, f"step {i}: vad {a[3]} vs {b[3]}"
        assert hu.rel_err(a[4], b[4]) < LOGIT_TOL
    assert any(a[0] == 1 and 0.0 < a[3] < 1.0 for a in rec["oracle"])


def run_tts(kind, cfg, steps, flags=0, forced=None):
    m = hu.Model(kind, cfg, seed=0, flags=flags)
    hu.set_conditions(m, cfg)
    n = cfg.text_card + 1
    hu.set_text_hook(m, lambda offset, sampled: ((offset * 7) % 5) * n + (offset * 13) % cfg.text_card)   # demuxed (second + 1, first) pairs
    rec = []
    for i in range(steps):
        out = m.lm_step_n([])
        rec.append((out, m.last_raw(), m.read("text_logits", cfg.text_card).copy(), m.read("transformer_out", cfg.dim).copy(),
                    [m.read(f"dep_logits{k}", cfg.card).copy() for k in range(cfg.dep_q)] if i >= cfg.delay_steps else None))
        if forced is not None:
            m.force_last(*forced[i][1])
    st = m.stats() if kind == "hip" else None
    m.free()
    return rec, st


@pytest.mark.parametrize("lt", [F32, Q8_0, Q4_K], ids=["f32", "q8_0", "q4_k"])
def test_tts_branches_match_oracle(lt):
    # configs[1] (moshi-tts): cross-attention over a cached F32 K/V in every layer (q through a row view of the quantised in_proj, LayerNorm
    # eps 0), demuxed text embeddings, condition_sum, low-rank Depth embeddings (Q4_K falls back to Q4_0 at width 128), a weight schedule
    # sharing weight sets between steps, delay_steps; the text token comes from the hook (the state machine is above the boundary)
    cfg = hu.hot.tiny_tts(hu.L, linear_type=lt)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    steps = 8
    ref, _ = run_tts("oracle", cfg, steps)
    # F32 weights: summation noise only. Quantised: one Q8_0 / Q8_K activation flip (module docstring) perturbs a cache row and then stays
    # in every later frame of this 8-frame run, so the bar is the flip level — and the two device plans, which see bit-identical
    # activations, must agree with each other far more closely than that.
    text_tol, dep_tol = (1e-5, 1e-4) if lt == F32 else (5e-2, 0.2)
    runs = {}
    for name, flags in (("fused", 0), ("per-node", 1 | 2 | 4)):
        got, st = run_tts("hip", cfg, steps, flags=flags, forced=ref)
        runs[name] = got
        for i, (a, b) in enumerate(zip(ref, got)):
            assert a[0][0] == b[0][0], f"{name} step {i}: produced flag"
            assert hu.rel_err(a[3], b[3]) < text_tol, f"{name} step {i}: transformer_out rel err {hu.rel_err(a[3], b[3]):.2e}"
            if a[4] is not None:
                compare_frame(cfg, (a[1], a[2], a[4]), (b[1], b[2], b[4]), text_tol, dep_tol, f"{name} step {i}")
            else:
                assert a[1] == b[1]                              # Depth skipped: all -1
    dd = [max(hu.rel_err(a[2], b[2]), hu.rel_err(a[3], b[3])) for a, b in zip(runs["per-node"], runs["fused"])]
    assert np.median(dd) < 1e-5 and max(dd) < 1e-2, f"fused vs per-node: {dd}"


# ---- batched prompt prefill --------------------------------------------------------------------------------------
@pytest.mark.parametrize("lt", [F32, Q8_0, Q4_K], ids=["f32", "q8_0", "q4_k"])
def test_batched_prefill_on_the_device_matches_frame_by_frame_oracle(lt):
    # 20 prompt frames as [dim, T] passes of 8 on the device; reference = the oracle stepping them one by one (as lm.h:1063-1134 does).
    # Afterwards both continue with ordinary frames (teacher-forced): same tokens, logits within the type's bar.
    cfg = hu.hot.tiny_personaplex(hu.L, linear_type=lt)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    rng = np.random.default_rng(31)
    frames = [[int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist() for _ in range(20)]
    after = [rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist() for _ in range(4)]
    rec = {}
    for kind in ("oracle", "hip", "hip-frames"):
        m = hu.Model("hip" if kind != "oracle" else "oracle", cfg, seed=0)
        if kind == "hip":
            m.prefill(frames, 8)
        else:
            for f in frames:
                m.lm_step_n(f)
        r = [(None, None, None, m.read("transformer_out", cfg.dim).copy())]
        for i, ia in enumerate(after):
            m.lm_step(ia)
            r.append(snapshot(m, cfg) + (m.read("transformer_out", cfg.dim).copy(),))
            if kind != "oracle":
                m.force_last(*rec["oracle"][i + 1][0])
        rec[kind] = r
        m.free()
    text_tol, dep_tol = (1e-5, 1e-4) if lt == F32 else (5e-2, 0.2)
    for kind in ("hip", "hip-frames"):
        for i, (a, b) in enumerate(zip(rec["oracle"], rec[kind])):
            assert hu.rel_err(a[3], b[3]) < text_tol, f"{kind} stage {i}: transformer_out rel err {hu.rel_err(a[3], b[3]):.2e}"
            if i:
                compare_frame(cfg, a[:3], b[:3], text_tol, dep_tol, f"{kind} stage {i}")
    # batched and frame-by-frame on the device: same integer dot products row by row, so they agree far below the quantiser-flip level
    # batched and frame-by-frame on the device: same integer dot products row by row; the float sums over super-blocks associate
    # differently, so quantised types may still part by a flip
    dd = [hu.rel_err(a[3], b[3]) for a, b in zip(rec["hip-frames"], rec["hip"])]
    assert np.median(dd) < (1e-5 if lt == F32 else 1e-2), f"device prefill vs device frame-by-frame: {dd}"


def test_tts_depth_at_real_width_ring_of_32():
    # the tts Depth transformer: 1024 wide (16 heads x 64), 32 chained steps over a ring of 32 = len(schedule) (lm_default.h:86-90); rings
    # longer than 8 slots keep the stand-alone attention kernel (recomputing 32 slots in every out_proj workgroup measured no faster)
    cfg = hu.hot.tiny_tts(hu.L, linear_type=F32, dep_q=32)
    cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden = 1024, 16, 2, 2816
    cfg.linear_type = Q8_0
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    steps = 6
    ref, _ = run_tts("oracle", cfg, steps)
    runs = {}
    for name, flags in (("fused", 0), ("per-node", 1 | 2 | 4)):
        got, st = run_tts("hip", cfg, steps, flags=flags, forced=ref)
        runs[name] = got
        for i, (a, b) in enumerate(zip(ref, got)):
            if a[4] is not None:
                compare_frame(cfg, (a[1], a[2], a[4]), (b[1], b[2], b[4]), 5e-2, 0.2, f"{name} step {i}")
    # the two device plans against each other, with the same rule: equal picks, or near-ties, along the 32-step chain
    for i, (a, b) in enumerate(zip(runs["per-node"], runs["fused"])):
        if a[4] is not None:
            compare_frame(cfg, (a[1], a[2], a[4]), (b[1], b[2], b[4]), 5e-2, 0.2, f"fused vs per-node step {i}")


@pytest.mark.parametrize("n_q", [8, 32])
def test_mimi_file_decode_matches_oracle(tmp_path, n_q):
    # BASELINE.json configs[0] on the device: a 1-second .mimi file (13 frames of n_q codes) through the real codec, against the CPU device
    import test_driver_cpu as tdc
    rng = np.random.default_rng(n_q)
    path = str(tmp_path / "one_second.mimi")
    hu.write_mimi(path, rng.integers(0, 2048, (13, n_q)).tolist())
    _, ref = hu.decode_mimi_file("oracle", path, tdc.mimi_decoder_cfg)
    _, got = hu.decode_mimi_file("hip", path, tdc.mimi_decoder_cfg)
    for i in range(13):
        a, b = ref[i * 1920:(i + 1) * 1920], got[i * 1920:(i + 1) * 1920]
        assert hu.rel_err(a, b) < PCM_TOL, f"frame {i}: pcm rel err {hu.rel_err(a, b):.2e}"


def test_long_context_decode_is_deterministic():
    # 1300 sts frames from 1000 filled ring slots (the split-attention regime, across the 1024-slot range doubling) at the full moshika
    # configuration, twice from fresh models: identical tokens and PCM. The cross-workgroup hand-offs (split attention, fused argmax, VQ merge)
    # once lost a store-vs-counter race about every 1e5 hand-offs, which no oracle comparison of a few frames can see - this can.
    import zlib
    cfg = hu.hot.moshika(hu.L)
    runs = []
    for rep in range(2):
        m = hu.Model("hip", cfg, seed=0)
        hu.L.moshi_hot_set_context_fill(m.m, 1000)
        rng = np.random.default_rng(3)
        toks, crc = [], 0
        for i in range(1300):
            r, txt, aud, pcm = m.sts_frame((rng.standard_normal(1920) * 0.05).astype(np.float32))
            toks.append((r, txt, tuple(aud)))
            crc = zlib.crc32(pcm.tobytes(), crc)
        runs.append((toks, crc))
        m.free()
    first = next((i for i, (a, b) in enumerate(zip(runs[0][0], runs[1][0])) if a != b), None)
    assert runs[0] == runs[1], f"two identical runs differ, first at frame {first}"


def test_depth_shard_step_graphs_match_the_chained_graph_on_the_device():
    # SURVEY.md 8e plumbing on the MI355X backend, one rank (no transport): per-step Depth graphs + K/V-row messages vs the chained graph, and vs
    # the oracle's chained graph; the two-rank transport itself is covered on CPU (tests/test_depth_shard_cpu.py, gloo)
    from moshi_cpp_amd import shard
    cfg = hu.hot.tiny(hu.L, linear_type=F32, embed_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    ref, _ = run_lm("oracle", cfg, 8)
    chained, _ = run_lm("hip", cfg, 8)
    cfg.dep_shard_world, cfg.dep_shard_rank = 1, 0
    m = hu.Model("hip", cfg, seed=0)
    sh = shard.DepthShard(hu.L, m.m, cfg, 0, 1, None)
    sh.install()
    rng = np.random.default_rng(3)
    for i in range(8):
        r, txt, aud = m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
        assert (r, txt, aud) == ref[i][:3] == chained[i][:3], f"step {i}"
        assert m.last_raw() == ref[i][5]
    m.free()


def test_sharded_frame_behind_the_c_abi_with_rccl_on_the_backend_stream():
    # the hop loop lives behind the C-ABI and the broadcasts are ncclBroadcast calls the harness makes itself (librccl.so opened at run
    # time) on the backend's own HIP stream. One GPU = a world of one rank: every hop is still a real RCCL call, stream-ordered between the step graph
    # that packs the message and the next graph - the tokens must be the chained graph's, 1 + dep_q broadcasts per frame, no Python between the hops.
    import torch
    from moshi_cpp_amd import shard
    cfg = hu.hot.tiny(hu.L, linear_type=F32, embed_type=F32)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    chained, _ = run_lm("hip", cfg, 8)
    cfg.dep_shard_world, cfg.dep_shard_rank = 1, 0
    m = hu.Model("hip", cfg, seed=0)
    sh = shard.DepthShard(hu.L, m.m, cfg, 0, 1, None, device=torch.device("cuda", 0))
    assert sh.transport.startswith("rccl")
    sh.install()
    rng = np.random.default_rng(3)
    for i in range(8):
        r, txt, aud = m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
        assert (r, txt, aud) == chained[i][:3], f"step {i}"
    assert sh.hops == 8 * (1 + cfg.dep_q), sh.hops
    for k in range(50):
        sh.hop(0)                                     # back-to-back broadcasts on the stream (the latency probe of bench.py --shard depth)
    hu.L.ggml_backend_synchronize(m.be)
    m.free()


def test_tensor_parallel_segments_on_the_device_match_the_oracle():
    # SURVEY.md 8f.2 plumbing on the MI355X backend, one rank (the two-rank sum is covered on CPU, tests/test_temporal_tp_cpu.py): the 2 L + 1 segment
    # graphs over sliced weights vs the oracle running the same segments, several stream positions
    from moshi_cpp_amd import shard
    cfg = hu.hot.tiny(hu.L, linear_type=F32, embed_type=F32)
    cfg.ffn_hidden = 1024
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.tp_world, cfg.tp_rank = 1, 0
    outs = {}
    for kind in ("oracle", "hip"):
        m = hu.Model(kind, cfg, seed=0)
        tp = shard.TemporalTP(hu.L, m.m, cfg, 0, 1, None)
        rng = np.random.default_rng(4)
        outs[kind] = [tp.stack((rng.standard_normal(cfg.dim) * 3).astype(np.float32)) for _ in range(6)]
        m.free()
    for i, (a, b) in enumerate(zip(outs["oracle"], outs["hip"])):
        assert hu.rel_err(a, b) < 1e-5, f"position {i}: {hu.rel_err(a, b):.2e}"


def test_tensor_parallel_stack_with_rccl_all_reduce_on_the_backend_stream():
    # TemporalTP given a device sets RCCL up itself (unique id -> moshi_hot_depth_shard_rccl_init, which makes the backend's device current first) and
    # the 2 L all-reduces of moshi_hot_tp_stack are ncclAllReduce calls on the backend's stream. One GPU = a world of one rank: every reduction is still a real
    # RCCL call between two segment graphs; the result must be the transport-less stack's, bit for bit.
    import torch
    from moshi_cpp_amd import shard
    cfg = hu.hot.tiny(hu.L, linear_type=Q4_K, embed_type=Q4_0)
    cfg.ffn_hidden = 1024
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.tp_world, cfg.tp_rank = 1, 0
    outs = {}
    for how in ("plain", "rccl"):
        m = hu.Model("hip", cfg, seed=0)
        tp = shard.TemporalTP(hu.L, m.m, cfg, 0, 1, None, device=torch.device("cuda", 0) if how == "rccl" else None)
        assert tp.transport.startswith("rccl") == (how == "rccl")
        rng = np.random.default_rng(4)
        outs[how] = [tp.stack((rng.standard_normal(cfg.dim) * 3).astype(np.float32)) for _ in range(6)]
        if how == "rccl":
            assert tp.reductions == 6 * 2 * cfg.num_layers, tp.reductions
            # (a second init on the same model replaces the communicator instead of leaking it)
            shard._rccl_bootstrap(hu.L, m.m, 0, 1, None, None, torch.device("cuda", 0))
            again = tp.stack(outs["plain"][0] * 0 + 1)
            assert np.isfinite(again).all()
        m.free()
    for i, (a, b) in enumerate(zip(outs["plain"], outs["rccl"])):
        assert np.array_equal(a, b), f"position {i}"


@pytest.mark.parametrize("dim,heads,ffn", [(2048, 16, 5632), (4096, 32, 11264)])
def test_two_tensor_parallel_ranks_run_their_q4k_slices_on_the_device(dim, heads, ffn):
    # SURVEY.md 8f.2 with tp_world = 2 ON THE DEVICE: rank 0 and rank 1 each hold half of every Temporal matrix (their heads' in_proj rows, the matching
    # 256-aligned column block of out_proj, half of both linear_in halves and the matching column block of linear_out, their heads' KV ring), cut from
    # the unsplit synthetic Q4_K matrices. With one GPU in the box the two ranks' models run one after the other on it and the all-reduce is a host-side
    # sum of their partial vectors; what is checked is what the ranks compute: the 2 L + 1 segment graphs over SLICED weights through the fused HIP
    # kernels, against the unsplit stack on the oracle, at the tts / stt width and at moshika's (bars of tests/test_temporal_tp_cpu.py).
    from moshi_cpp_amd import shard
    base = hu.hot.tiny(hu.L, linear_type=Q4_K, embed_type=Q4_0)
    base.dim, base.num_heads, base.ffn_hidden, base.num_layers, base.context = dim, heads, ffn, 2, 48
    base.enable_mimi_encoder = base.enable_mimi_decoder = 0

    def cfg_for(world, rank):
        c = hu.hot.Config.from_buffer_copy(base)
        c.tp_world, c.tp_rank = world, rank
        return c
    ref_m = hu.Model("oracle", cfg_for(1, 0), seed=0)
    ref = shard.TemporalTP(hu.L, ref_m.m, ref_m.cfg, 0, 1, None)
    ranks = [hu.Model("hip", cfg_for(2, r), seed=0) for r in range(2)]
    full = hu.L.moshi_hot_weight_bytes(ref_m.m, 0)
    assert all(hu.L.moshi_hot_weight_bytes(m.m, 0) < 0.62 * full for m in ranks), "a rank holds half of every Temporal matrix"
    rng = np.random.default_rng(8)
    last = 2 * base.num_layers
    part = [np.zeros(dim, np.float32) for _ in ranks]
    errs = []
    for pos in range(6):
        x = (rng.standard_normal(dim) * 3).astype(np.float32)
        want = ref.stack(x)
        for m in ranks:
            hu.L.moshi_hot_tp_begin(m.m, x.ctypes.data)
        for i in range(last + 1):
            for m in ranks:
                hu.L.moshi_hot_tp_segment(m.m, i)
            if i < last:                                  # all_reduce(sum) of the two partial vectors
                for m, p in zip(ranks, part):
                    hu.L.moshi_hot_tp_msg_read(m.m, p.ctypes.data)
                tot = (part[0] + part[1]).astype(np.float32)
                for m in ranks:
                    hu.L.moshi_hot_tp_msg_write(m.m, tot.ctypes.data)
        outs = []
        for m in ranks:
            o = np.zeros(dim, np.float32)
            hu.L.moshi_hot_tp_end(m.m, o.ctypes.data)
            outs.append(o)
        assert np.array_equal(outs[0], outs[1]), "both ranks end a stack pass with the same vector"
        errs.append(hu.rel_err(want, outs[0]))
    st = ranks[0].stats()
    assert st.graph_replays > 0, "the segment graphs are cached graphs: from the second stream position on they replay as hipGraphs"
    for m in ranks + [ref_m]:
        m.free()
    # positions without a rounding flip agree to summation noise (measured 2-3e-7); a position where one Q8_K / BF16 value rounds the other way - the summed
    # partials differ from the unsplit sums by ~1e-7, as between any two correct implementations - moves by a quantiser step (measured 2.7e-3 / 1.1e-2 at 4096)
    assert np.median(errs) < 1e-5 and max(errs) < 3e-2 and sum(e < 1e-6 for e in errs) >= 3, errs


def test_split_attention_with_eight_wave_workgroups_passes_the_long_ring_tests():
    # MI355X_ATTN_SPLIT_NW=8 (192 / 384-slot ranges, one workgroup per CU at 3 000 slots; profiles/r03_ab_attn_split_width.txt) is a shipped option: the
    # long-ring parity tests of this file and the pre-filled-ring probe of test_full_width_parity.py run against it in a child process (the choice is read once per process)
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MI355X_ATTN_SPLIT_NW="8")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_hip_frame.py"), os.path.join(here, "test_full_width_parity.py"),
                        "-k", "long_ring_split_attention or long_context_decode or prefilled_ring_node_by_node"],
                       capture_output=True, text=True, env=env, timeout=1500, cwd=os.path.dirname(here))
    assert r.returncode == 0 and " passed" in r.stdout and "failed" not in r.stdout, (r.stdout + r.stderr)[-3000:]
