"""-m gpu: every ggml op the moshi.cpp graphs emit, MI355X backend vs the CPU oracle on
the same seeded inputs, through the C-ABI. F32 tolerance: |a-b| <= 1e-5 * max|ref| (the reference's own
replay tolerance for ggml_graph_compute_with_ctx, src/replay.h:333-341); I32 exact."""
import numpy as np
import pytest

import ggml_util as gu
from ggml_util import BF16, F16, F32, I32, Q4_0, Q4_K, Q8_0

pytestmark = pytest.mark.gpu

rng = np.random.default_rng(7)


def rnd(*shape):
    return rng.standard_normal(shape).astype(np.float32)


@pytest.mark.parametrize("op", ["add", "sub", "mul", "div"])
def test_binary_broadcast(op):
    a, b = rnd(3, 5, 33), rnd(1, 5, 1) + 3.0

    def build(g):
        return [getattr(g, op)(g.input(a), g.input(b))]
    gu.compare(build, atol_rel=0, rtol=0)


def test_binary_strided_views():
    a, w = rnd(1, 512, 1), rnd(512, 1, 4)

    def build(g):
        x = g.input(a)            # [1, 512]
        wt = g.input(w)           # [4, 1, 512]
        outs = []
        for i in range(4):        # the depthwise upsample trick of conv.h:262-278
            sub = g.view_3d(wt, 1, 512, 1, wt.contents.nb[2], wt.contents.nb[2], wt.contents.nb[0] * i)
            outs.append(g.mul(x, sub))
        y = outs[0]
        for o in outs[1:]:
            y = g.concat(y, o, 0)
        return [y]
    gu.compare(build, atol_rel=0, rtol=0)


@pytest.mark.parametrize("uop", ["neg", "silu", "gelu", "elu"])
def test_unary(uop):
    a = rnd(7, 129) * 3

    def build(g):
        return [getattr(g, uop)(g.input(a))]
    gu.compare(build, atol_rel=2e-3 if uop == "gelu" else 1e-6)   # gelu: one f16 ulp of the table semantics


def test_scale_clamp_inplace():
    a = rnd(4, 100)

    def build(g):
        x = g.input(a)
        y = g.scale(x, 0.37)
        c = g.clamp(y, -0.2, 0.3)     # in place on y
        z = g.add(c, y)               # y already clamped (order semantics)
        return [z]
    gu.compare(build, atol_rel=0, rtol=0)


def test_sum_sumrows_argmax():
    a = rnd(3, 4, 1000)
    m = rnd(5, 2048)
    m[2, 7] = m[2, 900] = 50.0      # tie: ggml_vec_argmax_f32 keeps overwriting while max == x[i], so the LAST index wins
    m[4, :] = -np.inf                # all -inf: ne0 - 1

    def build(g):
        x = g.input(a)
        return [g.sum(x), g.sum_rows(x), g.argmax(g.input(m))]
    gu.compare(build, atol_rel=1e-6)


def test_argsort_topk():
    a = rnd(3, 300)
    a[1, 10] = a[1, 200]            # tie: lower index first

    def build(g):
        x = g.input(a)
        return [g.cont(g.argsort_top_k(x, 25)), g.argsort(x, 0)]
    gu.compare(build)


def test_norm_rmsnorm():
    a = rnd(2, 4096) * 2 + 0.5

    def build(g):
        x = g.input(a)
        return [g.rms_norm(x, 1e-8), g.norm(x, 1e-5)]
    gu.compare(build, atol_rel=1e-6)


def test_soft_max_ext_mask():
    a = rnd(4, 2, 300)
    mask = np.zeros((2, 300), np.float32)
    mask[0, 100:] = -np.inf
    mask[1, 250:] = -np.inf

    def build(g):
        return [g.soft_max_ext(g.input(a), g.input(mask), 0.125, 0.0), g.soft_max(g.input(a))]
    gu.compare(build, atol_rel=1e-6)


def test_cpy_cast_cont_permute():
    a = rnd(2, 3, 4, 8)
    idx = rng.integers(0, 100, (6,)).astype(np.int32)

    def build(g):
        x = g.input(a)
        p = g.cont(g.permute(x, 0, 2, 1, 3))
        t = g.cont(g.transpose(x))
        h = g.cast(g.cast(x, F16), F32)
        b = g.cast(g.cast(x, BF16), F32)
        i = g.cast(g.cast(g.input(idx, I32), F32), I32)
        dst = g.input(np.zeros((2, 3, 4, 8), np.float32))
        c = g.cpy(g.scale(x, 2.0), dst)
        return [p, t, h, b, i, c]
    gu.compare(build, atol_rel=0, rtol=0)


def test_concat_repeat_arange_views():
    a, b, rr = rnd(2, 3, 5), rnd(2, 4, 5), rnd(1, 7)

    def build(g):
        x, y = g.input(a), g.input(b)
        c1 = g.concat(x, y, 1)
        c0 = g.concat(x, x, 0)
        r = g.repeat_4d(g.input(rr), 7, 5, 2, 1)
        ar = g.arange(0.0, 10.0, 1.0)
        v = g.cont(g.view_2d(c1, 3, 7, c1.contents.nb[1], 4))
        return [c1, c0, r, ar, v]
    gu.compare(build, atol_rel=0, rtol=0)


@pytest.mark.parametrize("ttype", [F32, F16, BF16, Q4_0, Q8_0, Q4_K])
def test_get_rows(ttype):
    k, rows = 512, 40
    idx = np.array([3, 0, 39, 17], np.int32)

    def build(g):
        if ttype == Q4_0:
            tab = g.input_raw(gu.random_q4_0(np.random.default_rng(1), rows, k), ttype, k, rows)
        elif ttype == Q8_0:
            tab = g.input_raw(gu.random_q8_0(np.random.default_rng(1), rows, k), ttype, k, rows)
        elif ttype == Q4_K:
            tab = g.input_raw(gu.random_q4_K(np.random.default_rng(1), rows, k), ttype, k, rows)
        else:
            tab = g.input(np.random.default_rng(1).standard_normal((rows, k)).astype(np.float32), ttype)
        return [g.get_rows(tab, g.input(idx, I32))]
    gu.compare(build, atol_rel=0, rtol=0)


def test_set_rows_bf16_cache():
    cache0 = rnd(4, 10, 64)
    rows = rnd(4, 2, 64)
    idx = np.array([9, 3], np.int32)

    def build(g):
        cache = g.input(cache0, BF16)
        k = g.set_rows(cache, g.input(rows), g.input(idx, I32))
        full = g.cast(k, F32)
        return [full]
    gu.compare(build, atol_rel=0, rtol=0)


@pytest.mark.parametrize("wtype,K,M,N", [(F32, 512, 96, 2), (F16, 256, 64, 3), (BF16, 128, 50, 1), (Q8_0, 256, 33, 2),
                                         (Q4_0, 512, 17, 1), (Q4_K, 1024, 40, 2), (Q4_K, 4096, 64, 1), (Q4_K, 2816, 48, 1),
                                         (F32, 512, 1536, 1), (BF16, 4096, 256, 1), (F16, 1024, 100, 1),
                                         # block mat-vec fast path (K % 256 == 0, one activation column) for the other two block formats
                                         (Q8_0, 256, 33, 1), (Q8_0, 1024, 300, 1), (Q8_0, 4096, 520, 1), (Q8_0, 2816, 64, 1),
                                         (Q4_0, 512, 17, 1), (Q4_0, 1024, 300, 1), (Q4_0, 4096, 520, 1)])
def test_mul_mat_types(wtype, K, M, N):
    r = np.random.default_rng(K + M)
    x = r.standard_normal((N, K)).astype(np.float32)
    if wtype == Q4_K:
        wraw = gu.random_q4_K(r, M, K)
    elif wtype == Q8_0:
        wraw = gu.random_q8_0(r, M, K)
    elif wtype == Q4_0:
        wraw = gu.random_q4_0(r, M, K)
    else:
        wraw = (r.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)

    def build(g):
        w = g.input_raw(wraw, wtype, K, M) if wtype in (Q4_K, Q8_0, Q4_0) else g.input(wraw, wtype)
        return [g.mul_mat(w, g.input(x))]
    gu.compare(build, atol_rel=2e-6)


def test_mul_mat_batched_heads_bf16():
    # K.q and V^T.p shapes of torch.h:232-235 with head broadcast
    D, Cc, H, T = 64, 50, 8, 2
    kc, q = rnd(H, Cc, D), rnd(H, T, D)

    def build(g):
        k = g.input(kc, BF16)
        s = g.mul_mat(k, g.input(q))                        # [C, T, H]
        vt = g.cont(g.transpose(k))                          # [C, D, H]
        o = g.mul_mat(vt, g.soft_max(s))                     # [D, T, H]
        return [s, o]
    gu.compare(build, atol_rel=2e-6)


@pytest.mark.parametrize("Cin,Cout,K,s,L", [(1, 64, 7, 1, 40), (64, 128, 8, 4, 64), (512, 1024, 7, 1, 8), (256, 512, 1, 1, 3), (64, 32, 3, 1, 50), (128, 64, 3, 1, 482), (32, 64, 1, 1, 1920),
                                           # few output positions (mul_mat_smallm_kernel: activation rows staged in LDS): M = 8 at K = 8192 fills its 128 KB, M = 2 / 1 at the codec's T = 2 layers
                                           (512, 1024, 16, 8, 72), (1024, 512, 3, 1, 4), (512, 512, 4, 2, 4), (256, 48, 1, 1, 5)])
def test_conv_1d(Cin, Cout, K, s, L):
    r = np.random.default_rng(Cin + K)
    w = (r.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    x = r.standard_normal((Cin, L)).astype(np.float32)
    bias = r.standard_normal((Cout, 1)).astype(np.float32)

    def build(g):
        y = g.conv_1d(g.input(w, F16), g.input(x), s, 0, 1)
        return [g.add(y, g.input(bias))]
    gu.compare(build, atol_rel=2e-6)


@pytest.mark.parametrize("Cin,Cout,K,s,L,wt", [(1024, 512, 16, 8, 2, F32), (128, 64, 8, 4, 30, F32), (64, 32, 4, 2, 5, F16)])
def test_conv_transpose_1d(Cin, Cout, K, s, L, wt):
    r = np.random.default_rng(Cin + K)
    w = (r.standard_normal((Cin, Cout, K)) / np.sqrt(Cin)).astype(np.float32)
    x = r.standard_normal((Cin, L)).astype(np.float32)

    def build(g):
        return [g.conv_transpose_1d(g.input(w, wt), g.input(x), s, 0, 1)]
    gu.compare(build, atol_rel=1e-6)


def test_timestep_embedding():
    def build(g):
        ts = g.input(np.array([0.0, 1.0, 57.0, 2999.0], np.float32))
        return [g.timestep_embedding(ts, 128, 10000)]
    gu.compare(build, atol_rel=2e-6)


def test_add_inplace_and_negative_view_offsets():
    # conv.h:282-297: lower = add_inplace(view(y), partial); y = view(lower, full, offset 0); cpy into state
    y0, prev0 = rnd(4, 24), rnd(4, 24)

    def build(g):
        y = g.scale(g.input(y0), 1.0)
        prev = g.input(prev0)
        PT = 8
        partial = g.view_3d(prev, PT, 4, 1, prev.contents.nb[1], prev.contents.nb[2], prev.contents.nb[0] * (24 - PT))
        lower = g.view_3d(y, PT, 4, 1, y.contents.nb[1], y.contents.nb[2], 0)
        lower = g.add_inplace(lower, partial)
        full = g.view_3d(lower, 24, 4, 1, y.contents.nb[1], y.contents.nb[2], 0)
        st = g.cpy(full, prev)
        out = g.cont(g.view_3d(st, 24 - PT, 4, 1, st.contents.nb[1], st.contents.nb[2], 0))
        # lm.h:509-527: chained 1-element views then a negative offset back to the start
        toks = g.input(np.zeros(4, np.int32), I32)
        v = g.view_1d(toks, 1, 0)
        c0 = g.cpy(g.input(np.array([11], np.int32), I32), v)
        v = g.view_1d(v, 1, 4)
        c1 = g.cpy(g.input(np.array([22], np.int32), I32), v)
        allv = g.view_1d(v, 4, (1 << 64) - 4)
        return [out, g.cont(st)], [], [c0, c1, allv]

    def build2(g):
        outs, _, extra = build(g)
        for e in extra:
            pass
        return outs + [extra[2]] if False else (outs, [])
    # run with explicit expansion order: c0, c1 first, then the view read
    def build3(g):
        outs, _, extra = build(g)
        return [extra[0], extra[1], g.cont(extra[2])] + outs
    gu.compare(build3, atol_rel=0, rtol=0)


def test_sampler_chain_top_k_with_given_noise():
    # moshi_sample_token for temp > 0 (sampling.h:4-64) with the exponential noise supplied as an input
    n, k = 500, 10
    logits = rnd(1, n) * 3
    noise = rng.exponential(1.0, (1, k)).astype(np.float32)

    def build(g):
        lg = g.input(logits)
        probs = g.soft_max(g.scale(lg, 1.0 / 0.7))
        indices = g.argsort_top_k(probs, k)
        rows = g.get_rows(g.cont(g.permute(probs, 1, 0, 2, 3)), indices)
        p2 = g.permute(rows, 1, 0, 2, 3)
        in2 = g.reshape_2d(p2, p2.contents.ne[0], p2.contents.ne[1] * p2.contents.ne[2] * p2.contents.ne[3])
        q = g.div(in2, g.input(noise))
        nxt = g.argmax(q)
        nxt4 = g.reshape_4d(nxt, nxt.contents.ne[0], p2.contents.ne[1], p2.contents.ne[2], p2.contents.ne[3])
        tok = g.get_rows(g.cont(g.permute(indices, 1, 0, 2, 3)), nxt4)
        return [tok, g.cont(indices), g.cont(q)]
    gu.compare(build)


@pytest.mark.parametrize("n,k,temp", [(2048, 250, 0.8), (32000, 25, 0.7)])
def test_fused_sampler_against_the_numpy_restatement(n, k, temp):
    # the fused sampling kernel (soft-max, radix select of the k-th largest, rank sort, p / noise, last-maximum arg-max) against the numpy restatement of
    # moshi_sample_token directly - the bench's two shapes (audio: 2048 logits, top-k 250, temperature 0.8; text: 32000, 25, 0.7), ties inside the top-k included
    import test_oracle_golden as tg
    r = np.random.default_rng(n + k + 1)
    for trial in range(4):
        logits = (r.standard_normal((1, n)) * 3).astype(np.float32)
        if trial == 3:
            logits[0, 11] = logits[0, 900] = logits[0, 5] = logits.max() + 0.5
        noise = r.exponential(1.0, (1, k)).astype(np.float32)
        (tok, idx), st = gu.run_graph("hip", lambda g: tg._sample_token_graph(g, logits, noise, temp, k))
        want_tok, want_idx = tg._sample_token_numpy(logits[0], noise[0], temp, k)
        assert int(np.asarray(tok).reshape(-1)[0]) == want_tok, trial


# ---- the large-matrix launch shape of the block mat-vec (one workgroup of 8 waves per CU, >= 1536 tiles) and the prologue /
# ---- epilogue variants that only occur at moshika's real widths
@pytest.mark.parametrize("wtype", [Q4_K, Q8_0, Q4_0])
def test_big_matvec_rmsnorm_prologue_residual_epilogue(wtype):
    K, M = 4096, 6144            # 1536 tiles of 64 super-blocks
    r = np.random.default_rng(K + M + wtype)
    x = r.standard_normal((1, K)).astype(np.float32)
    alpha = (1.0 + 0.1 * r.standard_normal((1, K))).astype(np.float32)
    res = r.standard_normal((1, M)).astype(np.float32)
    wraw = {Q4_K: gu.random_q4_K, Q8_0: gu.random_q8_0, Q4_0: gu.random_q4_0}[wtype](r, M, K)

    def build(g):
        w = g.input_raw(wraw, wtype, K, M)
        xn = g.mul(g.input(alpha), g.rms_norm(g.input(x), 1e-8))
        return [g.add(g.input(res), g.mul_mat(w, xn))]
    gu.compare(build, atol_rel=2e-6)


@pytest.mark.parametrize("wtype", [Q4_K, Q8_0, Q4_0])
def test_big_matvec_gated_ffn_prequantised_activation(wtype):
    # linear_out of the Temporal FFN: silu(h[:K]) * h[K:] with K = 11264 > 4096 is quantised once by its own kernel, the
    # mat-vec (2816 tiles -> large shape) copies the Q8_K blocks; + residual
    K, M = 11264, 4096
    r = np.random.default_rng(11264)
    h = r.standard_normal((1, 2 * K)).astype(np.float32)
    res = r.standard_normal((1, M)).astype(np.float32)
    wraw = {Q4_K: gu.random_q4_K, Q8_0: gu.random_q8_0, Q4_0: gu.random_q4_0}[wtype](r, M, K)

    def build(g):
        w = g.input_raw(wraw, wtype, K, M)
        hh = g.input(h)
        t = hh.contents
        left = g.view_4d(hh, t.ne[0] // 2, 1, t.ne[1], t.ne[2], t.nb[1] // 2, t.nb[1], t.nb[2], 0)      # gating.h:16-29
        right = g.view_4d(hh, t.ne[0] // 2, 1, t.ne[1], t.ne[2], t.nb[1] // 2, t.nb[1], t.nb[2], t.nb[1] // 2)
        gate = g.mul(g.silu(left), right)
        return [g.add(g.input(res), g.mul_mat(w, gate))]
    gu.compare(build, atol_rel=2e-6)


def test_big_matvec_argmax_epilogue_text_head():
    # text_linear: 4096 -> 32000 rows (large shape, 125 rows per workgroup) followed by the greedy argmax (fused through the
    # arrival ticket: the last workgroup scans all 32000 logits)
    K, M = 4096, 32000
    r = np.random.default_rng(32000)
    x = r.standard_normal((1, K)).astype(np.float32)
    wraw = gu.random_q4_K(r, M, K)

    def build(g):
        w = g.input_raw(wraw, Q4_K, K, M)
        logits = g.mul_mat(w, g.input(x))
        return [g.argmax(logits)], [logits]
    gu.compare(build, atol_rel=2e-6)


def test_small_uploads_to_the_same_bytes_keep_their_order():
    # ggml_backend_tensor_set of small tensors is batched and scattered by one kernel before the next compute; a zero fill followed by
    # the first value of the same state tensor (StateContext init then a condition upload) must not race inside one batch
    g = gu.Graph("hip")
    try:
        ts = [g.new(F32, 512) for _ in range(24)]
        g.build([g.add(ts[0], ts[1])])
        g.alloc()
        want = []
        for rep in range(20):
            for i, t in enumerate(ts):
                g.set(t, np.zeros(512, np.float32))
            for i, t in enumerate(ts):
                g.set(t, np.full(512, rep * 100.0 + i, np.float32))
                g.set(t, np.full(512, rep * 100.0 + i + 0.5, np.float32))
            got = [g.get(t)[0, 0, 0, 0] for t in (ts[0], ts[7], ts[23])]
            assert got == [rep * 100.0 + 0.5, rep * 100.0 + 7.5, rep * 100.0 + 23.5], (rep, got)
    finally:
        g.free()


@pytest.mark.parametrize("K,M,T", [(256, 16, 2), (512, 40, 5), (1024, 250, 16), (4096, 512, 17), (2048, 96, 32), (512, 64, 33), (4096, 128, 64), (11264, 64, 24),
                                   (2560, 72, 9),                                   # ragged last tile (10 super-blocks)
                                   (512, 8232, 20), (2560, 8200, 32), (1024, 12288, 7), (256, 9000, 40)])   # >= 8192 rows: shared activation tile variant
def test_batched_q4k_matmul_int8_mfma(K, M, T):
    # prompt prefill: T activation rows against Q4_K weights. Rows are quantised to Q8_K one by one (as ggml does for any T), the
    # sub-block dot products run on v_mfma_i32_16x16x32_i8; ragged M (not a multiple of 16) and T (not a multiple of 16) included
    r = np.random.default_rng(K + M + T)
    x = (r.standard_normal((T, K)) * r.uniform(0.2, 3.0, (T, 1))).astype(np.float32)
    wraw = gu.random_q4_K(r, M, K)

    def build(g):
        return [g.mul_mat(g.input_raw(wraw, Q4_K, K, M), g.input(x))]
    gu.compare(build, atol_rel=2e-6)

    def build3(g):     # the gated FFN hands its activation over as [K, 1, T] (gating.h:16-37)
        return [g.mul_mat(g.input_raw(wraw, Q4_K, K, M), g.input(x.reshape(T, 1, K)))]
    gu.compare(build3, atol_rel=2e-6)


def test_codebook_centroid_graph_on_the_device():
    # the load-time graph that turns a safetensors checkpoint's (embedding_sum, cluster_usage) into the RVQ centroids (core_vq.h:58-85):
    # clamp(usage, 1e-5, inf) -> cont(transpose) -> div(embedding_sum, .) -> copy into the separately allocated embedding tensor. A loader whose scratch
    # context sits on the MI355X backend gets the same bytes as on the host device.
    r = np.random.default_rng(12)
    card, dim = 2048, 256
    usage = np.abs(r.standard_normal(card)).astype(np.float32) * 50
    usage[r.integers(0, card, 40)] = 0.0                      # dead codes: the clamp's reason
    esum = (r.standard_normal((card, dim)) * 30).astype(np.float32)

    def build(g):
        cl = g.clamp(g.input(usage), 1e-5, float("inf"))
        ct = g.cont(g.transpose(cl))
        emb = g.div(g.input(esum), ct)
        return [g.cpy(emb, g.new(F32, dim, card))]
    ref, got, _ = gu.compare(build, rtol=0, atol_rel=0)
    assert np.array_equal(ref[0], got[0])
    assert np.isfinite(got[0]).all() and np.abs(got[0]).max() > 1e5     # the dead codes divide by 1e-5


def test_second_command_stream_and_stream_ordered_read_back():
    # ggml_backend_mi355x_init_stream: a backend handle with its own HIP stream / plan cache on the same GPU; buffers of either handle are plain device
    # memory to both. ggml_backend_tensor_get_async + ggml_backend_event_*: the copy is queued behind the submitted graphs and handed over by the
    # event wait (the run-ahead frame loop reads a step's tokens this way).
    import ctypes as C
    L = gu.lib()
    g = gu.Graph("hip")
    try:
        x = g.input(np.arange(16, dtype=np.float32))
        y = g.scale(g.add(x, x), 0.5)
        z = g.mul(y, y)
        g.build([z])
        g.alloc()
        be2 = L.ggml_backend_mi355x_init_stream(g.backend)
        assert be2 and L.ggml_backend_name(be2) != L.ggml_backend_name(g.backend)
        assert L.ggml_backend_mi355x_get_stream(be2) != L.ggml_backend_mi355x_get_stream(g.backend)
        ev = L.ggml_backend_event_new(L.ggml_backend_get_device(g.backend))
        for rep in range(3):                                    # plan, capture, replay - on the second stream
            g.set(x, np.arange(16, dtype=np.float32) + rep)
            L.ggml_backend_synchronize(g.backend)               # the upload went through the first handle's queue
            assert L.ggml_backend_graph_compute(be2, g.graph) == 0
            out = (C.c_float * 16)()
            L.ggml_backend_tensor_get_async(be2, z, out, 0, 64)
            L.ggml_backend_event_record(ev, be2)
            L.ggml_backend_event_synchronize(ev)
            assert np.array_equal(np.array(out[:]), (np.arange(16, dtype=np.float32) + rep) ** 2), rep
        # device-side ordering between the two streams: the first handle's stream waits for the event recorded on the second
        L.ggml_backend_event_record(ev, be2)
        L.ggml_backend_event_wait(g.backend, ev)
        L.ggml_backend_synchronize(g.backend)
        # several reads in flight, delivered in order by one synchronize
        outs = [(C.c_float * 4)() for _ in range(5)]
        for i, o in enumerate(outs):
            L.ggml_backend_tensor_get_async(be2, z, o, 16 * (i % 4), 16)
        L.ggml_backend_synchronize(be2)
        for i, o in enumerate(outs):
            assert np.array_equal(np.array(o[:]), (np.arange(4 * (i % 4), 4 * (i % 4) + 4, dtype=np.float32) + 2) ** 2)
        L.ggml_backend_event_free(ev)
        L.ggml_backend_free(be2)
        assert L.ggml_backend_mi355x_init_stream(L.ggml_backend_init_by_type(gu.pkg.DEV_CPU, None)) is None    # other backends: NULL, the caller keeps one stream
    finally:
        g.free()


@pytest.mark.parametrize("wt", ["f32", "q8_0"])
def test_voice_condition_graph(wt):
    # the one-shot conditioning graph of moshi-tts (src/moshi.cpp:296-366): two (embedding row -> output projection) terms summed into
    # `condition_sum`; the speaker latents [n, 512] transposed, projected to the model width (T = n activation rows: the batched mat-mul),
    # written over the first n rows of five repeats of the learnt padding through a cpy into a view, plus sinusoidal position embeddings ->
    # `condition_cross`; both results copied into separately allocated tensors (ScratchContext::build_forward_expand(src, dst))
    dim, cdim, sdim, n = 2048, 256, 512, 25

    def build(g):
        r = np.random.default_rng(5)                                   # the same weights on both executors

        def weight(rows, k):
            if wt == "f32":
                return g.input((r.standard_normal((rows, k)) / np.sqrt(k)).astype(np.float32))
            return g.input_raw(gu.random_q8_0(r, rows, k), Q8_0, k, rows)
        cfg_proj, control_proj, speaker_proj = weight(dim, cdim), weight(dim, cdim), weight(dim, sdim)
        cfg_embed = g.input(r.standard_normal((7, cdim)).astype(np.float32))
        control_embed = g.input(r.standard_normal((1, cdim)).astype(np.float32))
        padding = g.input(r.standard_normal((1, dim)).astype(np.float32))
        speaker_wavs = g.input(r.standard_normal((sdim, n)).astype(np.float32))      # ne = [n, 512]
        cfg_cond = g.mul_mat(cfg_proj, g.get_rows(cfg_embed, g.input(np.array([2], np.int32), I32)))
        control_cond = g.mul_mat(control_proj, g.get_rows(control_embed, g.input(np.array([0], np.int32), I32)))
        condition_sum = g.add(cfg_cond, control_cond)
        wavs_b = g.mul_mat(speaker_proj, g.cont(g.transpose(speaker_wavs)))          # [dim, n]
        cond = g.repeat_4d(padding, dim, n * 5, 1, 1)
        nb1 = cond.contents.nb[1]
        speaker_0 = g.cpy(wavs_b, g.view_2d(cond, dim, n, nb1, 0))
        cond = g.view_2d(speaker_0, dim, n * 5, nb1, 0)
        positions = g.input(np.arange(n * 5, dtype=np.float32))
        condition_cross = g.add(cond, g.scale(g.timestep_embedding(positions, dim, 10000), 1.0))
        out_sum, out_cross = g.new(F32, dim, 1), g.new(F32, dim, n * 5)
        return [g.cpy(condition_sum, out_sum), g.cpy(condition_cross, out_cross)]
    ref, got, _ = gu.compare(build, atol_rel=2e-6)
    assert ref[1].shape[-2:] == (n * 5, dim) and np.abs(ref[1][..., n:, :]).max() > 0     # padding rows + positions, speaker rows in front


@pytest.mark.parametrize("wt", ["q8_0", "q4_0"])
@pytest.mark.parametrize("K,M,T", [(256, 16, 2), (512, 40, 5), (2048, 250, 16), (2048, 6144, 17), (2048, 96, 32), (512, 64, 33), (8448, 128, 64), (768, 72, 9),
                                   (1024, 8200, 20)])
def test_batched_q80_q40_matmul_int8_mfma(wt, K, M, T):
    # prompt prefill of the `-q q8_0` checkpoints (tts / stt) and of the loader's Q4_K -> Q4_0 fall-back: T activation rows, each quantised
    # to Q8_0 as ggml does, one v_mfma_i32_16x16x32_i8 per 32-wide block and 16 x 16 tile; ragged M, T and tile counts (K = 768: three
    # groups of eight blocks against tiles of two / four)
    r = np.random.default_rng(K + M + T + len(wt))
    x = (r.standard_normal((T, K)) * r.uniform(0.2, 3.0, (T, 1))).astype(np.float32)
    gt = Q8_0 if wt == "q8_0" else Q4_0
    wraw = (gu.random_q8_0 if wt == "q8_0" else gu.random_q4_0)(r, M, K)

    def build(g):
        return [g.mul_mat(g.input_raw(wraw, gt, K, M), g.input(x))]
    gu.compare(build, atol_rel=2e-6)
    _, st = gu.run_graph("hip", build)
    assert st.kernels_in_last_plan == 1, st.kernels_in_last_plan      # row quantiser + MFMA mat-mul are one plan step; the generic mul_mat is not used


@pytest.mark.parametrize("wt", ["q8_0", "q4_0"])
def test_batched_q80_q40_matmul_fused_rows(wt):
    # the prefill layer shapes at the 2048-wide tts / stt widths: y = res + W (alpha * rms_norm(x)) and y = res + W (silu(h[:n]) * h[n:])
    r = np.random.default_rng(17 + len(wt))
    gt = Q8_0 if wt == "q8_0" else Q4_0
    gen = gu.random_q8_0 if wt == "q8_0" else gu.random_q4_0
    K, M, T = 2048, 6144, 24
    x = (r.standard_normal((T, K)) * r.uniform(0.2, 3.0, (T, 1))).astype(np.float32)
    alpha = (1.0 + 0.1 * r.standard_normal((1, K))).astype(np.float32)
    res = r.standard_normal((T, M)).astype(np.float32)
    wraw = gen(r, M, K)

    def build(g):
        xn = g.mul(g.rms_norm(g.input(x), 1e-8), g.input(alpha))
        return [g.add(g.input(res), g.mul_mat(g.input_raw(wraw, gt, K, M), xn))]
    gu.compare(build, atol_rel=2e-6)
    _, st = gu.run_graph("hip", build)
    assert st.kernels_in_last_plan == 1, st.kernels_in_last_plan

    n, M2 = 5632, 2048
    h = r.standard_normal((T, 2 * n)).astype(np.float32)
    res2 = r.standard_normal((T, M2)).astype(np.float32)
    wraw2 = gen(r, M2, n)

    def build2(g):
        hh = g.input(h)
        t = hh.contents
        left = g.view_4d(hh, t.ne[0] // 2, 1, t.ne[1], t.ne[2], t.nb[1] // 2, t.nb[1], t.nb[2], 0)
        right = g.view_4d(hh, t.ne[0] // 2, 1, t.ne[1], t.ne[2], t.nb[1] // 2, t.nb[1], t.nb[2], t.nb[1] // 2)
        y = g.mul_mat(g.input_raw(wraw2, gt, n, M2), g.mul(g.silu(left), right))
        return [g.add(g.input(res2), g.reshape_3d(y, M2, T, 1))]
    gu.compare(build2, atol_rel=2e-6)
    _, st = gu.run_graph("hip", build2)
    assert st.kernels_in_last_plan == 1, st.kernels_in_last_plan


@pytest.mark.parametrize("K,M,T", [(512, 96, 5), (4096, 8192, 32), (1024, 72, 17)])
def test_batched_q4k_matmul_rmsnorm_rows_and_residual(K, M, T):
    # the batched prefill layer shape: y = res + W (rms_norm(x) * alpha) for T rows: the norm runs inside the row quantiser, the add in the epilogue
    r = np.random.default_rng(K + M + T + 1)
    x = (r.standard_normal((T, K)) * r.uniform(0.2, 3.0, (T, 1))).astype(np.float32)
    alpha = (1.0 + 0.1 * r.standard_normal((1, K))).astype(np.float32)
    res = r.standard_normal((T, M)).astype(np.float32)
    wraw = gu.random_q4_K(r, M, K)

    def build(g):
        xn = g.mul(g.rms_norm(g.input(x), 1e-8), g.input(alpha))
        return [g.add(g.input(res), g.mul_mat(g.input_raw(wraw, Q4_K, K, M), xn))]
    gu.compare(build, atol_rel=2e-6)
    _, st = gu.run_graph("hip", build)
    assert st.kernels_in_last_plan == 1, st.kernels_in_last_plan      # one fused step: row quantiser + mat-mul, nothing else


@pytest.mark.parametrize("n,M,T", [(512, 80, 6), (11264, 64, 20)])
def test_batched_q4k_matmul_gated_rows_and_residual(n, M, T):
    # linear_out of the gated FFN for T rows (gating.h:16-37): silu(h[:n]) * h[n:] handed over as [n, 1, T], folded back to [M, T] and added
    r = np.random.default_rng(n + M + T)
    h = r.standard_normal((T, 2 * n)).astype(np.float32)
    res = r.standard_normal((T, M)).astype(np.float32)
    wraw = gu.random_q4_K(r, M, n)

    def build(g):
        hh = g.input(h)
        t = hh.contents
        left = g.view_4d(hh, t.ne[0] // 2, 1, t.ne[1], t.ne[2], t.nb[1] // 2, t.nb[1], t.nb[2], 0)
        right = g.view_4d(hh, t.ne[0] // 2, 1, t.ne[1], t.ne[2], t.nb[1] // 2, t.nb[1], t.nb[2], t.nb[1] // 2)
        y = g.mul_mat(g.input_raw(wraw, Q4_K, n, M), g.mul(g.silu(left), right))
        return [g.add(g.input(res), g.reshape_3d(y, M, T, 1))]
    gu.compare(build, atol_rel=2e-6)
    _, st = gu.run_graph("hip", build)
    assert st.kernels_in_last_plan == 1, st.kernels_in_last_plan


@pytest.mark.parametrize("kind", ["q4_K", "q8_0", "q4_0"])
def test_quantised_matvec_against_the_numpy_golden_vectors(kind):
    # the device against tests/golden/quant.npz directly (independent numpy block arithmetic, tests/golden/make_golden.py), not via the oracle
    import os
    Q = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quant.npz"))
    w = Q[kind + "_w"]
    gt = {"q4_K": gu.Q4_K, "q8_0": gu.Q8_0, "q4_0": gu.Q4_0}[kind]

    def build(g):
        return [g.mul_mat(g.input_raw(w, gt, Q["x"].size, w.shape[0]), g.input(Q["x"]))]
    for flags in (0, 1):
        y = gu.run_graph("hip", build, flags=flags)[0][0].reshape(-1)
        ref = Q[kind + "_y"]
        assert np.abs(y - ref).max() <= 2e-6 * np.abs(ref).max(), (kind, flags, np.abs(y - ref).max())


@pytest.mark.parametrize("n,k,case", [(32000, 25, "random"), (2048, 250, "random"), (2048, 250, "ties"), (32000, 25, "peaked"), (2049, 250, "zeros"), (300, 250, "random")])
def test_fused_sampler_matches_the_node_chain(n, k, case):
    # the one-launch sampler (k_sample_topk) at the real shapes of moshi_sample_token (text: 32000 logits, top 25; audio: 2048, top 250), incl.
    # exact ties inside and at the top-k cut, a distribution so peaked that most probabilities underflow to 0 (ties at zero: lower index first),
    # against the oracle's node-by-node chain; several noise draws each
    r = np.random.default_rng(n + k)
    logits = (r.standard_normal((1, n)) * 3).astype(np.float32)
    if case == "ties":
        logits[0, r.integers(0, n, 600)] = np.float32(1.25)          # hundreds of equal values straddling the cut
        logits[0, 5] = logits[0, 1900] = logits[0].max()              # and an exact tie for first place
    elif case == "peaked":
        logits *= 40
    elif case == "zeros":
        logits[0, 200:] = -1e4                                        # expf underflows: exact zeros from index 200 on, k = 250 reaches into them
    for draw in range(3):
        noise = r.exponential(1.0, (1, k)).astype(np.float32)

        def build(g):
            probs = g.soft_max(g.scale(g.input(logits), 1.0 / 0.8))
            indices = g.argsort_top_k(probs, k)
            rows = g.get_rows(g.cont(g.permute(probs, 1, 0, 2, 3)), indices)
            p2 = g.permute(rows, 1, 0, 2, 3)
            in2 = g.reshape_2d(p2, p2.contents.ne[0], p2.contents.ne[1] * p2.contents.ne[2] * p2.contents.ne[3])
            nxt = g.argmax(g.div(in2, g.input(noise)))
            nxt4 = g.reshape_4d(nxt, nxt.contents.ne[0], p2.contents.ne[1], p2.contents.ne[2], p2.contents.ne[3])
            return [g.get_rows(g.cont(g.permute(indices, 1, 0, 2, 3)), nxt4)]
        ref, got, st = gu.compare(build)
        assert st.fused_nodes_in_last_plan >= 12, "the sampler chain was not fused"
        plain, _ = gu.run_graph("hip", build, flags=1)
        assert np.array_equal(plain[0], ref[0])


@pytest.mark.parametrize("dst", [Q8_0, Q4_0, Q4_K])
@pytest.mark.parametrize("src", [F32, BF16, F16])
def test_requantising_cast_on_the_device_is_bit_exact(src, dst):
    # WeightLoader's load-time cast (src/loader.h:160-187: ggml_cast of a float checkpoint tensor to the -q type) run on the MI355X backend:
    # the bytes must equal the host quantiser's (ggml_quantize_row, what the host device runs for the same graph) and the oracle's cpy
    import ctypes as C
    r = np.random.default_rng(src * 100 + dst)
    K, M = 1024, 24
    x = (r.standard_normal((M, K)) * np.exp(r.standard_normal((M, K)))).astype(np.float32)
    x[1, 256:512] = 0.0; x[2, :256] = 2.5; x[3, 32:64] = 0.0; x[4] *= 1e-6; x[5, 256:288] = -1.0
    x[6, ::7] = 0.0
    results = {}
    for kind in ("oracle", "hip"):
        g = gu.Graph(kind)
        t = g.cast(g.input(x, src), dst)
        g.build([t])
        g.alloc()
        g.compute()
        n = g.L.ggml_nbytes(t)
        raw = C.create_string_buffer(n)
        g.L.ggml_backend_tensor_get(t, raw, 0, n)
        results[kind] = np.frombuffer(raw.raw, np.uint8).copy()
        g.free()
    # host quantiser on the values the source type can hold
    xs = gu.decode(gu.encode(x, src), src, x.shape)
    L = gu.lib()
    host = np.zeros(results["hip"].size, np.uint8)
    rb = host.size // M
    for i in range(M):
        row = np.ascontiguousarray(xs[i], np.float32)
        L.ggml_quantize_row(dst, row.ctypes.data, host[i * rb:(i + 1) * rb].ctypes.data, K)
    assert np.array_equal(results["hip"], host), f"device vs host quantiser: {np.count_nonzero(results['hip'] != host)} bytes differ"
    assert np.array_equal(results["oracle"], host), "oracle vs host quantiser"


def test_per_step_projection_split_requantised_on_the_device():
    # the Depth transformer's per-step in_proj split (transformer.h:780-848): row ranges of one float weight, each cast to the -q type
    import ctypes as C
    r = np.random.default_rng(99)
    K, M, steps = 512, 3 * 64, 3
    w = r.standard_normal((M, K)).astype(np.float32)
    outs = {}
    for kind in ("oracle", "hip"):
        g = gu.Graph(kind)
        wt = g.input(w, BF16)
        parts = []
        for k in range(steps):
            rows = g.view_2d(wt, K, M // steps, wt.contents.nb[1], wt.contents.nb[1] * (M // steps) * k)
            parts.append(g.cast(rows, Q4_K))
        g.build(parts)
        g.alloc()
        g.compute()
        got = []
        for t in parts:
            n = g.L.ggml_nbytes(t)
            raw = C.create_string_buffer(n)
            g.L.ggml_backend_tensor_get(t, raw, 0, n)
            got.append(np.frombuffer(raw.raw, np.uint8).copy())
        outs[kind] = got
        g.free()
    for a, b in zip(outs["oracle"], outs["hip"]):
        assert np.array_equal(a, b)
    assert not np.array_equal(outs["hip"][0], outs["hip"][1])


def test_paired_gate_ffn_matches_oracle_and_can_be_switched_off():
    # the paired form (default): linear_in's workgroups take matching rows of both halves and write silu(l) * r themselves (no [2 F] intermediate, no gate
    # kernel); MI355X_PAIRED_GATE=0 restores the separate gate kernel. Both checked at the Temporal FFN shape, each in a fresh process.
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, ggml_util as gu
from ggml_util import Q4_K
K, F = 4096, 11264
r = np.random.default_rng(5)
x = r.standard_normal((1, K)).astype(np.float32)
alpha = (1.0 + 0.1 * r.standard_normal((1, K))).astype(np.float32)
res = r.standard_normal((1, K)).astype(np.float32)
w_in = gu.random_q4_K(r, 2 * F, K); w_out = gu.random_q4_K(r, K, F)
def build(g):
    xn = g.mul(g.input(alpha), g.rms_norm(g.input(x), 1e-8))
    h = g.mul_mat(g.input_raw(w_in, Q4_K, K, 2 * F), xn)
    hh = h.contents
    l = g.view_4d(h, hh.ne[0] // 2, 1, hh.ne[1], hh.ne[2], hh.nb[1] // 2, hh.nb[1], hh.nb[2], 0)
    rr = g.view_4d(h, hh.ne[0] // 2, 1, hh.ne[1], hh.ne[2], hh.nb[1] // 2, hh.nb[1], hh.nb[2], hh.nb[1] // 2)
    y = g.mul_mat(g.input_raw(w_out, Q4_K, F, K), g.mul(g.silu(l), rr))
    return [g.add(g.input(res), y)]
ref, got, st = gu.compare(build, atol_rel=3e-5)
assert st.fused_nodes_in_last_plan >= 6
print("OK", st.kernels_in_last_plan)
'''
    out = {}
    for flag in ("0", None):
        env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.abspath(__file__)))
        env.pop("MI355X_PAIRED_GATE", None)
        if flag:
            env["MI355X_PAIRED_GATE"] = flag
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "OK" in r.stdout, (r.stdout + r.stderr)[-2000:]
        out[flag] = int(r.stdout.split()[-1])
    assert out[None] == out["0"] - 1, f"paired form should save exactly the gate kernel: {out}"


# ---- round 4: planner fusions found on the tts-shaped configuration ------------------------------------------------------------------------------------
def test_embedding_sum_of_35_terms_starting_from_a_computed_vector():
    # lm_utils.h:48-66 + lm.h:853-869 at tts shape: the demuxed text embedding (two projections, their sum) followed by 33 scaled codebook rows - the
    # fused sum takes up to 40 terms and may start from any F32 vector; same left-to-right float additions as the node chain
    r = np.random.default_rng(11)
    K, rows, nterm = 512, 40, 33
    tabs = [gu.random_q8_0(r, rows, K) for _ in range(nterm)]
    idx = r.integers(0, rows, nterm).astype(np.int32)
    scales = np.where(r.random(nterm) < 0.2, 0.0, 1.0).astype(np.float32)
    w1, w2 = gu.random_q8_0(r, K, 256), gu.random_q8_0(r, K, 256)
    x1, x2 = rnd(1, 256), rnd(1, 256)

    def build(g):
        base = g.add(g.mul_mat(g.input_raw(w1, Q8_0, 256, K), g.input(x1)), g.mul(g.mul_mat(g.input_raw(w2, Q8_0, 256, K), g.input(x2)), g.input(np.array([0.5], np.float32))))
        acc = base
        for t in range(nterm):
            e = g.mul(g.get_rows(g.input_raw(tabs[t], Q8_0, K, rows), g.input(idx[t:t + 1], I32)), g.input(scales[t:t + 1]))
            acc = g.add(acc, e)
        return [acc]
    ref, got, st = gu.compare(build, atol_rel=0, rtol=0)   # bit-exact against the oracle (the generic mul_mat of an unfused plan sums its blocks in another order)
    assert st.fused_nodes_in_last_plan >= 3 * nterm, f"the embedding sum was not fused ({st.fused_nodes_in_last_plan} nodes)"


@pytest.mark.parametrize("tab_type", [Q8_0, F32])
def test_low_rank_embedding_row_through_a_small_q8_0_projection(tab_type):
    # lm_utils.h:157-217: get_rows of a 128-wide table -> 128 -> 1024 Q8_0 linear -> cast to F32: one launch with the arithmetic of get_rows_kernel,
    # convert_rows_kernel (Q8_0 re-quantisation of the row) and the reference's vec_dot_q8_0_q8_0 (block terms added in block order)
    r = np.random.default_rng(5)
    K, M, rows = 128, 1024, 50
    tab = gu.random_q8_0(r, rows, K) if tab_type == Q8_0 else (r.standard_normal((rows, K)) * 0.3).astype(np.float32)
    w = gu.random_q8_0(r, M, K)
    for i in (0, 17, rows - 1):
        def build(g):
            t = g.input_raw(tab, Q8_0, K, rows) if tab_type == Q8_0 else g.input(tab)
            y = g.mul_mat(g.input_raw(w, Q8_0, K, M), g.get_rows(t, g.input(np.array([i], np.int32), I32)))
            return [g.add(g.cast(y, F32), g.input(np.ones((1, M), np.float32)))]
        ref, got, st = gu.compare(build, atol_rel=0, rtol=0)   # bit-exact against the oracle: the blocks' terms are added in vec_dot_q8_0_q8_0's order
        assert st.fused_nodes_in_last_plan >= 3
        plain, _ = gu.run_graph("hip", build, flags=1)          # (the generic mul_mat_kernel sums them as a butterfly: 1 ulp)
        assert np.abs(plain[0] - got[0]).max() <= 2e-6 * np.abs(got[0]).max()


def test_layer_norm_with_weight_and_bias_as_one_launch():
    # transformer.h:936-944 (norm_cross): ggml_norm -> mul(w) -> add(b) over one row that no mat-vec prologue takes (its consumer here is a plain scale)
    r = np.random.default_rng(9)
    x, w, b = rnd(1, 2048), (1 + 0.1 * r.standard_normal((1, 2048))).astype(np.float32), (0.02 * r.standard_normal((1, 2048))).astype(np.float32)

    def build(g):
        y = g.add(g.mul(g.norm(g.input(x), 1e-5), g.input(w)), g.input(b))
        return [g.scale(y, 2.0)]
    ref, got, st = gu.compare(build, atol_rel=0, rtol=0)
    assert st.fused_nodes_in_last_plan >= 3
    plain, _ = gu.run_graph("hip", build, flags=1)
    assert np.array_equal(plain[0], got[0])
