"""CPU: host-side object model behind include/ggml.h — conversions, shapes/strides of the op builders, view
aliasing, graph expansion order, context reset, GGUF round trip."""
import ctypes as C
import os
import tempfile

import numpy as np

import ggml_util as gu
from ggml_util import BF16, F16, F32, I32, Q4_K, Q8_0

L = gu.lib()
pkg = gu.pkg


def new_ctx(mb=8, no_alloc=True):
    return L.ggml_init(pkg.InitParams(mb << 20, None, no_alloc))


def ne(t):
    return [int(t.contents.ne[i]) for i in range(4)]


def nb(t):
    return [int(t.contents.nb[i]) for i in range(4)]


def test_fp16_bf16_conversions_match_numpy_bit_for_bit():
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(4000).astype(np.float32) * 10 ** rng.uniform(-8, 5, 4000).astype(np.float32),
                           np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 6.1e-5, 5.96e-8, 2.98e-8, 2.99e-8, np.inf, -np.inf], np.float32)])
    for v in vals:
        h = L.ggml_fp32_to_fp16(float(v))
        assert h == int(np.float32(v).astype(np.float16).view(np.uint16)), v
        b = L.ggml_fp32_to_bf16(float(v))
        assert b == int(gu.f32_to_bf16_bits(np.array([v]))[0]), v
    for h in list(range(0, 65536, 7)) + [0x7c00, 0xfc00, 1, 0x03ff, 0x0400]:
        a = np.array([h], np.uint16).view(np.float16).astype(np.float32)[0]
        got = L.ggml_fp16_to_fp32(h)
        assert (np.isnan(a) and np.isnan(got)) or a == got, h


def test_type_traits_and_nbytes():
    assert L.ggml_type_size(Q4_K) == 144 and L.ggml_blck_size(Q4_K) == 256
    assert L.ggml_type_size(Q8_0) == 34 and L.ggml_row_size(Q8_0, 64) == 68
    assert L.ggml_type_name(BF16) == b"bf16" and L.ggml_type_name(Q4_K) == b"q4_K"
    ctx = new_ctx()
    t = L.ggml_new_tensor_2d(ctx, Q4_K, 512, 3)
    assert L.ggml_nbytes(t) == 2 * 144 * 3 and nb(t)[:2] == [144, 288]
    v = L.ggml_view_2d(ctx, t, 512, 2, t.contents.nb[1], t.contents.nb[1])   # row-range view (torch.h:110-113)
    assert L.ggml_nbytes(v) == 2 * 288 and v.contents.view_offs == 288
    L.ggml_free(ctx)


def test_tensor_overhead_is_exact():
    # libmoshi sizes contexts as ggml_tensor_overhead() * n (src/context.h:206, 750)
    n = 7
    ctx = L.ggml_init(pkg.InitParams(L.ggml_tensor_overhead() * n, None, True))
    ts = [L.ggml_new_tensor_1d(ctx, F32, 10) for _ in range(n)]
    assert L.ggml_used_mem(ctx) == L.ggml_tensor_overhead() * n
    it, seen = L.ggml_get_first_tensor(ctx), 0
    while it:
        seen += 1
        it = L.ggml_get_next_tensor(ctx, it)
    assert seen == n
    L.ggml_reset(ctx)
    assert L.ggml_used_mem(ctx) == 0 and not L.ggml_get_first_tensor(ctx)
    L.ggml_free(ctx)


def test_builder_shapes():
    ctx = new_ctx()
    a = L.ggml_new_tensor_4d(ctx, F32, 128, 1, 32, 1)
    w = L.ggml_new_tensor_2d(ctx, Q4_K, 4096, 12288)
    x = L.ggml_new_tensor_2d(ctx, F32, 4096, 1)
    assert ne(L.ggml_mul_mat(ctx, w, x)) == [12288, 1, 1, 1]
    p = L.ggml_permute(ctx, a, 0, 2, 1, 3)
    assert ne(p) == [128, 32, 1, 1] and nb(p)[1] == nb(a)[2]
    assert L.ggml_is_contiguous(p)           # only size-1 dims moved
    t = L.ggml_transpose(ctx, L.ggml_new_tensor_2d(ctx, F32, 5, 3))
    assert ne(t)[:2] == [3, 5] and not L.ggml_is_contiguous(t) and L.ggml_is_transposed(t)
    cache = L.ggml_new_tensor_3d(ctx, BF16, 128, 3000, 32)
    rows = L.ggml_new_tensor_3d(ctx, F32, 128, 1, 32)
    idx = L.ggml_new_tensor_1d(ctx, I32, 1)
    sr = L.ggml_set_rows(ctx, cache, rows, idx)
    assert ne(sr) == ne(cache) and C.addressof(sr.contents.view_src.contents) == C.addressof(cache.contents)
    kern = L.ggml_new_tensor_3d(ctx, F16, 7, 512, 1024)
    data = L.ggml_new_tensor_2d(ctx, F32, 8, 512)
    assert ne(L.ggml_conv_1d(ctx, kern, data, 1, 0, 1)) == [2, 1024, 1, 1]
    tk = L.ggml_new_tensor_3d(ctx, F32, 16, 512, 1024)
    td = L.ggml_new_tensor_2d(ctx, F32, 2, 1024)
    assert ne(L.ggml_conv_transpose_1d(ctx, tk, td, 8, 0, 1)) == [24, 512, 1, 1]
    emb = L.ggml_new_tensor_2d(ctx, gu.Q4_0, 4096, 2049)
    assert ne(L.ggml_get_rows(ctx, emb, idx)) == [4096, 1, 1, 1]
    logits = L.ggml_new_tensor_2d(ctx, F32, 2048, 1)
    assert ne(L.ggml_argmax(ctx, logits)) == [1, 1, 1, 1] and L.ggml_argmax(ctx, logits).contents.type == I32
    assert ne(L.ggml_argsort_top_k(ctx, logits, 250)) == [250, 1, 1, 1]
    assert ne(L.ggml_timestep_embedding(ctx, L.ggml_new_tensor_1d(ctx, F32, 2), 64, 10000)) == [64, 2, 1, 1]
    L.ggml_free(ctx)


def test_views_resolve_to_root_with_wrapping_offsets():
    ctx = new_ctx()
    toks = L.ggml_new_tensor_1d(ctx, I32, 8)
    v = L.ggml_view_1d(ctx, toks, 1, 0)
    for _ in range(7):
        v = L.ggml_view_1d(ctx, v, 1, 4)
    assert v.contents.view_offs == 28 and C.addressof(v.contents.view_src.contents) == C.addressof(toks.contents)
    back = L.ggml_view_1d(ctx, v, 8, (1 << 64) - 28)     # lm.h:527 negative offset
    assert back.contents.view_offs == 0
    clamp = L.ggml_clamp(ctx, L.ggml_new_tensor_1d(ctx, F32, 4), 0.0, 1.0)
    assert clamp.contents.view_src                        # in place (transformer.h:269)
    L.ggml_free(ctx)


def test_graph_expansion_is_dfs_post_order_and_dedups():
    ctx = new_ctx()
    a = L.ggml_new_tensor_1d(ctx, F32, 4)
    b = L.ggml_scale(ctx, a, 2.0)
    c = L.ggml_neg(ctx, b)
    d = L.ggml_add(ctx, c, b)        # b shared
    side = L.ggml_cpy(ctx, b, L.ggml_new_tensor_1d(ctx, F32, 4))
    g = L.ggml_new_graph_custom(ctx, 64, False)
    L.ggml_build_forward_expand(g, side)     # side branch first (conv.h:75-76 relies on call order)
    L.ggml_build_forward_expand(g, d)
    names = [C.addressof(L.ggml_graph_node(g, i).contents) for i in range(L.ggml_graph_n_nodes(g))]
    assert names == [C.addressof(t.contents) for t in (b, side, c, d)]
    assert L.ggml_graph_n_leafs(g) == 2
    L.ggml_free(ctx)


def test_gguf_round_trip_through_host_buffers():
    cpu = L.ggml_backend_init_by_type(pkg.DEV_CPU, None)
    ctx = new_ctx()
    rng = np.random.default_rng(0)
    t1 = L.ggml_new_tensor_2d(ctx, F32, 8, 3)
    L.ggml_set_name(t1, b"lm.out_norm.alpha")
    t2 = L.ggml_new_tensor_2d(ctx, Q4_K, 256, 4)
    L.ggml_set_name(t2, b"lm.transformer.layers.0.self_attn.in_projs.0.weight")
    buf = L.ggml_backend_alloc_ctx_tensors(ctx, cpu)
    d1 = rng.standard_normal((3, 8)).astype(np.float32)
    d2 = gu.random_q4_K(rng, 4, 256)
    L.ggml_backend_tensor_set(t1, d1.tobytes(), 0, d1.nbytes)
    L.ggml_backend_tensor_set(t2, d2.tobytes(), 0, d2.nbytes)
    gg = L.gguf_init_empty()
    L.gguf_add_tensor(gg, t1)
    L.gguf_add_tensor(gg, t2)
    path = os.path.join(tempfile.mkdtemp(), "w.gguf")
    assert L.gguf_write_to_file(gg, path.encode(), False)
    L.gguf_free(gg)
    meta = C.c_void_p()
    g2 = L.gguf_init_from_file(path.encode(), pkg.GGUFInitParams(True, C.pointer(meta)))
    assert g2 and L.gguf_get_n_tensors(g2) == 2 and L.gguf_get_version(g2) == 3
    raw = open(path, "rb").read()
    off = L.gguf_get_data_offset(g2)
    for i, ref in enumerate((d1.tobytes(), d2.tobytes())):
        o, n = off + L.gguf_get_tensor_offset(g2, i), L.gguf_get_tensor_size(g2, i)
        assert raw[o:o + n] == ref
    t = L.ggml_get_tensor(meta, b"lm.out_norm.alpha")
    assert t and ne(t)[:2] == [8, 3] and L.gguf_get_tensor_type(g2, 1) == Q4_K
    L.gguf_free(g2)
    L.ggml_free(meta)
    L.ggml_backend_buffer_free(buf)
    L.ggml_free(ctx)
    L.ggml_backend_free(cpu)
