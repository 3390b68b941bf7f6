"""The reference's checkpoint path end to end (/root/reference/src/loader.h:85-99 from_gguf, 227-233 save_gguf, 235-271 load_gguf; src/moshi.cpp:625-694): a
model's weights are written with gguf_write_to_file, read back with gguf_init_from_file, uploaded tensor by tensor with ggml_backend_tensor_set and RUN -
LM steps and the codec - against the oracle on the generated weights. The configuration includes a linear whose row length is not a multiple of 256, i.e.
one that took the loader's Q4_K -> Q4_0 fall-back (loader.h:162-173) when the checkpoint was quantised."""
import ctypes as C
import os

import numpy as np
import pytest

import ggml_util as gu
import hot_util as hu

L = hu.L
pkg = hu.pkg


def config():
    cfg = hu.hot.tiny(L)                       # Q4_K linears, Q4_0 embeddings, Mimi codec with 3 levels
    cfg.depformer_low_rank = 128               # Depth embeddings of width 128 + a [128 -> dep_dim] linear: 128 % 256 != 0 -> stored as Q4_0
    return cfg


def frames(n):
    rng = np.random.default_rng(9)
    return [rng.standard_normal(1920).astype(np.float32) * 0.1 for _ in range(n)]


def from_gguf(kind, cfg, path):
    m = hu.Model.__new__(hu.Model)
    m.cfg, m.kind = cfg, kind
    m.be = hu.make_backend(kind)
    m.m = L.moshi_hot_create_from_gguf(m.be, C.byref(cfg), path.encode())
    assert m.m
    return m


def gguf_directory(path):
    meta = C.c_void_p()
    gg = L.gguf_init_from_file(path.encode(), pkg.GGUFInitParams(True, C.pointer(meta)))
    assert gg
    out = {}
    for i in range(L.gguf_get_n_tensors(gg)):
        out[L.gguf_get_tensor_name(gg, i).decode()] = (L.gguf_get_tensor_type(gg, i), L.gguf_get_tensor_size(gg, i))
    L.gguf_free(gg)
    return out


def write_checkpoint(tmp_path, cfg):
    src = hu.Model("oracle", cfg, seed=0)
    path = os.path.join(str(tmp_path), "tiny.gguf")
    assert L.moshi_hot_save_gguf(src.m, path.encode()) == 1
    return src, path


def test_a_model_read_back_from_its_gguf_file_is_the_same_model_on_the_host_device(tmp_path):
    cfg = config()
    src, path = write_checkpoint(tmp_path, cfg)
    d = gguf_directory(path)
    assert len(set(d)) == len(d) and len(d) > 60, "tensor names must be unique inside the file"
    types = {t for t, _ in d.values()}
    assert gu.Q4_K in types and gu.Q4_0 in types and gu.F32 in types and pkg.F16 in types
    lr = [n for n in d if n.endswith("low_rank.weight")]
    assert lr and all(d[n][0] == gu.Q4_0 for n in lr), "the 128-wide low-rank linears take the Q4_K -> Q4_0 fall-back"
    assert any(len(n) == 8 and all(ch in "0123456789abcdef" for ch in n) for n in d), "long checkpoint names are stored as 8-digit digests (loader.h:120-137)"
    back = from_gguf("oracle", cfg, path)
    fr = frames(4)
    a = [src.sts_frame(f) for f in fr]
    b = [back.sts_frame(f) for f in fr]
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[:3] == y[:3] and np.array_equal(x[3], y[3]), f"frame {i}: the model read back from the file differs"
    assert np.array_equal(src.read("text_logits", cfg.text_card), back.read("text_logits", cfg.text_card))
    src.free(); back.free()


def reference_digest(name):
    """WeightLoader::tensor_name (loader.h:120-137) over src/crc-bbf.h, restated on zlib: crc-bbf is the IEEE CRC-32 (width 32, poly 0x04c11db7, reflected,
    xor-in / xor-out 0xffffffff) = zlib.crc32; the reference's hex loop keeps the LOW nibble of each of the eight bytes of its 64-bit crc_t."""
    import zlib
    if len(name) < 64:                                  # GGML_MAX_NAME
        return name
    crc = zlib.crc32(name.encode())
    return "".join("0123456789abcdef"[(crc >> (8 * i)) & 0xf] for i in range(8))


def test_long_tensor_names_become_the_reference_crc_digest():
    def ours(name):
        buf = C.create_string_buffer(80)
        n = L.moshi_hot_tensor_file_name(name.encode(), buf, 80)
        assert n == len(buf.value)
        return buf.value.decode()
    assert reference_digest("123456789" * 8) == "04180000"   # CRC-32 0x8811a440: bytes 40 a4 11 88 -> low nibbles 0 4 1 8, then the four zero bytes of crc_t
    names = ["lm.transformer.layers.0.self_attn.in_projs.weight", "x" * 63, "x" * 64, "123456789" * 8,
             "mimi.decoder_transformer.transformer.layers.7.self_attn.out_projs.0.weight.extra.long.suffix",
             "lm.depformer.layers.5.gating.15.linear_out.weight.and.then.some.more.characters.to.cross.the.limit"]
    rng = np.random.default_rng(3)
    names += ["".join(chr(int(c)) for c in rng.integers(33, 127, size=int(n))) for n in rng.integers(64, 200, size=40)]
    for n in names:
        assert ours(n) == reference_digest(n), n
    assert ours("x" * 63) == "x" * 63 and len(ours("x" * 64)) == 8 and ours("x" * 64).endswith("0000")


@pytest.mark.gpu
def test_gguf_checkpoint_uploaded_to_the_device_and_run_against_the_oracle(tmp_path):
    cfg = config()
    src, path = write_checkpoint(tmp_path, cfg)
    dev = from_gguf("hip", cfg, path)
    fr = frames(5)
    errs = []
    for i, f in enumerate(fr):
        a, b = src.sts_frame(f), dev.sts_frame(f)
        assert a[:3] == b[:3], f"frame {i}: tokens {a[:3]} (oracle, generated weights) vs {b[:3]} (device, weights from the file)"
        if a[0]:
            assert hu.rel_err(a[3], b[3]) < 2e-3, f"frame {i}: pcm rel err {hu.rel_err(a[3], b[3]):.2e}"
        errs.append(hu.rel_err(src.read("text_logits", cfg.text_card), dev.read("text_logits", cfg.text_card)))
    # summation noise, except on a step where one Q8_K activation value rounds the other way (tests/test_hip_frame.py, module docstring)
    assert np.median(errs) < 1e-4 and max(errs) < 5e-2, f"text logits rel err per frame {errs}"
    # the bytes on the device are the bytes of the file
    for name in ("lm.depformer_emb.0.low_rank.weight", "lm.transformer.layers.0.self_attn.in_projs.weight", "lm.depformer.layers.1.gating.2.linear_in.weight"):
        ta, tb = C.cast(L.moshi_hot_weight(src.m, name.encode()), pkg.TP), C.cast(L.moshi_hot_weight(dev.m, name.encode()), pkg.TP)
        assert ta and tb
        n = L.ggml_nbytes(ta)
        xa, xb = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        L.ggml_backend_tensor_get(ta, xa.ctypes.data, 0, n); L.ggml_backend_tensor_get(tb, xb.ctypes.data, 0, n)
        assert np.array_equal(xa, xb), name
    st = pkg.Stats()
    L.ggml_backend_mi355x_get_stats(dev.be, C.byref(st))
    assert st.graph_replays > 0
    src.free(); dev.free()
