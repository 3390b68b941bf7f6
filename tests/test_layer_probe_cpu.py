"""CPU: mechanics of the teacher-forced layer probe (tests/parity_probe.py) that the -m gpu full-width parity tests are built on. The
"other executor" here is the oracle itself fed an input that differs by float noise, which is exactly what a correct device kernel with
another summation order looks like to the rounding sites."""
import numpy as np

import hot_util as hu
import parity_probe as pp
from ggml_util import Q4_K, Q4_0

L = hu.L


def tiny_lm(lt=Q4_K):
    cfg = hu.hot.tiny(L, linear_type=lt, embed_type=Q4_0)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def test_probe_is_deterministic_and_identical_inputs_give_identical_nodes():
    cfg = tiny_lm()
    m1, m2 = hu.Model("oracle", cfg), hu.Model("oracle", cfg)
    x = np.random.default_rng(0).standard_normal(cfg.dim).astype(np.float32) * 3
    for which, layer, ws, dim in ((0, 1, 0, cfg.dim), (1, 0, 2, cfg.dep_dim)):
        xin = x[:dim]
        a, ya = pp.probe(m1, which, layer, ws, xin, 0)
        b, yb = pp.probe(m2, which, layer, ws, xin, 0)
        st = pp.compare_layer(a, b, f"which {which}", clean_tol=0.0)
        assert np.array_equal(ya, yb) and st["flips"] == 0 and st["tainted"] == 0 and st["nodes"] >= 15 and st["sites"] >= 8
        # a second position: two live slots, the ring row written by the first probe is read back
        a, ya = pp.probe(m1, which, layer, ws, xin * 0.5, 1)
        b, yb = pp.probe(m2, which, layer, ws, xin * 0.5, 1)
        assert np.array_equal(ya, yb) and pp.compare_layer(a, b, "second slot", clean_tol=0.0)["flips"] == 0
    m1.free(); m2.free()


def test_float_noise_is_either_invisible_or_an_accounted_rounding_flip():
    # many noisy copies of the input: every node agrees to summation noise unless a counted flip sits upstream, and flips do occur
    cfg = tiny_lm()
    m1, m2 = hu.Model("oracle", cfg), hu.Model("oracle", cfg)
    rng = np.random.default_rng(1)
    flips = tainted = clean = 0
    for trial in range(12):
        x = rng.standard_normal(cfg.dim).astype(np.float32) * 3
        xn = (x * (1 + rng.standard_normal(cfg.dim).astype(np.float32) * 3e-7)).astype(np.float32)
        a, _ = pp.probe(m1, 0, 0, 0, x, 0)
        b, _ = pp.probe(m2, 0, 0, 0, xn, 0)
        st = pp.compare_layer(a, b, f"trial {trial}", clean_tol=5e-6, taint_tol=5e-2)
        flips += st["flips"]; tainted += st["tainted"]; clean += st["clean"]
    assert clean > 0 and flips > 0 and tainted > 0, (flips, tainted, clean)
    m1.free(); m2.free()


def test_numpy_quantisers_match_the_oracle_rows():
    # the host restatement used for flip counting against the oracle's own quantize_row (Q8_K / Q8_0)
    import ctypes as C
    from __graft_entry__ import load_oracle
    olib = load_oracle().load()
    rng = np.random.default_rng(2)
    v = (rng.standard_normal(1024) * np.exp(rng.standard_normal(1024))).astype(np.float32)
    v[256:512] = 0
    v[700] = -v[701]                       # +a / -a tie for the block maximum: the first one wins
    out = np.zeros(4 * 292, np.uint8)      # block_q8_K: f32 d, 256 int8, 16 int16
    olib.oracle_quantize_row(15, v.ctypes.data, out.ctypes.data, 1024)
    q = out.reshape(4, 292)[:, 4:260].view(np.int8).astype(np.int32).reshape(-1)
    assert np.array_equal(q, pp.q8_K(v))
    out = np.zeros(32 * 34, np.uint8)
    olib.oracle_quantize_row(8, v.ctypes.data, out.ctypes.data, 1024)
    q = out.reshape(32, 34)[:, 2:].view(np.int8).astype(np.int32).reshape(-1)
    d16 = out.reshape(32, 34)[:, :2].copy().view(np.uint16).astype(np.int32).reshape(-1)
    got = pp.q8_0(v)                       # the quants, then one entry per block: the bits of its F16 scale (a rounding site of its own)
    assert np.array_equal(q, got[:1024]) and np.array_equal(d16, got[1024:])
