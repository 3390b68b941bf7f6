"""CPU: known-answer tests for the block formats and integer dot products of the oracle:
hand-built Q4_K / Q8_0 / Q4_0 / Q8_K blocks with answers derived by hand from the format definition, plus
brute-force numpy cross-checks of the mat-vec path. Also checks the product's host (de)quantisers against the
oracle's independent implementations."""
import ctypes as C
import struct

import numpy as np

import ggml_util as gu
from ggml_util import BF16, F16, F32, Q4_0, Q4_K, Q8_0

O = gu.load_oracle().load()
L = gu.lib()
Q8_K = gu.pkg.Q8_K


def f16(x):
    return struct.pack("<e", x)


def deq(raw, t, k):
    return gu.dequantize(np.frombuffer(raw, np.uint8).reshape(1, -1), t, k)[0]


def test_q8_0_block_known_answer():
    qs = np.arange(-16, 16, dtype=np.int8)
    y = deq(f16(0.5) + qs.tobytes(), Q8_0, 32)
    assert np.array_equal(y, qs.astype(np.float32) * 0.5)


def test_q4_0_block_known_answer():
    # low nibbles = elements 0..15, high nibbles = 16..31, value = (nib - 8) * d
    nib_lo, nib_hi = np.arange(16, dtype=np.uint8), np.arange(15, -1, -1, dtype=np.uint8)
    y = deq(f16(2.0) + (nib_lo | (nib_hi << 4)).astype(np.uint8).tobytes(), Q4_0, 32)
    assert np.array_equal(y[:16], (nib_lo.astype(np.float32) - 8) * 2) and np.array_equal(y[16:], (nib_hi.astype(np.float32) - 8) * 2)


def q4k_block(d, dmin, sc, mn, nibbles):
    """nibbles: 256 values 0..15 in element order; sc/mn: 8 six-bit values each."""
    s = np.zeros(12, np.uint8)
    for j in range(4):
        s[j] = (sc[j] & 63) | ((sc[j + 4] >> 4) << 6)
        s[j + 4] = (mn[j] & 63) | ((mn[j + 4] >> 4) << 6)
        s[j + 8] = (sc[j + 4] & 0xF) | ((mn[j + 4] & 0xF) << 4)
    qs = np.zeros(128, np.uint8)
    nib = np.asarray(nibbles, np.uint8).reshape(4, 2, 32)      # group j: sub-block 2j low nibbles, 2j+1 high nibbles
    for j in range(4):
        qs[32 * j:32 * j + 32] = nib[j, 0] | (nib[j, 1] << 4)
    return f16(d) + f16(dmin) + s.tobytes() + qs.tobytes()


def test_q4_K_block_known_answers():
    # all-zero nibbles: w = -dmin * m
    sc, mn = [1, 2, 3, 4, 33, 34, 35, 63], [5, 6, 7, 8, 40, 50, 60, 63]
    y = deq(q4k_block(1.0, 0.5, sc, mn, np.zeros(256)), Q4_K, 256)
    assert np.array_equal(y, np.repeat(-0.5 * np.array(mn, np.float32), 32))
    # all-max nibbles, zero mins: w = d * sc * 15
    y = deq(q4k_block(0.25, 0.0, sc, [0] * 8, np.full(256, 15)), Q4_K, 256)
    assert np.array_equal(y, np.repeat(0.25 * 15 * np.array(sc, np.float32), 32))
    # single-hot scale: only sub-block 5 non-zero, nibble ramp
    one = [0, 0, 0, 0, 0, 7, 0, 0]
    ramp = np.tile(np.arange(16), 16)
    y = deq(q4k_block(2.0, 0.0, one, [0] * 8, ramp), Q4_K, 256)
    exp = np.zeros(256, np.float32)
    exp[160:192] = 2.0 * 7 * ramp[160:192]
    assert np.array_equal(y, exp)


def test_q8_K_activation_quantiser_known_answer():
    x = np.zeros(256, np.float32)
    x[3], x[100], x[200] = -2.0, 1.0, 0.5
    out = np.zeros(292, np.uint8)
    O.oracle_quantize_row(Q8_K, x.ctypes.data, out.ctypes.data, 256)
    d = np.frombuffer(out[:4].tobytes(), np.float32)[0]
    q = out[4:260].view(np.int8)
    bs = np.frombuffer(out[260:292].tobytes(), np.int16)
    # iscale = -127 / max where max is the signed extreme (-2): q(-2) = 127, d = 1/iscale = -2/127... sign carried by d
    assert q[3] == -127 or q[3] == 127
    assert np.isclose(d * q[3], -2.0, rtol=1e-6) and np.isclose(d * q[100], 1.0, atol=abs(d)) and q[0] == 0
    assert bs[0] == q[:16].sum() and bs[6] == q[96:112].sum() and bs[12] == q[192:208].sum()


def brute(wt, wraw, K, x):
    """mat-vec expectation from first principles: dequantised weights x activation rounded to the dot type."""
    w = gu.dequantize(wraw, wt, K).astype(np.float64)
    vt = {Q4_K: Q8_K, Q8_0: Q8_0, Q4_0: Q8_0}[wt]
    bs = {Q8_K: 292, Q8_0: 34}[vt] * (K // {Q8_K: 256, Q8_0: 32}[vt])
    xq = np.zeros(bs, np.uint8)
    O.oracle_quantize_row(vt, x.ctypes.data, xq.ctypes.data, K)
    if vt == Q8_K:
        blocks = xq.reshape(-1, 292)
        xd = np.concatenate([np.frombuffer(b[:4].tobytes(), np.float32)[0] * b[4:260].view(np.int8).astype(np.float64) for b in blocks])
    else:
        xd = gu.dequantize(xq.reshape(1, -1), Q8_0, K)[0].astype(np.float64)
    return w @ xd


def test_matvec_matches_first_principles():
    rng = np.random.default_rng(4)
    for wt, K, gen in ((Q4_K, 1024, gu.random_q4_K), (Q8_0, 256, gu.random_q8_0), (Q4_0, 512, gu.random_q4_0)):
        M = 12
        wraw = gen(rng, M, K)
        x = rng.standard_normal(K).astype(np.float32)
        y = np.zeros(M, np.float32)
        O.oracle_mul_mat_vec(wt, wraw.ctypes.data, wraw.shape[1], K, M, x.ctypes.data, y.ctypes.data)
        ref = brute(wt, wraw, K, x)
        assert np.allclose(y, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max()), (wt, np.abs(y - ref).max())


def test_product_host_quantisers_agree_with_oracle():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(512) * 3).astype(np.float32)
    for t, nbytes in ((Q8_0, 16 * 34), (Q4_0, 16 * 18), (Q8_K, 2 * 292), (F16, 1024), (BF16, 1024)):
        a, b = np.zeros(nbytes, np.uint8), np.zeros(nbytes, np.uint8)
        L.ggml_quantize_row(t, x.ctypes.data, a.ctypes.data, 512)
        O.oracle_quantize_row(t, x.ctypes.data, b.ctypes.data, 512)
        assert np.array_equal(a, b), t
    for t, gen in ((Q4_K, gu.random_q4_K), (Q8_0, gu.random_q8_0), (Q4_0, gu.random_q4_0)):
        raw = gen(rng, 1, 512)
        y = np.zeros(512, np.float32)
        L.ggml_dequantize_row(t, raw.ctypes.data, y.ctypes.data, 512)
        assert np.array_equal(y, gu.dequantize(raw, t, 512)[0]), t
    # the product's own Q4_K quantiser (load-time casts) round-trips within half a quantisation step
    q = np.zeros(2 * 144, np.uint8)
    L.ggml_quantize_row(Q4_K, x.ctypes.data, q.ctypes.data, 512)
    back = gu.dequantize(q.reshape(1, -1), Q4_K, 512)[0]
    assert np.abs(back - x).max() < 0.1 * np.abs(x).max()


def test_gelu_table_semantics():
    for v in (-11.0, -3.0, -0.5, 0.0, 0.7, 3.3, 11.0):
        h = np.float32(np.float16(v))
        ref = np.float32(np.float16(0.5 * h * (1 + np.tanh(np.float32(0.7978845608) * h * (1 + np.float32(0.044715) * h * h)))))
        exp = 0.0 if v <= -10 else (v if v >= 10 else ref)
        assert abs(O.oracle_gelu(v) - exp) <= 1e-3 * max(1.0, abs(exp))
