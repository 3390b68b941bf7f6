"""Block-quantised arithmetic written from the published block layouts alone (no oracle code is called). Pure numpy, no side effects:
imported by make_golden.py (quant.npz) and make_module_golden.py (the quantised Temporal / Depth stack fixture).
  block_q4_K {f16 d; f16 dmin; u8 scales[12]; u8 qs[128]}: 8 sub-blocks of 32, w = d*sc*nib - dmin*m, each 32-byte qs group holds
    sub-block 2j in its low nibbles and 2j+1 in its high nibbles; (sc, m) are 6-bit: j < 4: sc = s[j] & 63, m = s[j+4] & 63;
    j >= 4: sc = (s[j+4] & 15) | (s[j-4] >> 6) << 4, m = (s[j+4] >> 4) | (s[j] >> 6) << 4
  block_q8_0 {f16 d; i8 q[32]}; block_q4_0 {f16 d; u8 q[16]}: w = (nib - 8) d, low nibbles = elements 0-15, high = 16-31
  mul_mat quantises the activation row first: Q8_K (per 256: iscale = -127 / (the signed value of largest magnitude),
    q = min(127, rne(iscale x)), d = 1 / iscale) for Q4_K weights; Q8_0 (per 32: d = amax / 127, q = round-half-away(x / d), d kept as
    f16) for Q8_0 / Q4_0 weights; the dot is integer per (sub-)block, then scaled in float."""
import numpy as np


def f16(raw2):
    return np.ascontiguousarray(raw2).view(np.float16).astype(np.float32)


def q4k_fields(rows):
    r = rows.reshape(rows.shape[0], -1, 144)
    d, dmin = f16(r[:, :, 0:2])[..., 0], f16(r[:, :, 2:4])[..., 0]
    s = r[:, :, 4:16].astype(np.int32)
    sc = np.zeros(r.shape[:2] + (8,), np.int32); mn = np.zeros_like(sc)
    for j in range(4):
        sc[..., j] = s[..., j] & 63; mn[..., j] = s[..., j + 4] & 63
        sc[..., j + 4] = (s[..., j + 8] & 15) | ((s[..., j] >> 6) << 4)
        mn[..., j + 4] = (s[..., j + 8] >> 4) | ((s[..., j + 4] >> 6) << 4)
    qs = r[:, :, 16:144].reshape(r.shape[0], r.shape[1], 4, 32).astype(np.int32)
    nib = np.stack([qs & 15, qs >> 4], axis=3).reshape(r.shape[0], r.shape[1], 8, 32)   # sub-block 2j = low, 2j+1 = high
    return d, dmin, sc, mn, nib


def dequant_q4k(rows):
    d, dmin, sc, mn, nib = q4k_fields(rows)
    w = (d[..., None] * sc.astype(np.float32))[..., None] * nib.astype(np.float32) - (dmin[..., None] * mn.astype(np.float32))[..., None]
    return w.reshape(rows.shape[0], -1).astype(np.float32)


def quant_q8k(x):
    xb = x.reshape(-1, 256)
    q = np.zeros(xb.shape, np.int32); d = np.zeros(xb.shape[0], np.float32)
    for b in range(xb.shape[0]):
        j = int(np.argmax(np.abs(xb[b])))
        if xb[b, j] == 0:
            continue
        iscale = np.float32(-127.0) / xb[b, j]
        q[b] = np.minimum(127, np.rint((iscale * xb[b]).astype(np.float32)).astype(np.int32))
        d[b] = np.float32(1.0) / iscale
    return q, d


def matvec_q4k(rows, x):
    d, dmin, sc, mn, nib = q4k_fields(rows)
    q8, d8 = quant_q8k(x)
    q8s = q8.reshape(-1, 8, 32)
    isum = np.einsum("rbjk,bjk->rbj", nib.astype(np.int64), q8s.astype(np.int64))                # integer sub-block dots
    bsum = q8s.sum(-1)                                                                             # [blocks, 8]
    a = (isum * sc).sum(-1).astype(np.float32); m = (mn * bsum[None]).sum(-1).astype(np.float32)
    per_block = (d * d8[None]) * a - (dmin * d8[None]) * m
    return per_block.astype(np.float64).sum(-1).astype(np.float32)


def quant_q80(x):
    xb = x.reshape(-1, 32)
    amax = np.abs(xb).max(1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
    p = (xb * idv[:, None]).astype(np.float32)
    q = (np.sign(p) * np.floor(np.abs(p) + np.float32(0.5))).astype(np.int32)
    return q, d.astype(np.float16).astype(np.float32)


def matvec_q80(rows, x):
    r = rows.reshape(rows.shape[0], -1, 34)
    dw = f16(r[:, :, 0:2])[..., 0]
    qw = r[:, :, 2:34].view(np.int8).astype(np.int64)
    q8, d8 = quant_q80(x)
    isum = np.einsum("rbk,bk->rb", qw, q8.astype(np.int64)).astype(np.float32)
    return (isum * (dw * d8[None])).astype(np.float64).sum(-1).astype(np.float32)


def dequant_q40(rows):
    r = rows.reshape(rows.shape[0], -1, 18)
    dw = f16(r[:, :, 0:2])[..., 0]
    qs = r[:, :, 2:18].astype(np.int32)
    nib = np.concatenate([qs & 15, qs >> 4], axis=-1) - 8
    return (dw[..., None] * nib.astype(np.float32)).reshape(rows.shape[0], -1).astype(np.float32)


def matvec_q40(rows, x):
    r = rows.reshape(rows.shape[0], -1, 18)
    dw = f16(r[:, :, 0:2])[..., 0]
    qs = r[:, :, 2:18].astype(np.int64)
    nib = np.concatenate([qs & 15, qs >> 4], axis=-1) - 8
    q8, d8 = quant_q80(x)
    isum = np.einsum("rbk,bk->rb", nib, q8.astype(np.int64)).astype(np.float32)
    return ((isum * dw) * d8[None]).astype(np.float64).sum(-1).astype(np.float32)


