"""Generates tests/golden/ops.npz: inputs and expected outputs of PyTorch-CPU (fp32) restatements of the float
ops on the moshi.cpp hot path. Run in the build container only (`python tests/golden/make_golden.py`); the
fixture is data (inputs + expected outputs) and is what travels. It pins the CPU oracle, since the reference
ships no golden vectors of its own."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch.nn.functional as F

torch.manual_seed(0)
out = {}


def put(name, t):
    out[name] = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


# rms_norm (moshi RMSNorm: alpha * x / sqrt(mean(x^2) + eps)), layer_norm
x = torch.randn(3, 512) * 2 + 0.3
alpha = torch.rand(512) + 0.5
put("rms_x", x); put("rms_alpha", alpha)
put("rms_y", alpha * (x * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True).float() + 1e-8)))
w, b = torch.rand(512) + 0.5, torch.randn(512) * 0.1
put("ln_w", w); put("ln_b", b)
put("ln_y", F.layer_norm(x, (512,), w, b, 1e-5))

# activations
a = torch.randn(4, 300) * 3
put("act_x", a)
put("silu_y", F.silu(a)); put("elu_y", F.elu(a)); put("gelu_y", F.gelu(a, approximate="tanh"))

# masked softmax with scale
s = torch.randn(2, 3, 50)
mask = torch.zeros(3, 50); mask[0, 20:] = -float("inf"); mask[1, 35:] = -float("inf")
put("sm_x", s); put("sm_mask", mask)
put("sm_y", torch.softmax(s * 0.125 + mask, -1))

# conv1d with f16-rounded weights and inputs (ggml's im2col is F16), conv_transpose1d f32
cw = (torch.randn(24, 8, 5) / (40 ** 0.5)).half().float()
cx = torch.randn(1, 8, 33)
put("conv_w", cw); put("conv_x", cx[0])
put("conv_y_s1", F.conv1d(cx.half().float(), cw)[0]); put("conv_y_s2", F.conv1d(cx.half().float(), cw, stride=2)[0])
tw = torch.randn(8, 6, 4) / (8 ** 0.5)
tx = torch.randn(1, 8, 7)
put("convtr_w", tw); put("convtr_x", tx[0])
put("convtr_y", F.conv_transpose1d(tx, tw, stride=2)[0])

# RoPE as moshi applies it: interleaved (re, im) pairs, output de-interleaved [re | im] (rope.h:33-128)
D, T, H = 16, 3, 2
q = torch.randn(H, T, D)
offset = 5
j = torch.arange(D // 2, dtype=torch.float32)
freqs = torch.exp(-np.log(10000.0) * j / (D // 2))
ts = torch.arange(T, dtype=torch.float32) + offset
ang = ts[:, None] * freqs[None, :]
qr, qi = q[..., 0::2], q[..., 1::2]
put("rope_q", q); put("rope_offset", np.float32(offset))
put("rope_y", torch.cat([qr * torch.cos(ang) - qi * torch.sin(ang), qr * torch.sin(ang) + qi * torch.cos(ang)], -1))

# attention over a cache with additive mask (torch.h:225-237)
Dh, C, Hh, Tq = 8, 10, 2, 2
k, v, qq = torch.randn(Hh, C, Dh), torch.randn(Hh, C, Dh), torch.randn(Hh, Tq, Dh)
am = torch.zeros(Tq, C); am[0, 6:] = -float("inf"); am[1, 7:] = -float("inf")
put("att_k", k); put("att_v", v); put("att_q", qq); put("att_mask", am)
put("att_y", F.scaled_dot_product_attention(qq, k, v, attn_mask=am))

# nearest-centroid search of the RVQ encoder (core_vq.h:27-56)
cb = torch.randn(40, 6); xv = torch.randn(5, 6)
put("vq_cb", cb); put("vq_x", xv)
put("vq_idx", torch.cdist(xv, cb).argmin(-1).int())

# timestep embedding [cos | sin]
tt = torch.tensor([0.0, 1.0, 17.0, 250.0])
fr = torch.exp(-np.log(10000.0) * torch.arange(32, dtype=torch.float32) / 32)
put("ts_t", tt)
put("ts_y", torch.cat([torch.cos(tt[:, None] * fr), torch.sin(tt[:, None] * fr)], -1))

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ops.npz"), **out)
print("wrote ops.npz with", len(out), "arrays")


# ---------------------------------------------------------------------------------------------------------------------------
# quant.npz: block-quantised mat-vec, written from the published block layouts alone (no oracle code is called):
#   block_q4_K {f16 d; f16 dmin; u8 scales[12]; u8 qs[128]}: 8 sub-blocks of 32, w = d*sc*nib - dmin*m, each 32-byte qs group holds
#     sub-block 2j in its low nibbles and 2j+1 in its high nibbles; (sc, m) are 6-bit: j < 4: sc = s[j] & 63, m = s[j+4] & 63;
#     j >= 4: sc = (s[j+4] & 15) | (s[j-4] >> 6) << 4, m = (s[j+4] >> 4) | (s[j] >> 6) << 4
#   block_q8_0 {f16 d; i8 q[32]}; block_q4_0 {f16 d; u8 q[16]}: w = (nib - 8) d, low nibbles = elements 0-15, high = 16-31
#   mul_mat quantises the activation row first: Q8_K (per 256: iscale = -127 / (the signed value of largest magnitude),
#     q = min(127, rne(iscale x)), d = 1 / iscale) for Q4_K weights; Q8_0 (per 32: d = amax / 127, q = round-half-away(x / d), d kept as
#     f16) for Q8_0 / Q4_0 weights; the dot is integer per (sub-)block, then scaled in float.
from block_arith import *  # noqa: E402,F401,F403  (f16, q4k_fields, dequant_q4k, quant_q8k, matvec_q4k, quant_q80, matvec_q80, dequant_q40, matvec_q40)


def rand_blocks(rng, rows, k, kind):
    if kind == "q4_K":
        nb = k // 256
        o = rng.integers(0, 256, size=(rows, nb, 144), dtype=np.uint8)
        o[:, :, 0:2] = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -6).astype(np.float16).view(np.uint8).reshape(rows, nb, 2)
        o[:, :, 2:4] = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -5).astype(np.float16).view(np.uint8).reshape(rows, nb, 2)
        return o.reshape(rows, -1)
    bs = 34 if kind == "q8_0" else 18
    nb = k // 32
    o = rng.integers(0, 256, size=(rows, nb, bs), dtype=np.uint8)
    o[:, :, 0:2] = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -6).astype(np.float16).view(np.uint8).reshape(rows, nb, 2)
    if kind == "q8_0":
        body = o[:, :, 2:].view(np.int8)
        body[body == -128] = -127
    return o.reshape(rows, -1)


qout = {}
rng = np.random.default_rng(7)
K, M = 1024, 24
xq = (rng.standard_normal(K) * np.exp(rng.standard_normal(K) * 0.7)).astype(np.float32)
xq[256:300] = 0.0
qout["x"] = xq
for kind, mv, dq in (("q4_K", matvec_q4k, dequant_q4k), ("q8_0", matvec_q80, None), ("q4_0", matvec_q40, dequant_q40)):
    wrows = rand_blocks(rng, M, K, kind)
    qout[kind + "_w"] = wrows
    qout[kind + "_y"] = mv(wrows, xq)
    if dq is not None:
        qout[kind + "_deq"] = dq(wrows)
q8, d8 = quant_q8k(xq)
qout["x_q8k_q"], qout["x_q8k_d"] = q8.astype(np.int8), d8
q0, d0 = quant_q80(xq)
qout["x_q80_q"], qout["x_q80_d"] = q0.astype(np.int8), d0
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "quant.npz"), **qout)
print("wrote quant.npz with", len(qout), "arrays")
