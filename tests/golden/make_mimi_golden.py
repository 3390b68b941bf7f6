"""Generates tests/golden/mimi_encoder.npz: an OFFLINE PyTorch restatement of the Mimi encoder (SEANet conv stack -> 8-layer transformer -> stride-2
downsampling conv -> split residual VQ, Defossez et al. 2024 as configured in lm_default.h:416-578) applied to the whole signal at once with causal
padding, against the frame driver's STREAMING encoder on the CPU oracle, frame by frame, on the same synthetic weights. Stored: the audio, the codes the
PyTorch model assigns and the latent the first RVQ stack quantises; tests/test_oracle_golden.py replays the audio through the driver and compares.
Pins: streaming convolutions with carried tails == causal convolutions of the whole signal; ELU / GELU (ggml's F16 table) / LayerNorm / layer scale;
attention over the BF16 ring with RoPE == causal attention; conv_1d = im2col(F16) x F16 weights; nearest-centroid search and residual update.
Run in the build container only: `python tests/golden/make_mimi_golden.py`."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import hot_util as hu  # noqa: E402

L = hu.L
FRAMES = 5


def make_cfg(full=False):
    cfg = hu.hot.tiny(L)
    cfg.enable_lm = cfg.enable_mimi_decoder = 0
    if full:                       # the real quantiser: 8 levels of 2048 centroids (moshika / the benchmark); the default fixture keeps the test model's 3 x 64
        cfg.mimi_n_q, cfg.mimi_codebook_size = 8, 2048
    return cfg


def tensor(m, name):
    t = C.cast(L.moshi_hot_weight(m.m, name.encode()), hu.pkg.TP)
    assert t, name
    tt = t.contents
    shape = tuple(int(tt.ne[i]) for i in (3, 2, 1, 0))
    n = int(np.prod(shape))
    if tt.type == hu.pkg.F16:
        a = np.zeros(n, np.float16)
    else:
        assert tt.type == hu.pkg.F32, (name, tt.type)
        a = np.zeros(n, np.float32)
    L.ggml_backend_tensor_get(t, a.ctypes.data, 0, a.nbytes)
    a = a.astype(np.float32).reshape(shape)                                        # ggml ne (k, cin, cout, 1) -> [1, cout, cin, k]
    keep = 3 if (".conv" in name or "_proj.weight" in name) and name.endswith("weight") else 1
    while a.ndim > keep and a.shape[0] == 1:
        a = a[0]
    return torch.from_numpy(a.copy())


def f16(x):
    return x.to(torch.float16).to(torch.float32)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def causal_conv(x, w, b, stride):
    """x [Cin, T] -> [Cout, T / stride]; ggml: im2col in F16, F16 weights, exact products summed in double"""
    k = w.shape[-1]
    xp = F.pad(f16(x), (k - stride, 0)).to(torch.float64)[None]
    y = F.conv1d(xp, w.to(torch.float64), None, stride=stride)[0].to(torch.float32)
    return y + b[:, None] if b is not None else y


def elu(x):
    return torch.where(x > 0, x, torch.expm1(x))


def gelu_f16_table(x):     # ggml_vec_gelu_f32 [ggml-upstream]: the argument and the result pass through F16
    xh = f16(x)
    y = 0.5 * xh * (1.0 + torch.tanh(0.79788456080286535587989211986876 * xh * (1.0 + 0.044715 * xh * xh)))
    return f16(y)


def lin(W, x):             # [M, K] x [K, T] -> [M, T]: float products summed in double (ggml_vec_dot_f32)
    return (W[:, :, None] * x[None, :, :]).to(torch.float64).sum(1).to(torch.float32)


def layer_norm(x, w, b, eps):     # over channels, per column (ggml_norm: mean and variance in double)
    mu = x.to(torch.float64).mean(0, keepdim=True)
    v = (x - mu.to(torch.float32))
    var = (v * v).to(torch.float64).mean(0, keepdim=True)
    return (v * (1.0 / torch.sqrt(var.to(torch.float32) + eps))) * w[:, None] + b[:, None]


def transformer(m, prefix, x, H=8, context=250, max_period=10000.0):
    """x [D, T]: causal attention over the last `context` positions, K / V kept in BF16"""
    D, T = x.shape
    Dh = D // H
    theta = torch.exp(-np.log(max_period) * torch.arange(Dh // 2, dtype=torch.float32) / (Dh // 2))
    ang = torch.arange(T, dtype=torch.float32)[:, None] * theta[None, :]
    cos, sin = torch.cos(ang), torch.sin(ang)                                        # [T, Dh/2]

    def rope(t):                                                                     # t [H, T, Dh], interleaved pairs -> [re | im]
        re, im = t[..., 0::2], t[..., 1::2]
        return torch.cat([re * cos - im * sin, re * sin + im * cos], -1)
    for l in range(8):
        p = f"{prefix}.layers.{l}."
        g = lambda n: tensor(m, p + n)
        h = layer_norm(x, g("norm1.weight"), g("norm1.bias"), 1e-5)
        qkv = lin(g("self_attn.in_projs.weight"), h)                                 # [3D, T]
        q, k, v = (qkv[i * D:(i + 1) * D].view(H, Dh, T).permute(0, 2, 1) for i in range(3))
        q, k = bf16(rope(q)), bf16(rope(k))
        v = bf16(v)
        s = (q[:, :, None, :] * k[:, None, :, :]).to(torch.float64).sum(-1).to(torch.float32) * torch.tensor(1.0 / np.sqrt(Dh), dtype=torch.float32)   # [H, Tq, Tk]
        tq, tk = torch.arange(T)[:, None], torch.arange(T)[None, :]
        s = s.masked_fill(~((tk <= tq) & (tk > tq - context))[None], float("-inf"))
        e = torch.exp(s - s.max(-1, keepdim=True).values)
        pr = bf16(e * (1.0 / e.to(torch.float64).sum(-1, keepdim=True)).to(torch.float32))
        o = (pr[:, :, :, None] * v[:, None, :, :]).to(torch.float64).sum(2).to(torch.float32)     # [H, Tq, Dh]
        o = o.permute(0, 2, 1).reshape(D, T)
        x = x + lin(g("self_attn.out_projs.weight"), o) * g("layer_scale_1.scale")[:, None]
        h = layer_norm(x, g("norm2.weight"), g("norm2.bias"), 1e-5)
        x = x + lin(g("linear2.weight"), gelu_f16_table(lin(g("linear1.weight"), h))) * g("layer_scale_2.scale")[:, None]
    return x


def rvq_encode(m, stack, n_levels, x):
    """x [512, T] -> codes [n_levels, T], latent [256, T]"""
    lat = causal_conv(x, tensor(m, f"mimi.quantizer.{stack}.input_proj.weight"), None, 1)
    res, codes = lat.clone(), []
    for i in range(n_levels):
        cb = tensor(m, f"mimi.quantizer.{stack}.vq.layers.{i}._codebook.embedding")      # [2048, 256]
        d = cb[:, :, None] - res[None, :, :]
        d = (d * d).to(torch.float64).sum(1).to(torch.float32)                              # [2048, T]: float squares summed in double
        score = 1.0 / (d + 1.0)
        idx = (score.shape[0] - 1) - torch.flip(score, [0]).argmax(0)                       # ggml_vec_argmax_f32: the LAST maximum
        codes.append(idx)
        res = res - cb[idx].T
    return torch.stack(codes), lat


def causal_conv_transpose(x, w, b, stride, groups=1):
    """x [Cin, T] -> [Cout, T * stride]: the full transposed convolution with its last k - stride samples (the part a later input would still add to) cut off"""
    k = w.shape[-1]
    y = F.conv_transpose1d(x.to(torch.float64)[None], w.to(torch.float64), None, stride=stride, groups=groups)[0].to(torch.float32)
    y = y[:, :x.shape[1] * stride]
    return y + b[:, None] if b is not None else y


def decoder_main():
    """tests/golden/mimi_decoder.npz: codes -> RVQ decode (sum of centroids, 1x1 output projections) -> depthwise stride-2 upsampling -> 8 transformer layers ->
    SEANet decoder (transposed convolutions with overlap-add, residual blocks) -> 1920 samples per frame; PyTorch offline vs the driver streaming"""
    cfg = hu.hot.tiny(L)
    cfg.enable_lm = cfg.enable_mimi_encoder = 0
    m = hu.Model("oracle", cfg, seed=0)
    rng = np.random.default_rng(9)
    codes = rng.integers(0, cfg.mimi_codebook_size, (FRAMES, cfg.mimi_n_q)).astype(np.int32)
    got = np.concatenate([m.mimi_decode(c.tolist()) for c in codes])
    w = lambda n: tensor(m, n)

    def rvq_decode(stack, cs):                     # cs [levels, T]
        q = None
        for i in range(cs.shape[0]):
            e = w(f"mimi.quantizer.{stack}.vq.layers.{i}._codebook.embedding")[torch.from_numpy(cs[i].astype(np.int64))].T      # [256, T]
            q = e if q is None else q + e
        return causal_conv(q, w(f"mimi.quantizer.{stack}.output_proj.weight"), None, 1)
    ct = codes.T
    x = rvq_decode("rvq_first", ct[:1]) + rvq_decode("rvq_rest", ct[1:])                                                    # [512, FRAMES]
    x = causal_conv_transpose(x, w("mimi.upsample.convtr.weight"), None, 2, groups=512)                                     # [512, 2 FRAMES]
    x = transformer(m, "mimi.decoder_transformer.transformer", x)
    x = elu(causal_conv(x, w("mimi.decoder.model.0.conv.weight"), w("mimi.decoder.model.0.conv.bias").reshape(-1), 1))
    for i, st in enumerate((8, 6, 5, 4)):
        p = f"mimi.decoder.model.{2 + 3 * i}.convtr."
        x = causal_conv_transpose(x, w(p + "weight"), w(p + "bias").reshape(-1), st)
        p = f"mimi.decoder.model.{3 + 3 * i}.block."
        v = causal_conv(elu(x), w(p + "1.conv.weight"), w(p + "1.conv.bias").reshape(-1), 1)
        v = causal_conv(elu(v), w(p + "3.conv.weight"), w(p + "3.conv.bias").reshape(-1), 1)
        x = elu(x + v)
    x = causal_conv(x, w("mimi.decoder.model.14.conv.weight"), w("mimi.decoder.model.14.conv.bias").reshape(-1), 1)        # [1, 1920 FRAMES]
    m.free()
    pcm = x[0].numpy()
    for i in range(FRAMES):
        print(f"decoder frame {i}: pcm rel err {hu.rel_err(pcm[i * 1920:(i + 1) * 1920], got[i * 1920:(i + 1) * 1920]):.2e}")
    np.savez_compressed(os.path.join(HERE, "mimi_decoder.npz"), codes=codes, pcm=pcm)
    print("wrote mimi_decoder.npz")


def main(full=False):
    cfg = make_cfg(full)
    m = hu.Model("oracle", cfg, seed=0)
    rng = np.random.default_rng(8)
    pcm = (rng.standard_normal(FRAMES * 1920) * 0.2).astype(np.float32)
    got_codes = [m.mimi_encode(pcm[i * 1920:(i + 1) * 1920]) for i in range(FRAMES)]
    got_lat = m.read("enc_latent_first", 256)
    w = lambda n: tensor(m, n)
    x = torch.from_numpy(pcm)[None, :]
    x = causal_conv(x, w("mimi.encoder.model.0.conv.weight"), w("mimi.encoder.model.0.conv.bias").reshape(-1), 1)
    for i, st in enumerate((4, 5, 6, 8)):
        p = f"mimi.encoder.model.{1 + 3 * i}.block."
        v = causal_conv(elu(x), w(p + "1.conv.weight"), w(p + "1.conv.bias").reshape(-1), 1)
        v = causal_conv(elu(v), w(p + "3.conv.weight"), w(p + "3.conv.bias").reshape(-1), 1)
        x = x + v
        p = f"mimi.encoder.model.{3 + 3 * i}.conv."
        x = causal_conv(elu(x), w(p + "weight"), w(p + "bias").reshape(-1), st)
    x = causal_conv(elu(x), w("mimi.encoder.model.14.conv.weight"), w("mimi.encoder.model.14.conv.bias").reshape(-1), 1)     # [512, 2 * FRAMES]
    x = transformer(m, "mimi.encoder_transformer.transformer", x)
    x = causal_conv(x, w("mimi.downsample.conv.weight"), None, 2)                                                          # [512, FRAMES]
    c1, lat = rvq_encode(m, "rvq_first", 1, x)
    c2, _ = rvq_encode(m, "rvq_rest", cfg.mimi_n_q - 1, x)
    codes = torch.cat([c1, c2]).T.numpy().astype(np.int32)                                                                  # [FRAMES, n_q]
    m.free()
    print("driver (streaming, oracle):", got_codes)
    print("pytorch (offline)         :", codes.tolist())
    print("latent of the last frame rel err:", hu.rel_err(lat[:, -1].numpy(), got_lat))
    print("codes equal:", int((np.array(got_codes) == codes).sum()), "of", codes.size)
    assert full or np.array_equal(np.array(got_codes), codes)
    name = "mimi_encoder_full.npz" if full else "mimi_encoder.npz"
    np.savez_compressed(os.path.join(HERE, name), pcm=pcm, codes=codes, latent_first=lat.numpy())
    print("wrote", name)


if __name__ == "__main__":
    main()
    main(full=True)
    decoder_main()
