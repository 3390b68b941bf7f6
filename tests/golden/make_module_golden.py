"""Generates tests/golden/temporal_stack.npz: a PyTorch restatement of Moshi's streaming Temporal transformer (RMSNorm -> multi-head attention with
interleaved RoPE over a BF16 ring cache -> gated SiLU feed-forward, residual stream; out_norm -> text_linear) stepped for several frames, on the
synthetic F32 weights of a small model. Inputs (the stack's input vector of every step, taken from the driver) and the PyTorch outputs are stored;
tests/test_oracle_golden.py runs the SAME steps through the frame driver on the CPU oracle and compares. This pins the oracle's op semantics AND the
driver's graph construction (moshi_hot.cpp, restating transformer.h / rope.h / gating.h / torch.h) against an independent implementation of the
architecture at module level. Run in the build container only: `python tests/golden/make_module_golden.py`.

Roundings restated from ggml's CPU backend [ggml-upstream]: K / V rows are stored in BF16 (round to nearest even); a mat-mul whose first operand is BF16
converts the other one to BF16, multiplies in float and sums in double; F32 x F32 products are summed in double; soft_max: expf(x - max) in float, sum in
double, scaled by float(1 / sum)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import hot_util as hu  # noqa: E402

L = hu.L
F32 = hu.pkg.F32
STEPS = 9


sys.path.insert(0, HERE)
import block_arith as ba  # noqa: E402

LINEAR = F32                   # set by main(): F32, or Q4_K for the quantised fixture (every linear a Q4_K matrix, activations rounded to Q8_K)


MODEL = "tiny"                 # or "personaplex": 17 codebooks, 16 chained Depth steps over a ring of 8 that wraps inside every frame


def make_cfg():
    if MODEL == "personaplex":
        cfg = hu.hot.tiny_personaplex(L, linear_type=LINEAR, embed_type=F32, layers=2)
        cfg.context = 6
        cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
        return cfg
    cfg = hu.hot.tiny(L, linear_type=LINEAR, embed_type=F32, layers=2, context=6)     # ring of 6: steps 6.. wrap it
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def weight(m, name, rows, cols):
    t = C.cast(L.moshi_hot_weight(m.m, name.encode()), hu.pkg.TP)
    assert t, name
    if t.contents.type == hu.pkg.Q4_K:          # raw super-blocks: the products go through tests/golden/block_arith.py (numpy, from the block layout alone)
        raw = np.zeros((rows, cols // 256 * 144), np.uint8)
        L.ggml_backend_tensor_get(t, raw.ctypes.data, 0, raw.nbytes)
        return raw
    assert t.contents.type == F32, (name, t.contents.type)
    w = np.zeros((rows, cols), np.float32)
    L.ggml_backend_tensor_get(t, w.ctypes.data, 0, w.nbytes)
    return torch.from_numpy(w)


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def dot_f32(W, x):          # rows of W against x: float products, double sums (ggml_vec_dot_f32); Q4_K rows: x -> Q8_K, integer sub-block dots
    if isinstance(W, np.ndarray):
        return torch.from_numpy(ba.matvec_q4k(W, x.numpy().astype(np.float32)))
    return (W * x[None, :]).to(torch.float64).sum(-1).to(torch.float32)


def rms_norm(x, alpha, eps):
    ms = (x * x).to(torch.float64).sum() / x.numel()          # float squares summed in double
    return alpha * (x * (1.0 / torch.sqrt(ms.to(torch.float32) + eps)))


def main(linear=F32, out_name="temporal_stack.npz", model="tiny"):
    global LINEAR, MODEL
    LINEAR, MODEL = linear, model
    cfg = make_cfg()
    m = hu.Model("oracle", cfg, seed=0)
    D, H, Cap, F = cfg.dim, cfg.num_heads, cfg.context, cfg.ffn_hidden
    Dh = D // H
    W = []
    for l in range(cfg.num_layers):
        p = f"lm.transformer.layers.{l}."
        W.append(dict(n1=weight(m, p + "norm1.alpha", 1, D)[0], n2=weight(m, p + "norm2.alpha", 1, D)[0],
                      inp=weight(m, p + "self_attn.in_projs.weight", 3 * D, D), out=weight(m, p + "self_attn.out_projs.weight", D, D),
                      gin=weight(m, p + "gating.linear_in.weight", 2 * F, D), gout=weight(m, p + "gating.linear_out.weight", D, F)))
    out_norm = weight(m, "lm.out_norm.alpha", 1, D)[0]
    text_linear = weight(m, "lm.text_linear.weight", cfg.text_card, D)
    # the Depth transformer: dep_q chained steps per frame, step k with its own weight set k, attending to the rows steps 0 .. k-1 of the same frame
    # left in a ring of dep_q slots (no RoPE); input of step k = depformer_in[k] . transformer_out + embedding of the previous step's token (lm.h:446-553)
    DD, DH_, DF, Q = cfg.dep_dim, cfg.dep_heads, cfg.dep_ffn_hidden, cfg.dep_q
    DW = []
    for l in range(cfg.dep_layers):
        p = f"lm.depformer.layers.{l}."
        DW.append(dict(n1=weight(m, p + "norm1.alpha", 1, DD)[0], n2=weight(m, p + "norm2.alpha", 1, DD)[0],
                       inp=[weight(m, p + f"self_attn.in_projs.{k}.weight", 3 * DD, DD) for k in range(Q)],
                       out=[weight(m, p + f"self_attn.out_projs.{k}.weight", DD, DD) for k in range(Q)],
                       gin=[weight(m, p + f"gating.{k}.linear_in.weight", 2 * DF, DD) for k in range(Q)],
                       gout=[weight(m, p + f"gating.{k}.linear_out.weight", DD, DF) for k in range(Q)]))
    dep_in = [weight(m, f"lm.depformer_in.{k}.weight", DD, D) for k in range(Q)]
    dep_heads_w = [weight(m, f"lm.linears.{k}.weight", cfg.card, DD) for k in range(Q)]
    dep_text_emb = weight(m, "lm.depformer_text_emb.weight", cfg.text_card + 1, DD)
    dep_emb = [weight(m, f"lm.depformer_emb.{k}.weight", cfg.card + 1, DD) for k in range(Q - 1)]
    dep_logits, dep_tokens, text_tokens = [], [], []
    kc = [torch.zeros(H, Cap, Dh) for _ in W]
    vc = [torch.zeros(H, Cap, Dh) for _ in W]
    theta = torch.exp(-np.log(float(cfg.max_period)) * torch.arange(Dh // 2, dtype=torch.float32) / (Dh // 2))
    rng = np.random.default_rng(4)
    xs, outs, logits, toks = [], [], [], []
    for step in range(STEPS):
        tokens = [int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist()     # a "provided" frame: every codebook given
        m.lm_step_n(tokens)
        x = torch.from_numpy(m.read("transformer_in", D).copy())
        xs.append(x.numpy().copy()); toks.append(tokens)
        slot = step % Cap
        live = min(step + 1, Cap)                 # slots 0 .. live-1 hold rows (after the wrap: all of them)
        ang = torch.tensor(float(step)) * theta
        cos, sin = torch.cos(ang), torch.sin(ang)
        for l, w in enumerate(W):
            h = rms_norm(x, w["n1"], 1e-8)
            qkv = dot_f32(w["inp"], h)
            q, k, v = qkv[:D].view(H, Dh), qkv[D:2 * D].view(H, Dh), qkv[2 * D:].view(H, Dh)

            def rope(t):                          # interleaved (re, im) pairs -> [re | im] (rope.h:33-128)
                re, im = t[:, 0::2], t[:, 1::2]
                return torch.cat([re * cos - im * sin, re * sin + im * cos], -1)
            q, k = rope(q), rope(k)
            kc[l][:, slot] = bf16(k); vc[l][:, slot] = bf16(v)
            qb = bf16(q)
            s = (kc[l][:, :live] * qb[:, None, :]).to(torch.float64).sum(-1).to(torch.float32) * torch.tensor(1.0 / np.sqrt(Dh), dtype=torch.float32)
            e = torch.exp(s - s.max(-1, keepdim=True).values)
            pr = e * (1.0 / e.to(torch.float64).sum(-1, keepdim=True)).to(torch.float32)
            pb = bf16(pr)
            o = (vc[l][:, :live] * pb[:, :, None]).to(torch.float64).sum(1).to(torch.float32)          # [H, Dh]
            x = x + dot_f32(w["out"], o.reshape(D))
            h = rms_norm(x, w["n2"], 1e-8)
            g = dot_f32(w["gin"], h)
            left, right = g[:F], g[F:]
            x = x + dot_f32(w["gout"], (left / (1.0 + torch.exp(-left))) * right)
        y = rms_norm(x, out_norm, 1e-8)
        outs.append(y.numpy().copy())
        logits.append(dot_f32(text_linear, y).numpy().copy())
        got_out, got_logits = m.read("transformer_out", D), m.read("text_logits", cfg.text_card)
        # ---- Depth chain of this frame, fed the DRIVER's transformer_out and sampled text token (stored in the fixture as inputs) ----
        t_out = torch.from_numpy(got_out.copy())
        text_tok, drv_audio = m.last_raw()
        text_tokens.append(text_tok)
        DDh = DD // DH_
        Cd = cfg.dep_context if cfg.dep_context > 0 else Q          # the Depth ring: dep_q slots, or fewer - then it wraps inside the frame (PersonaPlex: 16 steps over 8)
        dkc = [torch.zeros(DH_, Cd, DDh) for _ in DW]
        dvc = [torch.zeros(DH_, Cd, DDh) for _ in DW]
        prev, frame_logits, frame_toks = None, [], []
        for k in range(Q):
            emb = dep_text_emb[text_tok] if k == 0 else dep_emb[k - 1][prev]
            xd = dot_f32(dep_in[k], t_out) + emb
            for l, w in enumerate(DW):
                h = rms_norm(xd, w["n1"], 1e-8)
                qkv = dot_f32(w["inp"][k], h)
                q, kk, v = qkv[:DD].view(DH_, DDh), qkv[DD:2 * DD].view(DH_, DDh), qkv[2 * DD:].view(DH_, DDh)
                dkc[l][:, k % Cd] = bf16(kk); dvc[l][:, k % Cd] = bf16(v)
                qb = bf16(q)
                nl = min(k + 1, Cd)                                     # slots 0 .. k while the ring fills, all of it afterwards (torch.h:205-223 for T = 1)
                sc_ = (dkc[l][:, :nl] * qb[:, None, :]).to(torch.float64).sum(-1).to(torch.float32) * torch.tensor(1.0 / np.sqrt(DDh), dtype=torch.float32)
                e = torch.exp(sc_ - sc_.max(-1, keepdim=True).values)
                pb = bf16(e * (1.0 / e.to(torch.float64).sum(-1, keepdim=True)).to(torch.float32))
                o = (dvc[l][:, :nl] * pb[:, :, None]).to(torch.float64).sum(1).to(torch.float32)
                xd = xd + dot_f32(w["out"][k], o.reshape(DD))
                h = rms_norm(xd, w["n2"], 1e-8)
                g_ = dot_f32(w["gin"][k], h)
                left, right = g_[:DF], g_[DF:]
                xd = xd + dot_f32(w["gout"][k], (left / (1.0 + torch.exp(-left))) * right)
            lg = dot_f32(dep_heads_w[k], xd)
            frame_logits.append(lg.numpy().copy())
            mx = lg.max()
            prev = int(torch.nonzero(lg == mx)[-1])          # ggml_vec_argmax_f32: the last maximum
            frame_toks.append(prev)
            prev = drv_audio[k]                              # (teacher-forced: the next step embeds the DRIVER's token, stored in the fixture)
        dep_logits.append(np.array(frame_logits)); dep_tokens.append(frame_toks)
        de = max(hu.rel_err(frame_logits[k], m.read(f"dep_logits{k}", cfg.card)) for k in range(Q))
        print(f"         depth chain: worst logits rel err {de:.2e}; tokens pytorch {frame_toks} driver {drv_audio}")
        assert frame_toks == drv_audio or linear != F32, "F32: the tokens must be identical"
        print(f"step {step}: transformer_out rel err {hu.rel_err(outs[-1], got_out):.2e}, text logits {hu.rel_err(logits[-1], got_logits):.2e}")
    m.free()
    np.savez_compressed(os.path.join(HERE, out_name), tokens=np.array(toks, np.int32), x_in=np.array(xs), transformer_out=np.array(outs),
                        text_logits=np.array(logits), text_tokens=np.array(text_tokens, np.int32),
                        dep_logits=np.array(dep_logits)[:, :, ::max(1, cfg.card // 64)],      # (every 32nd logit of a 2048-entry head: the fixture stays small)
                        dep_tokens=np.array(dep_tokens, np.int32))
    print("wrote", out_name)


if __name__ == "__main__":
    main()
    main(hu.pkg.Q4_K, "temporal_stack_q4k.npz")
    main(F32, "temporal_stack_personaplex.npz", model="personaplex")
