"""Generates tests/golden/hf_moshi.npz: Moshi's Temporal transformer and its chained Depth transformer as implemented by Hugging Face `transformers`
(models/moshi/modeling_moshi.py: MoshiForCausalLM, MoshiDepthDecoder - an independent port by other authors) run over the frame driver's synthetic F32
weights, against the driver's streaming LM step (graph construction of /root/reference/src/moshi/models/lm.h:446-677 over modules/transformer.h, rope.h,
gating.h). Nothing of the architecture is restated here (make_module_golden.py does that): the networks are HF's module trees; ours is only the name /
layout map - the fused in_proj split into q / k / v with the rows of q and k de-interleaved per head (HF rotates halves, the reference rotates interleaved
pairs), the per-step Depth weight sets stacked into HF's [codebook, out, in] tensors.
The Temporal model is fed the driver's own stack inputs (`transformer_in`: the embedding sums) for 10 provided frames - more than the ring of 6 holds, so
HF's sliding window and the driver's ring wrap are both exercised - and must reproduce `transformer_out` and the text logits; the Depth decoder is fed, per
frame, the driver's transformer_out, its text token and its audio tokens (teacher forcing, as HF's training-style forward expects) and must reproduce the
logits of every chained step.
Round 5: (a) ggml's BF16 rounding sites of the attention are emulated - HF's plain-float32 attention keeps K / V rows and probabilities in float32 where ggml
rounds the ring rows, the query and the probabilities to BF16 (mul_mat's activation operand takes the BF16 weight operand's type). They enter through
transformers' own extension point, `AttentionInterface.register`: the registered function is HF's eager_attention_forward with q, k, v and the probabilities
passed through `.bfloat16().float()` - four lines of soft_max(q k^T s + mask) v; projections, RoPE, norms, MLPs, residuals, the codebook-indexed linears
and the heads stay HF's module code. The bar drops from 5e-3 / 1e-2 to north_star's own 1e-3. (b) `--wide` writes hf_moshi_wide.npz: the same pin at
moshika's widths (dim 4096, 32 heads, FFN 11264, text vocabulary 32 000; Depth 1024 / 16 heads / 6 layers / 2816, 8 steps), 2 Temporal layers, a ring of 4
slots over 7 frames (it wraps), F32 weights.
Run in the build container only: `python tests/golden/make_hf_moshi_golden.py [--wide]`."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import hot_util as hu  # noqa: E402
import make_module_golden as mg  # noqa: E402  (weight(): one F32 weight matrix of the driver's model as a torch tensor)

L = hu.L
F32 = hu.pkg.F32
STEPS = 10


def deinterleave_rows(w, heads):
    """rows of q / k per head: interleaved (re, im) pairs -> [re half | im half] (rotate_half convention)"""
    dh = w.shape[0] // heads
    perm = torch.cat([torch.arange(0, dh, 2), torch.arange(1, dh, 2)])
    return w[torch.cat([h * dh + perm for h in range(heads)])]


def wide_config():
    cfg = hu.hot.moshika(L)
    cfg.linear_type = cfg.embed_type = F32
    cfg.num_layers, cfg.context = 2, 4
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def ggml_bf16_attention(module, query, key, value, attention_mask, scaling, dropout=0.0, **kwargs):
    """transformers' eager_attention_forward with ggml's rounding sites: ring rows (K, V), the query and the probabilities are BF16 values"""
    from transformers.models.moshi.modeling_moshi import repeat_kv
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    key_states, value_states = repeat_kv(key, module.num_key_value_groups), repeat_kv(value, module.num_key_value_groups)
    attn_weights = torch.matmul(bf(query), bf(key_states).transpose(2, 3)) * scaling
    if attention_mask is not None:
        attn_weights = attn_weights + attention_mask[:, :, :, : key_states.shape[-2]]
    attn_weights = torch.nn.functional.softmax(attn_weights, dim=-1, dtype=torch.float32)
    attn_output = torch.matmul(bf(attn_weights), bf(value_states)).transpose(1, 2).contiguous()
    return attn_output, attn_weights


def main(wide=False):
    from transformers import AttentionInterface, MoshiConfig
    from transformers.models.moshi.modeling_moshi import MoshiDepthDecoder, MoshiForCausalLM
    AttentionInterface.register("ggml_bf16_kv", ggml_bf16_attention)
    from transformers import AttentionMaskInterface
    from transformers.masking_utils import eager_mask
    AttentionMaskInterface.register("ggml_bf16_kv", eager_mask)   # (the Depth decoder builds its causal mask itself: additive float, as for "eager")
    global STEPS
    if wide:
        cfg, STEPS = wide_config(), 7
    else:
        cfg = hu.hot.tiny(L, linear_type=F32, embed_type=F32, layers=2, context=6)
        cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    m = hu.Model("oracle", cfg, seed=0)
    D, H, F, Q = cfg.dim, cfg.num_heads, cfg.ffn_hidden, cfg.dep_q
    DD, DH_, DF = cfg.dep_dim, cfg.dep_heads, cfg.dep_ffn_hidden
    hc = MoshiConfig(vocab_size=cfg.text_card, hidden_size=D, num_hidden_layers=cfg.num_layers, num_attention_heads=H, num_key_value_heads=H, audio_vocab_size=cfg.card,
                     ffn_dim=2 * F, num_codebooks=Q, sliding_window=cfg.context, max_position_embeddings=64, rms_norm_eps=1e-8, rope_theta=float(cfg.max_period),
                     depth_decoder_config=dict(vocab_size=cfg.text_card, hidden_size=DD, num_hidden_layers=cfg.dep_layers, num_attention_heads=DH_, num_key_value_heads=DH_,
                                               audio_vocab_size=cfg.card, ffn_dim=2 * DF, num_codebooks=Q, input_size=D, sliding_window=Q, max_position_embeddings=Q,
                                               rms_norm_eps=1e-8))
    hc._attn_implementation = "ggml_bf16_kv"
    hc.depth_decoder_config._attn_implementation = "ggml_bf16_kv"
    lm = MoshiForCausalLM(hc).eval().to(torch.float32)
    dd = MoshiDepthDecoder(hc.depth_decoder_config).eval().to(torch.float32)
    w = lambda n, r, c: mg.weight(m, n, r, c)
    sd, new = lm.state_dict(), {}

    def put(d, key, val):
        assert key in d and tuple(d[key].shape) == tuple(val.shape), (key, tuple(d[key].shape) if key in d else None, tuple(val.shape))
        new[key] = val.to(torch.float32).contiguous()
    for l in range(cfg.num_layers):
        p, q = f"lm.transformer.layers.{l}.", f"model.layers.{l}."
        win = w(p + "self_attn.in_projs.weight", 3 * D, D)
        put(sd, q + "self_attn.q_proj.linear.weight", deinterleave_rows(win[:D], H))
        put(sd, q + "self_attn.k_proj.linear.weight", deinterleave_rows(win[D:2 * D], H))
        put(sd, q + "self_attn.v_proj.linear.weight", win[2 * D:])
        put(sd, q + "self_attn.o_proj.linear.weight", w(p + "self_attn.out_projs.weight", D, D))
        put(sd, q + "mlp.fc1.weight", w(p + "gating.linear_in.weight", 2 * F, D))
        put(sd, q + "mlp.fc2.weight", w(p + "gating.linear_out.weight", D, F))
        put(sd, q + "input_layernorm.weight", w(p + "norm1.alpha", 1, D)[0])
        put(sd, q + "post_attention_layernorm.weight", w(p + "norm2.alpha", 1, D)[0])
    put(sd, "model.norm.weight", w("lm.out_norm.alpha", 1, D)[0])
    put(sd, "lm_head.weight", w("lm.text_linear.weight", cfg.text_card, D))
    new["model.embed_tokens.weight"] = sd["model.embed_tokens.weight"]          # (unused: the stack is fed the driver's embedding sums)
    assert not [k for k in sd if k not in new], [k for k in sd if k not in new]
    lm.load_state_dict(new, strict=True)
    sd, new = dd.state_dict(), {}
    put(sd, "text_embed_tokens.weight", w("lm.depformer_text_emb.weight", cfg.text_card + 1, DD))
    for k in range(Q - 1):
        put(sd, f"embed_tokens.{k}.weight", w(f"lm.depformer_emb.{k}.weight", cfg.card + 1, DD))
    put(sd, "input_projections.weight", torch.stack([w(f"lm.depformer_in.{k}.weight", DD, D) for k in range(Q)]))
    for l in range(cfg.dep_layers):
        p, q = f"lm.depformer.layers.{l}.", f"layers.{l}."
        wins = [w(p + f"self_attn.in_projs.{k}.weight", 3 * DD, DD) for k in range(Q)]
        put(sd, q + "self_attn.q_proj.linear.weight", torch.stack([x[:DD] for x in wins]))            # (no RoPE in the Depth transformer: rows as they are)
        put(sd, q + "self_attn.k_proj.linear.weight", torch.stack([x[DD:2 * DD] for x in wins]))
        put(sd, q + "self_attn.v_proj.linear.weight", torch.stack([x[2 * DD:] for x in wins]))
        put(sd, q + "self_attn.o_proj.linear.weight", torch.stack([w(p + f"self_attn.out_projs.{k}.weight", DD, DD) for k in range(Q)]))
        put(sd, q + "mlp.fc1.weight", torch.stack([w(p + f"gating.{k}.linear_in.weight", 2 * DF, DD) for k in range(Q)]))
        put(sd, q + "mlp.fc2.weight", torch.stack([w(p + f"gating.{k}.linear_out.weight", DD, DF) for k in range(Q)]))
        put(sd, q + "input_layernorm.weight", w(p + "norm1.alpha", 1, DD)[0])
        put(sd, q + "post_attention_layernorm.weight", w(p + "norm2.alpha", 1, DD)[0])
    put(sd, "lm_heads.weight", torch.stack([w(f"lm.linears.{k}.weight", cfg.card, DD) for k in range(Q)]))
    assert not [k for k in sd if k not in new], [k for k in sd if k not in new]
    dd.load_state_dict(new, strict=True)

    rng = np.random.default_rng(14)
    toks, xs, outs, logits, text_tokens, dep_tokens, dep_logits = [], [], [], [], [], [], []
    for step in range(STEPS):
        tokens = [int(rng.integers(0, cfg.text_card))] + rng.integers(0, cfg.card, cfg.n_q).tolist()     # a "provided" frame: every codebook given
        m.lm_step_n(tokens)
        toks.append(tokens)
        xs.append(m.read("transformer_in", D).copy())
        outs.append(m.read("transformer_out", D).copy()); logits.append(m.read("text_logits", cfg.text_card).copy())
        tt, aud = m.last_raw()
        text_tokens.append(tt); dep_tokens.append(aud)
        dep_logits.append(np.array([m.read(f"dep_logits{k}", cfg.card).copy() for k in range(Q)]))
    m.free()
    with torch.no_grad():
        # HF's decoder layers, norm and head, called one by one with an explicit additive mask: causal AND limited to the last `context` positions - what the
        # driver's ring of 6 slots holds (MoshiModel.forward builds no mask at all unless it is handed one, and then not a sliding one)
        T = STEPS
        qi, ki = torch.arange(T)[:, None], torch.arange(T)[None, :]
        mask = torch.zeros(T, T).masked_fill(~((ki <= qi) & (ki > qi - cfg.context)), float("-inf"))[None, None]
        hs, pos = torch.from_numpy(np.array(xs))[None], torch.arange(T)[None]
        for layer in lm.model.layers:
            hs = layer(hs, attention_mask=mask, position_ids=pos, past_key_values=None, use_cache=False)[0]
        hf_out = lm.model.norm(hs)
        hf_logits = lm.lm_head(hf_out)[0].numpy()                          # [STEPS, text_card]
        hf_out = hf_out[0].numpy()
        hf_dep = []
        for step in range(STEPS):
            ids = torch.tensor([[text_tokens[step]] + dep_tokens[step][:Q - 1]])
            d = dd(input_ids=ids, last_hidden_state=torch.from_numpy(outs[step])[None, None, :], attention_mask=torch.ones(1, Q, dtype=torch.long), use_cache=False)
            hf_dep.append(d.logits[0].numpy())                            # [Q, card]
    hf_dep = np.array(hf_dep)
    for step in range(STEPS):
        e1, e2 = hu.rel_err(hf_out[step], outs[step]), hu.rel_err(hf_logits[step], logits[step])
        e3 = max(hu.rel_err(hf_dep[step][k], dep_logits[step][k]) for k in range(Q))
        agree = [int(np.argmax(hf_dep[step][k])) for k in range(Q)] == dep_tokens[step]
        print(f"step {step}: transformer_out {e1:.2e}  text logits {e2:.2e} (argmax {int(np.argmax(hf_logits[step]))} vs {text_tokens[step]})  depth logits {e3:.2e}  depth tokens agree {agree}")
    np.savez_compressed(os.path.join(HERE, "hf_moshi_wide.npz" if wide else "hf_moshi.npz"), tokens=np.array(toks, np.int32), x_in=np.array(xs), transformer_out=hf_out, text_logits=hf_logits,
                        text_tokens=np.array(text_tokens, np.int32), dep_tokens=np.array(dep_tokens, np.int32), dep_logits=hf_dep)
    print("wrote", "hf_moshi_wide.npz" if wide else "hf_moshi.npz")


if __name__ == "__main__":
    main(wide="--wide" in sys.argv)
