"""Generates tests/golden/hf_mimi.npz: the Mimi codec as implemented by Hugging Face `transformers` (models/mimi/modeling_mimi.py, an independent port of
Kyutai's codec by other authors) run OFFLINE over the frame driver's synthetic weights, against the driver's STREAMING encoder / decoder (the graph
construction of /root/reference/src/moshi/models/compression.h:149-325 over modules/{conv,seanet,transformer}.h and quantization/{vq,core_vq}.h).
Unlike make_mimi_golden.py (a restatement written for this repo), nothing of the architecture is restated here: the network is HF's module tree; only a
name / layout map of the weights is ours - SEANet convolutions 1:1; the fused in_proj split into q / k / v with the rows of q and k de-interleaved per head
(HF rotates halves, the reference rotates interleaved pairs and stores them de-interleaved: rope.h:33-128); codebooks as embed_sum over unit cluster usage.
Stored: the input audio, the two latents HF's RVQ stacks quantise per frame (the inputs of their first levels), a code sequence and the audio HF decodes
from it. tests/test_oracle_golden.py replays both directions through the driver (oracle, and the MI355X device) and compares. HF computes in plain float32
(no F16 im2col, no BF16 ring, erf-GELU where ggml goes through its F16 tanh table), so the bar is the codec's own sensitivity (1e-2 of max), not 1e-6.
Run in the build container only: `python tests/golden/make_hf_mimi_golden.py`."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import hot_util as hu  # noqa: E402
from make_mimi_golden import tensor  # noqa: E402  (reads one weight tensor of the driver's model as a torch tensor)

L = hu.L
FRAMES = 6
N_Q = int(os.environ.get("HF_MIMI_N_Q", "32"))   # round 5: all 32 RVQ levels (BASELINE.json configs[1] / [2] decode / encode 32)


def hf_model(m):
    from transformers import MimiConfig, MimiModel
    cfg = MimiConfig(num_quantizers=N_Q, codebook_size=2048)
    assert (cfg.hidden_size, cfg.num_attention_heads, cfg.num_hidden_layers, cfg.intermediate_size, cfg.sliding_window, cfg.num_filters, cfg.upsampling_ratios,
            cfg.codebook_dim, cfg.kernel_size, cfg.last_kernel_size, cfg.compress) == (512, 8, 8, 2048, 250, 64, [8, 6, 5, 4], 256, 7, 3, 2)
    model = MimiModel(cfg).eval().to(torch.float32)
    sd = model.state_dict()
    new = {}
    w = lambda n: tensor(m, n)

    def put(key, val):
        assert key in sd, key
        assert tuple(sd[key].shape) == tuple(val.shape), (key, tuple(sd[key].shape), tuple(val.shape))
        new[key] = val.to(torch.float32).contiguous()
    for side in ("encoder", "decoder"):
        for k in sd:
            if not k.startswith(side + ".layers."):
                continue
            idx, rest = k[len(side) + 8:].split(".", 1)                       # e.g. "3", "conv.weight" / "1", "block.1.conv.weight"
            ours = f"mimi.{side}.model.{idx}.{rest}"
            if side == "decoder" and rest in ("conv.weight", "conv.bias") and int(idx) in (2, 5, 8, 11):
                ours = f"mimi.decoder.model.{idx}.convtr.{rest.split('.')[1]}"  # the transposed convolutions
            t = w(ours)
            put(k, t.reshape(sd[k].shape) if rest.endswith("bias") else t)
    for side in ("encoder", "decoder"):
        for l in range(8):
            p, q = f"mimi.{side}_transformer.transformer.layers.{l}.", f"{side}_transformer.layers.{l}."
            win = w(p + "self_attn.in_projs.weight")                           # [3 * 512, 512]: q | k | v, heads of 64 rows
            D, H = 512, 8
            perm = torch.cat([torch.arange(0, 64, 2), torch.arange(1, 64, 2)])  # interleaved pairs -> [real half | imaginary half]
            rows = torch.cat([h * 64 + perm for h in range(H)])
            put(q + "self_attn.q_proj.weight", win[0:D][rows])
            put(q + "self_attn.k_proj.weight", win[D:2 * D][rows])
            put(q + "self_attn.v_proj.weight", win[2 * D:3 * D])
            put(q + "self_attn.o_proj.weight", w(p + "self_attn.out_projs.weight"))
            put(q + "mlp.fc1.weight", w(p + "linear1.weight"))
            put(q + "mlp.fc2.weight", w(p + "linear2.weight"))
            put(q + "input_layernorm.weight", w(p + "norm1.weight").reshape(-1)); put(q + "input_layernorm.bias", w(p + "norm1.bias").reshape(-1))
            put(q + "post_attention_layernorm.weight", w(p + "norm2.weight").reshape(-1)); put(q + "post_attention_layernorm.bias", w(p + "norm2.bias").reshape(-1))
            put(q + "self_attn_layer_scale.scale", w(p + "layer_scale_1.scale").reshape(-1))
            put(q + "mlp_layer_scale.scale", w(p + "layer_scale_2.scale").reshape(-1))
    put("downsample.conv.weight", w("mimi.downsample.conv.weight"))
    put("upsample.conv.weight", w("mimi.upsample.convtr.weight").reshape(sd["upsample.conv.weight"].shape))
    for ours, theirs, n in (("rvq_first", "semantic_residual_vector_quantizer", 1), ("rvq_rest", "acoustic_residual_vector_quantizer", N_Q - 1)):
        put(f"quantizer.{theirs}.input_proj.weight", w(f"mimi.quantizer.{ours}.input_proj.weight"))
        put(f"quantizer.{theirs}.output_proj.weight", w(f"mimi.quantizer.{ours}.output_proj.weight"))
        for i in range(n):
            cb = w(f"mimi.quantizer.{ours}.vq.layers.{i}._codebook.embedding")
            put(f"quantizer.{theirs}.layers.{i}.codebook.embed_sum", cb)
            put(f"quantizer.{theirs}.layers.{i}.codebook.cluster_usage", torch.ones(cb.shape[0]))
            put(f"quantizer.{theirs}.layers.{i}.codebook.initialized", torch.ones(1))
    missing = [k for k in sd if k not in new]
    assert not missing, missing
    model.load_state_dict(new, strict=True)
    for mod in model.modules():                      # the reference's streaming convolutions start from zero state everywhere (conv.h:36-96); HF's
        if getattr(mod, "pad_mode", None) == "replicate":   # offline downsampling convolution replicates the first sample instead
            mod.pad_mode = "constant"
    return model


def main():
    cfg = hu.hot.tiny(L)
    cfg.enable_lm = 0
    cfg.mimi_n_q, cfg.mimi_codebook_size = N_Q, 2048
    m = hu.Model("oracle", cfg, seed=0)
    model = hf_model(m)
    rng = np.random.default_rng(12)
    pcm = (rng.standard_normal(FRAMES * 1920) * 0.2).astype(np.float32)
    lat = {}
    hooks = [model.quantizer.semantic_residual_vector_quantizer.input_proj.register_forward_hook(lambda mod, i, o: lat.__setitem__("first", o.detach()[0])),
             model.quantizer.acoustic_residual_vector_quantizer.input_proj.register_forward_hook(lambda mod, i, o: lat.__setitem__("rest", o.detach()[0]))]
    with torch.no_grad():
        hf_codes = model.encode(torch.from_numpy(pcm)[None, None, :], num_quantizers=N_Q).audio_codes[0].numpy()      # [n_q, FRAMES]
    for h in hooks:
        h.remove()
    got_codes, got_first, got_rest = [], [], []
    for i in range(FRAMES):
        got_codes.append(m.mimi_encode(pcm[i * 1920:(i + 1) * 1920]))
        got_first.append(m.read("enc_latent_first", 256)); got_rest.append(m.read("enc_latent_rest", 256))
    got_codes = np.array(got_codes).T
    hf_first, hf_rest = lat["first"].numpy().T, lat["rest"].numpy().T                                                 # [FRAMES, 256]
    for i in range(FRAMES):
        print(f"encoder frame {i}: latent rel err first {hu.rel_err(hf_first[i], got_first[i]):.2e} rest {hu.rel_err(hf_rest[i], got_rest[i]):.2e}  codes equal {int((hf_codes[:, i] == got_codes[:, i]).sum())}/{N_Q}")
    # decoder: a code sequence through both
    codes = rng.integers(0, 2048, (FRAMES, N_Q)).astype(np.int32)
    with torch.no_grad():
        hf_pcm = model.decode(torch.from_numpy(codes.T.astype(np.int64))[None]).audio_values[0, 0].numpy()
    got = np.concatenate([m.mimi_decode(c.tolist()) for c in codes])
    for i in range(FRAMES):
        print(f"decoder frame {i}: pcm rel err {hu.rel_err(hf_pcm[i * 1920:(i + 1) * 1920], got[i * 1920:(i + 1) * 1920]):.2e}")
    m.free()
    np.savez_compressed(os.path.join(HERE, "hf_mimi.npz"), pcm=pcm, latent_first=hf_first, latent_rest=hf_rest, enc_codes=hf_codes.T.astype(np.int32),
                        codes=codes, pcm_out=hf_pcm.astype(np.float32))
    print("wrote hf_mimi.npz")


if __name__ == "__main__":
    main()
