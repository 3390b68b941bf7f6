"""ctypes binding of include/moshi_hot.h (the host-side frame driver + synthetic weights: libmoshi-hot.so, a harness library that calls
libggml-mi355x.so through the public ggml C API only)."""
import ctypes as C

MAX_CB = 33


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "num_heads", "num_layers", "ffn_hidden", "context", "max_period",
                                         "text_card", "card", "n_q", "dep_q")] + \
               [("delays", C.c_int32 * MAX_CB)] + \
               [(n, C.c_int32) for n in ("dep_dim", "dep_heads", "dep_layers", "dep_ffn_hidden", "dep_context",
                                         "linear_type", "embed_type", "mimi_n_q", "mimi_codebook_size",
                                         "enable_lm", "enable_mimi_encoder", "enable_mimi_decoder")] + \
               [("temp", C.c_float), ("temp_text", C.c_float), ("top_k", C.c_int32), ("top_k_text", C.c_int32)] + \
               [(n, C.c_int32) for n in ("personaplex", "extra_heads", "extra_heads_dim",
                                         "demux_second_stream", "depformer_low_rank", "delay_steps", "cross_attention", "cross_len",
                                         "condition_sum", "dep_schedule_len")] + \
               [("dep_schedule", C.c_int32 * MAX_CB), ("update_scale", C.c_float),
                ("dep_shard_rank", C.c_int32), ("dep_shard_world", C.c_int32), ("depth_only", C.c_int32),
                ("tp_rank", C.c_int32), ("tp_world", C.c_int32), ("codec_stream", C.c_int32), ("chain_depth", C.c_int32)]

    @property
    def io_dep_q(self):
        """codebooks the frame protocol generates per step (lm.h:802-805)"""
        return 8 if self.personaplex else self.dep_q


P = C.c_void_p
SIGNATURES = {
    "moshi_hot_config_moshika": (None, [C.POINTER(Config)]),
    "moshi_hot_config_personaplex": (None, [C.POINTER(Config)]),
    "moshi_hot_lm_step_n": (C.c_int, [P, P, C.c_int, P, P, P]),
    "moshi_hot_lm_step_run_ahead": (C.c_int, [P, P, P, P]),
    "moshi_hot_lm_step_embedding": (None, [P, P]),
    "moshi_hot_set_conditions": (None, [P, P, P]),
    "moshi_hot_prefill": (None, [P, P, C.c_int, C.c_int]),
    "moshi_hot_set_text_hook": (None, [P, P, P]),
    "moshi_hot_personaplex_prompt_tokens": (C.POINTER(C.c_int32), []),
    "moshi_hot_personaplex_system_prompts": (None, [P, P, C.c_int]),
    "moshi_hot_personaplex_system_prompts_batched": (None, [P, P, C.c_int, C.c_int]),
    "moshi_hot_create": (P, [P, C.POINTER(Config), C.c_uint64]),
    "moshi_hot_free": (None, [P]),
    "moshi_hot_save_gguf": (C.c_int, [P, C.c_char_p]),
    "moshi_hot_create_from_gguf": (P, [P, C.POINTER(Config), C.c_char_p]),
    "moshi_hot_tensor_file_name": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "moshi_hot_mimi_encode": (None, [P, P, P]),
    "moshi_hot_mimi_decode": (None, [P, P, P]),
    "moshi_hot_lm_step": (C.c_int, [P, P, P, P]),
    "moshi_hot_sts_frame": (C.c_int, [P, P, P, P, P]),
    "moshi_hot_sts_pipeline_begin": (None, [P, P]),
    "moshi_hot_sts_pipeline_frame": (C.c_int, [P, P, P, P, P]),
    "moshi_hot_sts_pipeline_end": (C.c_int, [P, P, P, P]),
    "moshi_hot_sts_pipeline_vad": (C.c_float, [P]),
    "moshi_hot_offset": (C.c_int64, [P]),
    "moshi_hot_weight_bytes": (C.c_size_t, [P, C.c_int]),
    "moshi_hot_read_last": (C.c_int, [P, C.c_char_p, P, C.c_int64]),
    "moshi_hot_weight": (P, [P, C.c_char_p]),
    "moshi_hot_graph": (P, [P, C.c_int]),
    "moshi_hot_set_context_fill": (None, [P, C.c_int64]),
    "moshi_hot_fill_ring": (None, [P, C.c_int, C.c_int, C.c_uint64, C.c_float]),
    "moshi_hot_ring_bytes": (C.c_int64, [P, C.c_int, C.c_int, C.c_int, P, C.c_int64, C.c_int]),
    "moshi_hot_host_ring": (C.c_int, [P, P, C.c_int]),
    "moshi_hot_force_last": (None, [P, C.c_int32, P]),
    "moshi_hot_set_timing": (None, [P, C.c_int]),
    "moshi_hot_get_timing": (None, [P, P]),
    "moshi_hot_last_raw_tokens": (None, [P, P, P]),
    "moshi_hot_layer_probe": (C.c_int, [P, C.c_int, C.c_int, C.c_int, P, C.c_int, P, P, P]),
    "moshi_hot_depth_shard_msg": (P, [P, C.POINTER(C.c_int64)]),
    "moshi_hot_depth_shard_tout": (P, [P, C.POINTER(C.c_int64)]),
    "moshi_hot_depth_shard_begin_export": (None, [P, C.c_int32, C.c_int]),
    "moshi_hot_depth_shard_begin_import": (C.c_int, [P]),
    "moshi_hot_depth_shard_step": (None, [P, C.c_int]),
    "moshi_hot_depth_shard_import": (None, [P, C.c_int]),
    "moshi_hot_depth_shard_tokens": (None, [P, P, C.c_int]),
    "moshi_hot_depth_shard_set_transport": (None, [P, C.c_void_p, C.c_void_p]),
    "moshi_hot_depth_shard_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "moshi_hot_depth_shard_rccl_init": (C.c_int, [P, C.c_int, C.c_int, C.c_char_p]),
    "moshi_hot_depth_shard_rccl_free": (None, [P]),
    "moshi_hot_depth_shard_broadcast": (None, [P, C.c_int, C.c_int]),
    "moshi_hot_depth_shard_install": (None, [P]),
    "moshi_hot_depth_shard_stop": (None, [P]),
    "moshi_hot_depth_shard_serve": (C.c_int64, [P]),
    "moshi_hot_depth_shard_hops": (C.c_int64, [P]),
    "moshi_hot_set_depth_hook": (None, [P, P, P]),
    "moshi_hot_tp_msg": (P, [P, C.POINTER(C.c_int64)]),
    "moshi_hot_tp_begin": (None, [P, P]),
    "moshi_hot_tp_segment": (None, [P, C.c_int]),
    "moshi_hot_tp_end": (None, [P, P]),
    "moshi_hot_tp_set_transport": (None, [P, C.c_void_p, C.c_void_p]),
    "moshi_hot_tp_stack": (None, [P, P, P]),
    "moshi_hot_tp_reductions": (C.c_int64, [P]),
    "moshi_hot_tp_install": (None, [P]), "moshi_hot_tp_serve": (C.c_int64, [P]), "moshi_hot_tp_stop": (None, [P]), "moshi_hot_tp_frames": (C.c_int64, [P]),
    "moshi_hot_tp_msg_read": (None, [P, P]),
    "moshi_hot_tp_msg_write": (None, [P, P]),
}
DEPTH_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int32, C.POINTER(C.c_int32))
BCAST_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_int64)
NODE_VISITOR = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)


def attach(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError -> loud failure in load()
        fn.restype, fn.argtypes = res, args


def moshika(lib):
    cfg = Config()
    lib.moshi_hot_config_moshika(C.byref(cfg))
    return cfg


def personaplex(lib):
    cfg = Config()
    lib.moshi_hot_config_personaplex(C.byref(cfg))
    return cfg


def tiny(lib, linear_type=12, embed_type=2, layers=2, dep_q=3, n_q=6, context=24):
    """A small model with the same structure as moshika (for parity tests that the oracle finishes in seconds)."""
    cfg = moshika(lib)
    cfg.dim, cfg.num_heads, cfg.num_layers, cfg.ffn_hidden, cfg.context = 512, 4, layers, 768, context
    cfg.text_card, cfg.card, cfg.n_q, cfg.dep_q = 500, 64, n_q, dep_q
    for i in range(MAX_CB):
        cfg.delays[i] = 0
    d = [0, 0] + [1] * (dep_q - 1) + [0] + [1] * (n_q - dep_q - 1) if dep_q > 0 else [0] + [0] * n_q
    for i, v in enumerate(d[:n_q + 1]):
        cfg.delays[i] = v
    cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden, cfg.dep_context = 256, 4, 2, 512, dep_q
    cfg.linear_type, cfg.embed_type = linear_type, embed_type
    cfg.mimi_n_q, cfg.mimi_codebook_size = n_q - dep_q, 64
    return cfg


def tiny_personaplex(lib, linear_type=12, embed_type=2, layers=2):
    """PersonaPlex's structure (17 codebooks, 16 chained Depth steps over a ring of 8, tools/personaplex-config.json) at test widths."""
    cfg = personaplex(lib)
    cfg.dim, cfg.num_heads, cfg.num_layers, cfg.ffn_hidden, cfg.context = 512, 4, layers, 768, 24
    cfg.text_card = 2100          # PROMPT_TOKENS reach id 2008
    cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden = 256, 4, 2, 512
    cfg.linear_type, cfg.embed_type = linear_type, embed_type
    return cfg


TEXT_HOOK = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int64, C.c_int32)


def tiny_tts(lib, linear_type=12, embed_type=2, layers=2, dep_q=8, cross_len=5):
    """The tts branches at test widths: cross-attention in every layer, demuxed text embeddings, low-rank
    Depth embeddings, a per-step weight schedule sharing weight sets, delay_steps, condition_sum; n_q == dep_q (no input streams)."""
    cfg = tiny(lib, linear_type=linear_type, embed_type=embed_type, layers=layers, dep_q=dep_q, n_q=dep_q)
    for i in range(MAX_CB):
        cfg.delays[i] = 0
    for i in range(2, dep_q + 1):
        cfg.delays[i] = 1 + (i % 2)
    cfg.card = 64
    cfg.demux_second_stream, cfg.depformer_low_rank, cfg.delay_steps = 1, 128, 2
    cfg.cross_attention, cfg.cross_len, cfg.condition_sum = 1, cross_len, 1
    sched = ([0, 1, 2, 3, 3, 3, 4, 4] + [5] * MAX_CB)[:dep_q]
    cfg.dep_schedule_len = len(sched)
    for i, v in enumerate(sched):
        cfg.dep_schedule[i] = v
    cfg.dep_context = 0            # ring capacity = len(schedule) (lm_default.h:86-90)
    cfg.mimi_n_q = dep_q
    return cfg


def tts_like(lib):
    """BASELINE.json configs[1] (moshi-tts tts-1.6b q8_0) at full size. The model's config.json is downloaded, not in the reference repo; widths follow
    the comments of include/moshi/moshi.h:111-139 (dim 2048, 16 heads, 16 layers, n_q = dep_q = 32, card 2048, text_card 8000, context 500, depformer
    1024 / 16 heads / 4 layers / feed-forward 3072, low-rank 128, demux, cross-attention); delays, schedule and condition length are placeholders of
    the right shape. Q8_0 linears and embeddings."""
    cfg = moshika(lib)
    cfg.dim, cfg.num_heads, cfg.num_layers, cfg.ffn_hidden, cfg.context = 2048, 16, 16, 5632, 500
    cfg.text_card, cfg.card, cfg.n_q, cfg.dep_q = 8000, 2048, 32, 32
    for i in range(MAX_CB):
        cfg.delays[i] = 0
    for i in range(2, 33):
        cfg.delays[i] = 2
    cfg.dep_dim, cfg.dep_heads, cfg.dep_layers, cfg.dep_ffn_hidden, cfg.dep_context = 1024, 16, 4, 2048, 0
    cfg.linear_type, cfg.embed_type = 8, 8
    cfg.demux_second_stream, cfg.depformer_low_rank, cfg.delay_steps = 1, 128, 16
    cfg.cross_attention, cfg.cross_len, cfg.condition_sum = 1, 64, 1
    sched = list(range(8)) + [8] * 24
    cfg.dep_schedule_len = 32
    for i, v in enumerate(sched):
        cfg.dep_schedule[i] = v
    cfg.mimi_n_q = 32
    cfg.enable_mimi_encoder = 0
    return cfg


def stt_like(lib):
    """BASELINE.json configs[2] (moshi-stt stt-1b q4_k) at full size, same caveat as tts_like: dim 2048, 16 layers, n_q 32 input codebooks, no Depth
    transformer, 3 extra heads of width 6 (the third is the VAD head), Mimi encoder with 32 RVQ levels."""
    cfg = moshika(lib)
    cfg.dim, cfg.num_heads, cfg.num_layers, cfg.ffn_hidden, cfg.context = 2048, 16, 16, 5632, 750
    cfg.text_card, cfg.card, cfg.n_q, cfg.dep_q = 8000, 2048, 32, 0
    for i in range(MAX_CB):
        cfg.delays[i] = 0
    cfg.extra_heads, cfg.extra_heads_dim = 3, 6
    cfg.mimi_n_q = 32
    cfg.enable_mimi_decoder = 0
    return cfg
