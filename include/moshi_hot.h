// moshi_hot.h — C-ABI of the host-side driver of the streaming-decode hot path.
//
// In the reference this layer is libmoshi's C++ API (include/moshi/moshi.h:24-203): it builds the ggml
// graphs of the Temporal transformer, the chained Depth transformer and the Mimi codec once, then
// submits them every 80 ms frame. libmoshi itself stays the caller in a real
// deployment (it links this library's ggml surface unchanged); it cannot be built in
// this environment (SentencePiece / FFmpeg / model files are absent), so this driver restates the same
// graph construction and frame protocol — same op sequences, same upload / compute / read-back order —
// over synthetic weights, for the parity tests and for bench.py. Every function cites what it mirrors.
//
// Everything here runs on whichever ggml backend it is handed: the MI355X device, or the host device
// with the parity oracle attached (tests only).
#pragma once

#include "ggml.h"
#include "ggml-backend.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MOSHI_HOT_MAX_CODEBOOKS 33

// model hyper-parameters (tools/moshi-config.json keys; src/config.h:148-346)
struct moshi_hot_config {
    // Temporal transformer
    int32_t dim, num_heads, num_layers, ffn_hidden, context, max_period;
    int32_t text_card, card, n_q, dep_q;
    int32_t delays[MOSHI_HOT_MAX_CODEBOOKS];      // n_q + 1 entries
    // Depth transformer ("depformer")
    int32_t dep_dim, dep_heads, dep_layers, dep_ffn_hidden, dep_context;
    // weight storage types (ggml_type): linear layers / embedding tables
    int32_t linear_type, embed_type;
    // Mimi codec
    int32_t mimi_n_q;            // RVQ levels used by encoder and decoder (8 for moshi-sts)
    int32_t mimi_codebook_size;  // 2048
    int32_t enable_lm, enable_mimi_encoder, enable_mimi_decoder;
    // sampling: temp <= 0 -> greedy argmax (the parity mode, sampling.h:57-63)
    float   temp, temp_text;
    int32_t top_k, top_k_text;
    // model variants (all zero = moshika)
    int32_t personaplex;             // "model_type": "personaplex" (lm_default.h:223): the Depth graph chains dep_q (16) steps, the frame
                                     // protocol exposes 8 of them and takes the other speaker's 8 (lm.h:803-805); delay ring one row deeper (lm.h:728)
    int32_t extra_heads, extra_heads_dim;   // stt: linear heads on transformer_out; extra_heads[2] is the VAD head (lm.h:966-976)
    // tts (BASELINE.json configs[1])
    int32_t demux_second_stream;     // text embeddings E[id % (text_card+1)]·out1 + s·E[id / (text_card+1) - 1]·out2 (lm_utils.h:48-85)
    int32_t depformer_low_rank;      // width of the Depth embedding tables, followed by a low_rank linear to dep_dim (lm_utils.h:157-217); 0 = none
    int32_t delay_steps;             // Depth skipped while offset < delay_steps; tokens -1 until delays[q+1] + delay_steps (src/moshi.cpp:905, lm.h:915-921)
    int32_t cross_attention, cross_len;   // per-layer cross-attention over condition_cross F32[dim, cross_len] (transformer.h:343-396, 714-762)
    int32_t condition_sum;           // sum_condition F32[dim] added to the embedding sum (lm.h:579-581)
    int32_t dep_schedule_len;        // depformer_weights_per_step_schedule (lm_default.h:71-81); also the Depth ring capacity when dep_context == 0
    int32_t dep_schedule[MOSHI_HOT_MAX_CODEBOOKS];
    // synthetic weights only: standard deviation of the residual-update projections (out_proj, gating linear_out) relative to the
    // default 1/sqrt(fan_in); 0 = 1. Values < 1 give a CONTRACTIVE stack (every layer's update is small against the residual stream), on
    // which ggml's discontinuous roundings no longer compound chaotically: the parity tests then assert bit-exact greedy tokens and 1e-3
    // logits over many free-running frames at the full benchmark widths (same shapes, types and bytes moved).
    float   update_scale;
    // Depth-transformer codebook shard (lm.h:505-527, lm_default.h:136-146,187-216). dep_shard_world > 1: this model holds
    // the per-step Depth weight sets (depformer_in[k], in_projs[k], out_projs[k], gating[k], linears[k], depformer_emb[k-1]) of the steps
    // k with k % dep_shard_world == dep_shard_rank only - 1/world of the 375 MB - and runs the Depth chain as per-step graphs behind the
    // moshi_hot_depth_shard_* calls. depth_only: no Temporal stack, embeddings or codec at all (the ranks other than the Temporal owner).
    int32_t dep_shard_rank, dep_shard_world, depth_only;
    // Tensor-parallel Temporal stack (moshi_streaming_transformer_layer, transformer.h:910-1039). tp_world > 1: this model holds
    // rank tp_rank's SLICES of every Temporal layer - in_proj rows of its heads (q | k | v), the matching 1/N column block of out_proj, linear_in rows
    // [r F/N, (r+1) F/N) of both gate halves, the matching column block of linear_out - and its heads' KV rings; the stack runs as 2 L + 1 segment
    // graphs with a sum over ranks of one F32[dim] partial between them (moshi_hot_tp_* below). Column blocks fall on 256-value boundaries, so
    // the Q8_K activation blocks and every integer block dot are those of the unsplit layer; only the final float sums split.
    int32_t tp_rank, tp_world;
    // 1: the Mimi encode / decode graphs (and their scratch graphs) are built on a second command stream of the same GPU
    // (ggml_backend_mi355x_init_stream), so that moshi_hot_sts_pipeline_frame overlaps the codec of neighbouring frames with the LM step.
    // Ignored (one stream) on any other backend. Results are the same either way.
    int32_t codec_stream;
    // 1: the Depth graph takes the text token straight from the Temporal graph's sampler output in device memory and is queued right behind it - one host
    // wait per LM step instead of two - and the next frame's Temporal step inputs (mask row, RoPE phase, ring slot: functions of the stream position
    // only) are queued behind the Depth graph. Same graphs otherwise, same tokens. Not with a text hook, a Depth hook, demux or delay_steps (the
    // reference's host code inspects the text token between the two graphs there, lm.h:880-921).
    // 2: additionally the Temporal graph's embedding indices of the model's own codebooks are views of the same device-side token state, and
    // moshi_hot_sts_pipeline_frame runs AHEAD: step k is queued before step k - 1's tokens have been read (moshika-shaped models only; anything else
    // falls back to blocking steps behind the same calls).
    int32_t chain_depth;
};

typedef struct moshi_hot_model moshi_hot_model_t;

// fills cfg with tools/moshi-config.json (moshika-7B) under `-q q4_k`: Q4_K linears, Q4_0 embeddings
GGML_API void moshi_hot_config_moshika(struct moshi_hot_config * cfg);
// tools/personaplex-config.json under `-q q4_k` (BASELINE.json configs[4]; run with context 2000 there)
GGML_API void moshi_hot_config_personaplex(struct moshi_hot_config * cfg);

// allocates weights (synthetic, deterministic in `seed`), persistent state (KV rings, conv tails) and
// the scratch contexts on `backend` (moshi_alloc / mimi_alloc / moshi_lm_load / moshi_lm_start,
// src/moshi.cpp:88-130, 851-898)
GGML_API moshi_hot_model_t * moshi_hot_create(ggml_backend_t backend, const struct moshi_hot_config * cfg, uint64_t seed);
GGML_API void moshi_hot_free(moshi_hot_model_t * m);

// mimi_encode_send + mimi_encode_receive (src/moshi.cpp:215-234): 1920 samples -> mimi_n_q codes
GGML_API void moshi_hot_mimi_encode(moshi_hot_model_t * m, const float * pcm, int32_t * codes);
// mimi_decode_send + mimi_decode_receive (src/moshi.cpp:273-292): mimi_n_q codes -> 1920 samples
GGML_API void moshi_hot_mimi_decode(moshi_hot_model_t * m, const int32_t * codes, float * pcm);
// moshi_lm_send2 + moshi_lm_receive (src/moshi.cpp:904-926) = one moshi_lmgen_step (lm.h:778-979):
// in_audio = the (n_q - dep_q) codes of the other speaker; returns 1 when text/out_audio are valid
GGML_API int moshi_hot_lm_step(moshi_hot_model_t * m, const int32_t * in_audio, int32_t * text_token, int32_t * out_audio);
// moshi_lmgen_step in full (lm.h:778-979). n_tokens == n_q + 1: "provided" mode — every codebook of this frame is given (text first),
// the model still steps but its samples are not written to the delay ring: PersonaPlex prompt frames (lm.h:1063-1075, 1088-1097).
// Otherwise tokens = the other speaker's (n_q - dep_q) codes as in moshi_hot_lm_step. vad (may be NULL): softmax(extra_heads[2]·transformer_out)[0].
GGML_API int moshi_hot_lm_step_n(moshi_hot_model_t * m, const int32_t * tokens, int n_tokens, int32_t * text_token, int32_t * out_audio, float * vad);
// The LM step alone under run-ahead (config.chain_depth = 2; the codec-free half of moshi_hot_sts_pipeline_frame): queues the step of in_audio behind
// the previous one and returns the PREVIOUS step's result (1 / 0 like moshi_hot_lm_step, -1 when there is none yet). in_audio = NULL drains the last step.
GGML_API int moshi_hot_lm_step_run_ahead(moshi_hot_model_t * m, const int32_t * in_audio, int32_t * text_token, int32_t * out_audio);
// tts conditions: sum F32[dim] (may be NULL) and cross F32[dim * cross_len] (may be NULL); the cross-attention K/V of every layer are
// computed once here (init(), transformer.h:343-396). Call before the first step.
GGML_API void moshi_hot_set_conditions(moshi_hot_model_t * m, const float * sum, const float * cross);
// on_text_hook (lm.h:880-900): replaces the sampled text token (the TTS state machine lives above this boundary)
typedef int32_t (*moshi_hot_text_hook_t)(void * user, int64_t offset, int32_t sampled);
GGML_API void moshi_hot_set_text_hook(moshi_hot_model_t * m, moshi_hot_text_hook_t hook, void * user);
// one voice-prompt frame from a precomputed input embedding F32[dim] (moshi_lmgen_step_voice_prompt, lm.h:1004-1037): the Temporal
// stack runs on the scratch context (moshi_lmmodel_forward_embedding, lm.h:694-709), text is forced to 3, the Depth graph steps
GGML_API void moshi_hot_lm_step_embedding(moshi_hot_model_t * m, const float * embedding);
// n_frames provided frames (tokens: n_frames x (n_q + 1), text first) as batched [dim, T] passes of at most `chunk` frames (0 = 64, the largest the batched kernels take):
// leaves the delay ring, the offsets and the Temporal KV ring as n_frames calls of moshi_hot_lm_step_n(.., n_q + 1, ..) would,
// without running the Depth graph or the text head
GGML_API void moshi_hot_prefill(moshi_hot_model_t * m, const int32_t * tokens, int n_frames, int chunk);
// PROMPT_TOKENS (lm.h:983-987): 17 ids, text first
GGML_API const int32_t * moshi_hot_personaplex_prompt_tokens(void);
// moshi_lmgen_step_system_prompts without a voice (lm.h:1118-1134): 6 silence frames, the text prompt, 6 silence frames
GGML_API void moshi_hot_personaplex_system_prompts(moshi_hot_model_t * m, const int32_t * text_prompt, int n_text);
// the same 12 + n_text prompt frames as batched passes (moshi_hot_prefill): same state afterwards
GGML_API void moshi_hot_personaplex_system_prompts_batched(moshi_hot_model_t * m, const int32_t * text_prompt, int n_text, int chunk);
// one iteration of the moshi-sts --bench loop (tools/moshi-sts.cpp:770-808); returns 1 when a frame was produced
GGML_API int moshi_hot_sts_frame(moshi_hot_model_t * m, const float * pcm_in, int32_t * text_token, int32_t * audio_tokens, float * pcm_out);

// The moshi-sts --bench loop software-pipelined (tools/moshi-sts.cpp:770-808 feeds every frame without waiting for playback): the LM step of
// frame k runs on the backend's stream while the codec stream (config.codec_stream) decodes frame k - 1 and encodes frame k + 1. begin: encodes
// frame 0. frame: pcm_next = input of frame k + 1 (NULL: none follows); returns bit 0 = text_token / audio_tokens are valid, bit 1 = pcm_prev holds
// the output of frame k - 1, bit 2 = run-ahead (chain_depth = 2): the tokens are frame k - 1's too, because the LM step of frame k was queued behind
// the previous one before the host looked at that one's tokens (they reach the next Temporal graph through device memory; the LM stream never waits
// for the host). end: finishes what is outstanding: bit 0 = tokens of the last frame (run-ahead only), bit 1 = pcm_last is the last frame's output.
// Tokens and PCM are bit-identical to moshi_hot_sts_frame's: every graph consumes the same inputs and states in the same order.
GGML_API void moshi_hot_sts_pipeline_begin(moshi_hot_model_t * m, const float * pcm0);
GGML_API int  moshi_hot_sts_pipeline_frame(moshi_hot_model_t * m, const float * pcm_next, int32_t * text_token, int32_t * audio_tokens, float * pcm_prev);
GGML_API int  moshi_hot_sts_pipeline_end(moshi_hot_model_t * m, int32_t * text_token, int32_t * audio_tokens, float * pcm_last);
// stt / tts shaped models run the same loop with the codec half they have (encode of frame k + 1, or decode of frame k - 1, beside the LM step of frame k);
// the VAD head's probability of the last stepped frame (lm.h:966-976) is read here
GGML_API float moshi_hot_sts_pipeline_vad(moshi_hot_model_t * m);

// introspection for tests / bench
GGML_API int64_t moshi_hot_offset(moshi_hot_model_t * m);                      // frames stepped so far
GGML_API void    moshi_hot_last_raw_tokens(moshi_hot_model_t * m, int32_t * text_token, int32_t * audio_tokens);  // sampled this step, before the delay ring
GGML_API size_t  moshi_hot_weight_bytes(moshi_hot_model_t * m, int part);     // 0 temporal, 1 depth, 2 mimi enc, 3 mimi dec, 4 embeddings, 5 tensor-parallel slices (a subset of 0)
GGML_API int     moshi_hot_read_last(moshi_hot_model_t * m, const char * what, float * out, int64_t n);  // "text_logits", "transformer_out", "transformer_in", "stack_out", "dep_logits<k>", "enc_latent_first" / "enc_latent_rest" (F32[256]: what the RVQ stacks quantise)
// a weight tensor by its checkpoint-style name (e.g. "lm.transformer.layers.0.self_attn.in_projs.weight"); tests/ref_link uses it
// to hand the SAME tensors to the reference's own graph builders
GGML_API struct ggml_tensor * moshi_hot_weight(moshi_hot_model_t * m, const char * name);
// the cached graph of a phase once it has run (0 temporal, 1 depth, 2 mimi encoder, 3 mimi decoder), for structural comparison
GGML_API struct ggml_cgraph * moshi_hot_graph(moshi_hot_model_t * m, int which);
// per-phase wall clock (synchronises around each phase while on): us_per_call[4] = mimi encode, temporal, depth, mimi decode
GGML_API void    moshi_hot_set_timing(moshi_hot_model_t * m, int on);
GGML_API void    moshi_hot_get_timing(moshi_hot_model_t * m, double * us_per_call);
// teacher forcing for parity runs: overwrite the tokens the last moshi_hot_lm_step wrote into the delay ring
GGML_API void    moshi_hot_force_last(moshi_hot_model_t * m, int32_t text_token, const int32_t * audio_tokens);
GGML_API void    moshi_hot_set_context_fill(moshi_hot_model_t * m, int64_t offset); // jump the Temporal ring to a given fill level (bench only)
// write the same pseudo-random BF16 rows (approximately N(0, scale^2), a function of seed / which / layer / position only) into EVERY slot of the K and V
// rings (transformer.h:156-172) of one layer (layer >= 0) or of all layers (-1) of the Temporal (which = 0) or Depth (1) transformer: two executors
// filled alike hold identical caches, so attention over hundreds or thousands of live slots can be compared node by node (tests only)
// the reference's checkpoint path end to end (src/loader.h:85-99, 227-271): write every weight tensor of a model to a GGUF file / build a model whose
// weights are read back from such a file (names, types and sizes checked against the configuration) instead of being generated
GGML_API int     moshi_hot_save_gguf(moshi_hot_model_t * m, const char * path);
GGML_API moshi_hot_model_t * moshi_hot_create_from_gguf(ggml_backend_t backend, const struct moshi_hot_config * cfg, const char * path);
// the name a checkpoint tensor carries inside a GGUF file (WeightLoader::tensor_name, src/loader.h:120-137): the name itself below GGML_MAX_NAME characters,
// else the reference's 8-character CRC digest (src/crc-bbf.h). out receives at most n - 1 characters + NUL; returns the length of the file name
GGML_API int     moshi_hot_tensor_file_name(const char * checkpoint_name, char * out, int n);
// test hook: the host-side delay ring (rows x (n_q + 1) int32, row-major) -> dst; returns the value count (dst NULL: just the count)
GGML_API int     moshi_hot_host_ring(moshi_hot_model_t * m, int32_t * dst, int max_values);
GGML_API void    moshi_hot_fill_ring(moshi_hot_model_t * m, int which, int layer, uint64_t seed, float scale);
// the K (kv = 0) / V (kv = 1) ring of one layer as its bytes (BF16 [D, C, H]), read out (write = 0) or overwritten (write = 1): parity runs that restart every
// frame from another executor's state. Returns the ring's size in bytes (-1: no such ring); buf may be NULL to ask for the size. which: 0 Temporal, 1 Depth.
GGML_API int64_t moshi_hot_ring_bytes(moshi_hot_model_t * m, int which, int layer, int kv, void * buf, int64_t nbytes, int write);
// Parity probe: ONE transformer layer (moshi_streaming_transformer_layer, transformer.h:910-1039) of the Temporal (which = 0) or Depth
// (which = 1, with weight set `weight_set`) stack on the scratch context, fed x_in F32[dim] at stream position `offset` (mask row, RoPE
// phase and ring slot as transformer.h:1182-1215 computes them), over the model's own weights and KV ring of that layer (the new K / V
// rows are written at slot offset % capacity). Every tensor of the graph buffer is poisoned with 0xFF bytes before the compute, so a node
// a backend never materialised (fused away) reads back as NaN. After the compute `visit` is called for every node in execution order while
// the buffer is still alive (read values with ggml_backend_tensor_get); x_out (may be NULL) receives the layer output. Returns the node count.
typedef void (*moshi_hot_node_visitor_t)(void * user, int index, struct ggml_tensor * node);
GGML_API int     moshi_hot_layer_probe(moshi_hot_model_t * m, int which, int layer, int weight_set, const float * x_in, int offset, float * x_out,
                                       moshi_hot_node_visitor_t visit, void * user);

// ---- Depth codebook shard: one frame = begin, then for k = 0 .. dep_q-1 the owner runs `step` and everybody else `import` ---------------
// The 8-slot Depth ring is REPLICATED on every rank; what an owner hands on per step is one message of dep_layers * dep_dim + 8 32-bit words:
// its new K and V ring rows of all layers as the BF16 values the ring stores (2 * dep_layers * dep_dim of them: 24 KB at 6 x 1024) and, in the tail,
// the sampled token as F32 (+ the frame's stop flag in word 1 of the tail).
// The transport between ranks is the caller's (torch.distributed over RCCL / xGMI in bench.py, gloo in the CPU test): the message lives in
// one tensor whose storage pointer is returned here - a device pointer on the MI355X backend - so it can be handed to a collective as is.
GGML_API void *  moshi_hot_depth_shard_msg(moshi_hot_model_t * m, int64_t * n_floats);        // step message (rows + token)
GGML_API void *  moshi_hot_depth_shard_tout(moshi_hot_model_t * m, int64_t * n_floats);       // frame message: transformer_out F32[dim] + 8 (word dim = 1 while frames follow)
GGML_API void    moshi_hot_depth_shard_begin_export(moshi_hot_model_t * m, int32_t text_token, int more);   // Temporal owner: pack the frame message; text token feeds step 0
GGML_API int     moshi_hot_depth_shard_begin_import(moshi_hot_model_t * m);                   // others: unpack it; returns the "more frames" flag
GGML_API void    moshi_hot_depth_shard_step(moshi_hot_model_t * m, int k);                    // owner of step k: one Depth step (lm.h:505-527 body) + pack the message
GGML_API void    moshi_hot_depth_shard_import(moshi_hot_model_t * m, int k);                  // non-owner: rows -> ring slot k % capacity, token -> token vector
GGML_API void    moshi_hot_depth_shard_tokens(moshi_hot_model_t * m, int32_t * out, int n);   // the frame's sampled tokens so far
// The whole sharded frame behind the C-ABI - no interpreter on the critical path: the per-step loop above (owner: step + broadcast, the others: broadcast
// + import) runs inside these calls, the broadcasts go through ONE transport:
//   * RCCL, called from here (librccl.so is opened at run time; the harness links no HIP): ncclBroadcast on the backend's own HIP stream
//     (ggml_backend_mi355x_get_stream), i.e. stream-ordered behind the step's kernels and in front of the next ones - no host synchronisation and no
//     event per hop. Rank 0 makes the 128-byte unique id, the caller carries it to the other ranks (any side channel), every rank calls _rccl_init.
//   * or a caller-supplied function (host memory on the CPU device: gloo in tests/test_depth_shard_cpu.py; also a bring-up aid on devices).
typedef void (*moshi_hot_bcast_t)(void * user, void * data, int64_t bytes, int root);
GGML_API void    moshi_hot_depth_shard_set_transport(moshi_hot_model_t * m, moshi_hot_bcast_t fn, void * user);
GGML_API int     moshi_hot_depth_shard_rccl_unique_id(char * id128);                                            // 0 = ok (rank 0)
// _rccl_init makes the model's backend device current on the calling thread before ncclCommInitRank (the communicator binds to the current device) and
// replaces a communicator an earlier call made; the broadcasts / all-reduces return RCCL's code through GGML_ASSERT (a failed collective is not recoverable here).
GGML_API int     moshi_hot_depth_shard_rccl_init(moshi_hot_model_t * m, int rank, int world, const char * id128); // 0 = ok; every rank, collectively
GGML_API void    moshi_hot_depth_shard_rccl_free(moshi_hot_model_t * m);
GGML_API void    moshi_hot_depth_shard_broadcast(moshi_hot_model_t * m, int which, int root);                    // one broadcast of the step (0) / frame (1) message through the transport (latency probes)
// Temporal owner (rank 0): installs the sharded frame as the Depth half of moshi_hot_lm_step_n (export, broadcast, dep_q hops, read the tokens)
GGML_API void    moshi_hot_depth_shard_install(moshi_hot_model_t * m);
GGML_API void    moshi_hot_depth_shard_stop(moshi_hot_model_t * m);                                            // rank 0: tells the other ranks' _serve to return
GGML_API int64_t moshi_hot_depth_shard_hops(moshi_hot_model_t * m);                                            // broadcasts this rank has taken part in so far
GGML_API int64_t moshi_hot_depth_shard_serve(moshi_hot_model_t * m);                                           // every other rank: frames served until rank 0 stops
// ---- tensor-parallel Temporal stack: begin(x) ; for i in 0 .. 2 L: { segment(i) ; [i < 2 L: all-reduce(sum) of the message over the ranks] } ; end(out) ----
// segment 2 l   : x += message (l > 0) ; message = out_proj_r(attention_r(in_proj_r(norm1(x))))      (this rank's heads, its KV ring shard)
// segment 2 l+1 : x += message         ; message = linear_out_r(silu * mul of linear_in_r(norm2(x)))  (this rank's F / N hidden units)
// segment 2 L   : x += message         (the stack output)
GGML_API void *  moshi_hot_tp_msg(moshi_hot_model_t * m, int64_t * n_floats);          // the partial F32[dim] (device / host pointer: hand it to the collective)
GGML_API void    moshi_hot_tp_begin(moshi_hot_model_t * m, const float * x);            // stack input F32[dim]; advances the stream position (mask row, RoPE phase, ring slot)
GGML_API void    moshi_hot_tp_segment(moshi_hot_model_t * m, int i);
GGML_API void    moshi_hot_tp_end(moshi_hot_model_t * m, float * out);                  // stack output F32[dim]
// the same loop behind the C-ABI: every segment and the all-reduce (sum) of the partial between two of them - ncclAllReduce on the backend's stream through
// the model's communicator (moshi_hot_depth_shard_rccl_init), or a caller-supplied function over host memory (set_transport; CPU device / gloo tests)
typedef void (*moshi_hot_allreduce_t)(void * user, float * data, int64_t n_floats);
GGML_API void    moshi_hot_tp_set_transport(moshi_hot_model_t * m, moshi_hot_allreduce_t fn, void * user);
GGML_API void    moshi_hot_tp_stack(moshi_hot_model_t * m, const float * x, float * out);
GGML_API int64_t moshi_hot_tp_reductions(moshi_hot_model_t * m);
// The tensor-parallel stack as a FRAME mode (bench.py --shard temporal): on rank 0, after _install, the Temporal half of every moshi_hot_lm_step* is
//   embedding-sum graph -> broadcast of x (F32[dim] + a "more frames" flag; ncclBroadcast on the backend's stream, or the function of
//   moshi_hot_depth_shard_set_transport) -> the 2 L + 1 segments with their 2 L all-reduces -> out_norm / text head / sampler graph,
// and the Depth graph follows on rank 0 as in any LM step (lm.h:555-607, 659-677 around transformer.h:910-971). Every other rank sits in _serve: it takes the
// broadcast, runs its segments and joins the all-reduces until rank 0 calls _stop. Models are created with tp_world = N (>= 1), tp_rank, chain_depth = 0.
GGML_API void    moshi_hot_tp_install(moshi_hot_model_t * m);
GGML_API int64_t moshi_hot_tp_serve(moshi_hot_model_t * m);     // returns the frames served
GGML_API void    moshi_hot_tp_stop(moshi_hot_model_t * m);
GGML_API int64_t moshi_hot_tp_frames(moshi_hot_model_t * m);    // stack passes this rank has run in frame mode
GGML_API void    moshi_hot_tp_msg_read(moshi_hot_model_t * m, float * out);        // the partial-sum message, host copy out / in: lets ONE process sum the
GGML_API void    moshi_hot_tp_msg_write(moshi_hot_model_t * m, const float * in);   // partials of several ranks' models (tests without a second GPU)
// replaces the local chained Depth graph inside moshi_hot_lm_step_n: fn(user, text_token, audio[dep_q]) must fill all dep_q tokens
typedef void (*moshi_hot_depth_hook_t)(void * user, int32_t text_token, int32_t * audio);
GGML_API void    moshi_hot_set_depth_hook(moshi_hot_model_t * m, moshi_hot_depth_hook_t fn, void * user);

#ifdef __cplusplus
}
#endif
