// hip_common.h — shared declarations of the MI355X (gfx950) backend: tensor descriptors passed by
// value to kernels, and the host-side launch wrappers the planner (hip_backend.hip) calls.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ggml_impl.h"

#define HIP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    GGML_ABORT("HIP error %s at %s: %s", hipGetErrorName(e_), #expr, hipGetErrorString(e_)); } while (0)

// strided tensor view as seen by a kernel
struct tdesc {
    char *  data;
    int64_t ne[4];
    int64_t nb[4];
    int     type;
};

static inline tdesc make_tdesc(const struct ggml_tensor * t) {
    tdesc d;
    d.data = (char *) t->data;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = (int64_t) t->nb[i]; }
    d.type = (int) t->type;
    return d;
}

static inline int64_t td_nelements(const tdesc & t) { return t.ne[0] * t.ne[1] * t.ne[2] * t.ne[3]; }

// ---- generic (one kernel per ggml node) ------------------------------------------------------------
void k_binary(hipStream_t s, int op, tdesc dst, tdesc a, tdesc b);
void k_unary(hipStream_t s, int uop, tdesc dst, tdesc a);
void k_scale(hipStream_t s, tdesc dst, tdesc a, float scale, float bias);
void k_clamp(hipStream_t s, tdesc dst, tdesc a, float mn, float mx);
void k_cpy(hipStream_t s, tdesc dst, tdesc src);                 // type converting, logical order
void k_concat(hipStream_t s, tdesc dst, tdesc a, tdesc b, int dim);
void k_repeat(hipStream_t s, tdesc dst, tdesc a);
void k_pad(hipStream_t s, tdesc dst, tdesc a);
void k_arange(hipStream_t s, tdesc dst, float start, float step);
void k_sum_rows(hipStream_t s, tdesc dst, tdesc a);              // also used for ggml_sum on a flattened view
void k_sum_all(hipStream_t s, tdesc dst, tdesc a);
void k_argmax(hipStream_t s, tdesc dst, tdesc a);
void k_argsort(hipStream_t s, tdesc dst, tdesc a, int desc);
void k_norm(hipStream_t s, tdesc dst, tdesc a, float eps, int rms);
void k_norm_affine(hipStream_t s, tdesc a, float eps, int rms, const float * w, const float * b, float * out);   // one row: out = norm(a) * w (+ b), the three launches' float operations
void k_soft_max(hipStream_t s, tdesc dst, tdesc a, tdesc mask, int has_mask, float scale);
void k_get_rows(hipStream_t s, tdesc dst, tdesc a, tdesc idx);
void k_set_rows(hipStream_t s, tdesc dst, tdesc src, tdesc idx);
void k_im2col(hipStream_t s, tdesc dst, tdesc x, int64_t K, int s0, int p0, int d0);
size_t k_conv_transpose_1d_ws_size(const struct ggml_tensor * w, const struct ggml_tensor * x);
void k_conv_transpose_1d(hipStream_t s, tdesc dst, tdesc w, tdesc x, int s0, void * ws);
void k_timestep_embedding(hipStream_t s, tdesc dst, tdesc ts, int dim, int max_period, const float * addend = nullptr, int addend_n = 0);
// generic matrix product: activation rows are first converted to the weight type's dot type
// (q8_K / q8_0 / f16 / bf16) into `ws` (size from k_mul_mat_ws_size), then dotted
size_t k_mul_mat_ws_size(const struct ggml_tensor * a, const struct ggml_tensor * b);
// optional fused epilogue of the dense product: dst[m, n] = residual[m, n] + (dot + bias[n])  (ggml order: add(y, bias) then add(u, y))
//   side job of an extra workgroup (streaming conv): prev <- last TP samples of concat(prev, act(x)), which the preceding im2col (or the
//   producer's scatter, conv_scatter) has finished reading (moshi_streaming_conv_1d's tail copy, conv.h:60-75)
// The NEXT convolution's im2col panel filled by the launch that produces its input (no im2col launch between the two): every output element (position l,
// channel c) of the producing launch also lands, as f16(act(value)), at panel[ow * K + c * Kw + kk] for each (ow, kk) with ow * s0 + kk == l + TP - what
// stream_im2col_kernel would have written from the finished tensor - and the producing launch's extra workgroup writes the columns that come from the
// consumer's carried tail (ow * s0 + kk < TP: f16(prev[..]), the tail as the consumer's previous launch left it). panel == nullptr: nothing to do.
struct conv_scatter { uint16_t * panel; const float * prev; int K, Kw, s0, TP, M, C, elu, pad; };
struct mm_epilogue { const float * bias; const char * residual; int64_t res_nb0, res_nb1;
                     float * tail_prev; int tail_TP, tail_pre_elu, tail_L, tail_C; const char * tail_x; int64_t tail_nb0, tail_nb1;
                     conv_scatter sc;
                     // a 1-tap convolution over a few positions whose input no conv launch produces (the RVQ projections): the few-row product converts its
                     // activation rows from the F32 tensor itself while it stages them in LDS (af_x != nullptr: element (row m, k) at af_x + m * af_nb0 + k * af_nb1,
                     // through the ELU if af_elu) - the F16 values the im2col launch would have stored - and that launch is not planned
                     const char * af_x; int64_t af_nb0, af_nb1; int af_elu, af_pad; };
void k_mul_mat(hipStream_t s, tdesc dst, tdesc a, tdesc b, void * ws, const mm_epilogue * epi = nullptr);
bool k_mul_mat_is_few_rows(tdesc a, tdesc b);   // true: k_mul_mat runs this product as the few-row kernel (the only form that takes mm_epilogue::af_x)
// streaming conv1d helpers (moshi_streaming_conv_1d, conv.h:50-96): F16 im2col straight from (carried tail, new samples) with an
// optional ELU on the new samples, and the tail update
void k_stream_im2col(hipStream_t s, tdesc dst, const float * prev, int TP, tdesc x, int Kw, int s0, int pre_elu);
void k_conv_tail(hipStream_t s, float * prev, int TP, tdesc x, int pre_elu);
// streaming conv_transpose_1d tail (conv.h:282-309): overlap-add with the carried partial, state update, bias, window
void k_convtr_finish(hipStream_t s, tdesc out, float * prev, const float * bias, const void * ws, int K, int OC, int L, int s0, int nsplit, const conv_scatter * sc = nullptr);
// depthwise transposed conv of one input frame + streaming tail: y[k, c] = x[c] * w[k, c]
void k_dw_convtr_frame(hipStream_t s, float * out, float * prev, const float * bias, const char * x, int64_t x_cs, const char * w, int64_t w_cs, int K, int PT, int C, int64_t out_cs = -1, int64_t out_ks = 1);   // out[c * out_cs + k * out_ks] (default: out_cs = K - PT)
int  k_conv_transpose_1d_partial(hipStream_t s, tdesc w, tdesc x, void * ws, int pre_elu);   // returns the number of ic splits written to ws
// scatter of batched small uploads: descs/blob live in pinned host memory mapped into the device
struct upload_desc { char * dst; uint32_t offset; uint32_t size; };
void k_scatter_uploads(hipStream_t s, const upload_desc * descs, const char * blob, int n);

// ---- fused hot-path kernels --------------------------------------------------------------------------
// y = W x (+ residual), x produced on the fly by an optional prologue, for one activation row (T = 1)
enum mv_prologue { MV_PLAIN = 0, MV_RMSNORM = 1, MV_GATE_SILU = 2, MV_LAYERNORM = 3, MV_GELU = 4, MV_PREQ8K = 5, MV_ATTN = 6 };
struct attn_args;
// one (scaled) embedding row: dequant(table[*index]) * *scale
struct embed_src { const char * table; int64_t row_bytes; int64_t n_rows; int type; const int32_t * index; const float * scale; };
#define MV_MAX_COLS 4
struct mv_args {
    int         wtype;          // ggml_type of W
    const char * w;             // [K, M] rows of `row_bytes`
    int64_t     row_bytes;
    int64_t     K, M;
    int         prologue;
    const float * x;            // PLAIN: x[K]; RMSNORM: raw x[K]; GATE_SILU: h[2K] (left|right halves)
    const float * alpha;        // RMSNORM: alpha[K]; LAYERNORM: weight[K]
    const float * beta;         // LAYERNORM: bias[K] or NULL
    float       eps;
    int         ncols;          // activation columns (1 for quantised weights, <= MV_MAX_COLS otherwise)
    int64_t     x_cs, y_cs, r_cs;   // column strides of x / y / residual in floats
    int         out_act;        // 0 none, 1 gelu (ggml's F16-table gelu) applied to W x first
    const float * out_scale;    // optional per-row scale applied to W x before the residual (layer_scale)
    const float * residual;     // optional, [M, ncols]
    float *     y;              // [M, ncols]
    float *     x_out;          // optional: prologue result written by block 0 (keeps the ggml node materialised)
    embed_src   res_embed;      // optional (table != NULL, instead of `residual`): y = W x + one embedding row (Q4_K path)
    int32_t *   argmax_out[2];  // optional (Q4_K path): index of the last maximum of y (ggml_vec_argmax_f32), written by the last workgroup to finish
    unsigned *  ticket;         //   ... arrival counter for that (zero between launches)
    const attn_args * attn;     // MV_ATTN: (host pointer) the attention whose output is x; short ring, recomputed per workgroup
    // gated FFN, "paired" form (moshi_activation_gating, gating.h:11-37): W is linear_in [K, 2 F]; a workgroup takes rows_per_wg / 2 rows of the
    // left half AND the same rows of the right half, so it can finish g = silu(W_l x) * (W_r x) itself: y receives g[F] (M stays 2 F). The [2 F]
    // intermediate and the separate gate kernel disappear.
    int64_t     pair_F;         // 0 = off
    // Entries of a launch-ordered run that are NOT mat-vecs but may become phases of a persistent step program (hip_chain.hip): special = 1 - a single-token
    // attention step (`attn`, host pointer owned by the plan); special = 2 - one embedding row through a small Q8_0 projection (`lr`). The generic chain analysis
    // stops at them; launched on their own they run k_attn_decode / k_lowrank_embed. special = 3 - a top-k sampler (`smp`: moshi_sample_token with temp > 0,
    // sampling.h:4-64) behind a logits mat-vec: the Depth step program takes it as the tail of its linears[k] phase (hip_chain_nest.h), k_sample_topk otherwise.
    int         special;
    const struct lowrank_embed_args * lr;
    const struct sample_args * smp;
};
// Persistent chain engine (hip_chain.hip): a run of consecutive, dependent small Q4_K mat-vecs executed by ONE launch - resident workgroups,
// a loader wave streaming every phase's weights through an LDS ring ahead of the dependency chain, data-tagged hand-offs between phases.
struct chain_plan;
bool   k_chain_default_on();                         // MI355X_CHAIN (default 1; 0 = one launch per mat-vec)
// usable_cus: the compute units the launching stream may use (the device's count, or fewer on a CU-masked stream): the engine's workgroups wait for each
// other, so a chain is only taken when its whole grid is resident there (hipOccupancyMaxActiveBlocksPerMultiprocessor x usable_cus >= grid; 256 / 128 / 64)
int    k_chain_accept(const mv_args * mv, int n, int usable_cus, bool allow_step_program = true);     // how many of the n consecutive mat-vecs (in launch order) one chain launch can take (0: none)
size_t k_chain_ws_size(const mv_args * mv, int n, int usable_cus, bool allow_step_program = true);    // device workspace for exactly that run (tables + hand-off buffers)
// allow_step_program: a run that is the Depth transformer's steps at their known shapes may execute as the compile-time step program (hip_chain_nest.h)
chain_plan * k_chain_create(hipStream_t s, const mv_args * mv, int n, void * ws, unsigned * err, int usable_cus, bool allow_step_program);
bool   k_chain_is_step_program(const chain_plan * c);
void   k_chain_launch(hipStream_t s, const chain_plan * c);
void   k_chain_free(chain_plan * c);
int    k_chain_length(const chain_plan * c);
int64_t k_chain_weight_bytes(const chain_plan * c);
bool k_matvec_supported(int wtype, int64_t K, int64_t M);
bool k_matvec_pair_ok(int wtype, int64_t K, int64_t F);
// optional per-launch timing of the dominant kernel (matvec_q4k_kernel) with HIP start/stop events that are
// attached to the dispatch itself (hipExtLaunchKernelGGL), i.e. kernel begin -> kernel end like a profiler
struct mv_profile {
    struct rec { hipEvent_t start, stop; int64_t bytes; int variant; };   // variant: 0 = LDS-staged tiles (the large matrices), 1 = register streaming (WS = 1: the small ones), 2 = inproj_attn_kernel (in_proj tiles + the attention in one launch)
    rec * recs; int capacity; int used;
};
void k_matvec_set_profile(mv_profile * p);
void k_matvec(hipStream_t s, const mv_args & a);
// gated-FFN activation silu(h[:K]) * h[K:] quantised to padded Q8_K blocks (K/256 x 304 B) for a following MV_PREQ8K mat-vec
#define MV_XBLK_BYTES 304
void k_gate_quant_q8k(hipStream_t s, const float * h, int64_t K, void * out_blocks, int wtype);   // activation format follows the weight type
// alpha * rms_norm(x) quantised once to the same padded blocks (K <= 4096); n_out (optional) receives the normed floats
void k_norm_quant_q8k(hipStream_t s, const float * x, const float * alpha, float eps, int64_t K, void * out_blocks, int wtype, float * n_out);
// batched Q4_K / Q8_0 / Q4_0 mat-mul for prompt prefill (T = 2..64 activation rows): rows quantised to Q8_K (Q4_K weights) or Q8_0 into `ws`, then 16x16x32 int8 MFMA tiles
size_t k_mm_q4k_batched_ws_size(int64_t K, int64_t T);
bool k_mm_q4k_batched_supported(int wtype, int64_t K, int64_t M, int64_t T);
// prologue: MV_PLAIN (x[K, T] as is), MV_RMSNORM (alpha * rms_norm(x), eps) or MV_GATE_SILU (x = h[2K, T]: silu(h[:K]) * h[K:]); residual optional
void k_mm_q4k_batched(hipStream_t s, int wtype, const char * w, int64_t row_bytes, int64_t K, int64_t M, int64_t T, const float * x, int64_t x_cs,
                      void * ws, float * y, int64_t y_cs, const float * residual, int64_t r_cs, int prologue = 0, const float * alpha = nullptr, float eps = 0.f);

// streaming self-attention over a ring KV cache (T <= 4 new tokens): RoPE(q,k) -> cache write -> masked
// softmax(K q) V restricted to un-masked slots; see hip_kernels_fused.hip
struct attn_args {
    const float * q; const float * k; const float * v;   // element (d, t, h) at base[d + t*ts + h*hs], F32 (slices of the in_proj output)
    int64_t q_ts, q_hs, k_ts, k_hs, v_ts, v_hs;          // strides in floats
    const float * rot;          // timestep embedding [D, T]: cos(D/2) | sin(D/2) per row, or NULL (no RoPE)
    const float * mask;         // [C, T] additive mask (0 / -inf)
    const int32_t * index;      // [T] ring slots to write
    char * kcache; char * vcache;   // BF16 [D, C, H]
    int64_t k_nb1, k_nb2, v_nb1, v_nb2;
    int H, D, C, T;
    float scale;
    float * out;                // element (d, h, t) at out[t*out_ts + h*D + d]
    int64_t out_ts;
    // blocks of T > 4 new rows (batched prompt prefill): n_groups = ceil(T / 4) workgroups per head (blockIdx.y), each taking rows
    // 4 g .. 4 g + 3. Launched twice: write_only = 1 puts every row's K / V into the ring, then all groups attend concurrently
    // (their own rows come from registers, earlier groups' rows from the ring, later rows are masked).
    int n_groups, write_only;
    // 2 <= T <= 4 new rows (the codec transformers step two 25 Hz frames per call): row_split = 1 gives every query row its own workgroup (blockIdx.y = t).
    // Every workgroup rotates ALL T new rows (its scores of a later row need the earlier rows' K / V, which it takes from its own LDS, never from the
    // ring); only the workgroup of row 0 writes the ring. The two rows' score -> soft_max -> P x V chains then run side by side instead of back to back.
    int row_split;
};
// Long rings (C >= ATTN_SPLIT_MIN_C, T = 1) are split over ceil(C / ATTN_SPLIT_SLOTS) workgroups per head; `ws` (zeroed once,
// k_attn_decode_ws_size bytes) carries scores, partial outputs and the per-head arrival counters between them. ws may be NULL
// for short rings.
#define ATTN_SPLIT_MIN_C 1024
#define ATTN_SPLIT_SLOTS 128      // ring slots per workgroup once a head is split ...
#define ATTN_SPLIT_BIG_MIN 1024   // ... doubled when more than this many slots are live (bench sweeps at 230 / 600 / 2900 live slots)
#define ATTN_SINGLE_MAX 384       // up to this many live slots the head's first workgroup does everything alone (round 5 sweep, profiles/r05_sweep_attn_single_max.txt: 160 -> 384 is -85 / -50 us of Temporal at 190 / 270 live slots, neutral from 360 on; 768 loses at 510)
size_t k_attn_decode_ws_size(const attn_args & a);
bool k_attn_split_resident(const attn_args & a, int usable_cus);   // may a head be split over workgroups that wait for each other on this many compute units?
void k_attn_decode(hipStream_t s, const attn_args & a, void * ws = nullptr, unsigned * err = nullptr);   // *err <- 1 if a head-wide wait timed out

// in_proj + the attention that consumes it as ONE launch of 256 resident workgroups (inproj_attn_kernel): `a` is the RMS-normed Q4_K mat-vec whose output
// holds `at`'s q | k | v. supported(): shapes, and whether the whole grid fits the compute units the stream may use (the parts of a head wait for each
// other). ws: k_inproj_attn_ws_size bytes, zeroed once.
bool   k_inproj_attn_supported(const mv_args & a, const attn_args & at, int usable_cus);
size_t k_inproj_attn_ws_size(const mv_args & a, const attn_args & at);
void   k_inproj_attn(hipStream_t s, const mv_args & a, const attn_args & at, void * ws, unsigned * err);

// single-token cross-attention over cached F32 K / V [D, Tc, H] without a mask (moshi_streaming_multihead_cross_attention,
// transformer.h:714-762): scores = K q (float products, double sums), soft_max(scale * s), out = sum_t p_t V_t; one workgroup per head
struct xattn_args { const float * q; const char * k; const char * v; int64_t k_nb1, k_nb2, v_nb1, v_nb2; int H, D, Tc; float scale; float * out; };
void k_cross_attn(hipStream_t s, const xattn_args & a);

// sum of (scaled) embedding rows, left-to-right
#define EMBED_SUM_MAX 40      // terms of one fused embedding sum (tts: 32 audio codebooks + the demuxed text pair); the kernel is instantiated for 24 and 40
struct embed_sum_args { embed_src src[EMBED_SUM_MAX]; int n; int64_t K; float * out; };
void k_embed_sum(hipStream_t s, const embed_sum_args & a);
// one residual-VQ encode level (core_vq.h:27-56, 171-194): nearest centroid of `resid`, its index, and resid - centroid
struct vq_level_args {
    const char * emb; int64_t emb_row_bytes; int D, NC;      // F32 codebook [D, NC]
    const char * resid; int64_t resid_stride;                // element j at resid + j * resid_stride
    const float * add_c; const float * num;                  // score = num[c] / (dist + add_c[0])
    float * resid_out; float * idx_f; int32_t * idx_i;        // resid_out may be NULL (last level)
    float * cand_val; int32_t * cand_idx; unsigned * counter; // workspace: one candidate per workgroup + arrival counter (zeroed once)
};
// dst[i] = (dst type) *src[i]: a tree of concats of one-element F32 tensors collapsed into one launch (RVQ code vectors)
#define GATHER_MAX 32
struct gather_args { const float * src[GATHER_MAX]; int n; void * dst; int dst_type; };
void k_gather_scalars(hipStream_t s, const gather_args & a);
// moshi_sample_token with temp > 0 (sampling.h:4-64): soft_max(logits / temp) -> top-k (descending, ties by lower index) -> p / Exp(1) noise ->
// argmax -> the chosen index, as one launch (the node chain costs ~12 launches and a full argsort of up to 32 000 values)
#define SAMPLE_MAX_N 32768
#define SAMPLE_MAX_K 256
// get_rows(table, index) -> mul_mat(W: Q8_0 [K, M], .) [-> cast to F32]: one embedding row through a small projection (lowrank_embed_kernel)
struct lowrank_embed_args { const char * table; int64_t row_bytes, n_rows; int type; const int32_t * index; const char * w; int64_t w_row_bytes; int K, M; float * out; };
void k_lowrank_embed(hipStream_t s, const lowrank_embed_args & a);
struct sample_args { const float * logits; int n; float scale; int k; const float * noise; int32_t * out; int32_t * out2; };   // out2: optional copy (the token vector slot)
void k_sample_topk(hipStream_t s, const sample_args & a);
#define VQ_LEVEL_WS_BYTES 4096
void k_vq_level(hipStream_t s, const vq_level_args & a);
// Consecutive levels of one RVQ stack (each level's residual is the previous level's output: core_vq.h:27-56 inside vq.h:97-114's loop) as ONE persistent
// launch (hip_chain.hip, vq_chain_kernel): the workgroups score their centroids of level l, exchange their candidates as tagged granules, every workgroup
// merges them and subtracts the winning centroid from its own copy of the residual - one hand-off per level instead of a launch per level.
struct vq_chain_plan;
bool   k_vq_chain_accept(const vq_level_args * lv, int n, int usable_cus);      // n >= 2 chained levels of one shape whose grid is resident on this stream
size_t k_vq_chain_ws_size(int n);
vq_chain_plan * k_vq_chain_create(hipStream_t s, const vq_level_args * lv, int n, void * ws, unsigned * err);
void   k_vq_chain_launch(hipStream_t s, const vq_chain_plan * c);
void   k_vq_chain_free(vq_chain_plan * c);
