// ggml.h — C-ABI drop-in boundary, part 1 of 4 (tensors, contexts, op builders, graphs).
//
// This header is a from-scratch restatement of the subset of the ggml public C API that
// Codes4Fun/moshi.cpp binds to on its streaming-decode hot path. ggml itself is NOT vendored in the
// reference (README.md:183-199), so the surface below is reconstructed from the reference's call
// sites; every declaration cites the reference line that uses it. A libmoshi / moshi-sts / moshi-tts /
// moshi-stt build that includes <ggml.h> from here and links libggml-mi355x.so gets the MI355X HIP
// backend with no source change.
//
// Struct fields that the reference touches directly (src/context.h:138-165, src/ggml_cap.h:1431-1466):
// ggml_tensor::{ne, nb, type, op, src, view_src, view_offs, data, name, buffer, flags}.
#pragma once

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_API __attribute__((visibility("default")))

#define GGML_MAX_DIMS        4
#define GGML_MAX_SRC         10
#define GGML_MAX_OP_PARAMS   64
#define GGML_MAX_NAME        64    // src/loader.h:120-137: names >= this are replaced by a CRC
#define GGML_DEFAULT_GRAPH_SIZE 2048  // src/context.h:489 uses GGML_DEFAULT_GRAPH_SIZE * 4
#define GGML_MEM_ALIGN       16

#define GGML_PAD(x, n) (((x) + (n) - 1) & ~((n) - 1))
#define GGML_UNUSED(x) (void)(x)

#define GGML_ABORT(...) ggml_abort(__FILE__, __LINE__, __VA_ARGS__)
#define GGML_ASSERT(x) do { if (!(x)) GGML_ABORT("GGML_ASSERT(%s) failed", #x); } while (0)

GGML_API void ggml_abort(const char * file, int line, const char * fmt, ...) __attribute__((noreturn));

enum ggml_status {
    GGML_STATUS_ALLOC_FAILED = -2,
    GGML_STATUS_FAILED       = -1,
    GGML_STATUS_SUCCESS      = 0,
    GGML_STATUS_ABORTED      = 1,
};

// numeric values follow the GGUF on-disk type ids so that files written by the reference
// (src/loader.h:227-233) load here unchanged
enum ggml_type {
    GGML_TYPE_F32  = 0,
    GGML_TYPE_F16  = 1,
    GGML_TYPE_Q4_0 = 2,
    GGML_TYPE_Q4_1 = 3,
    GGML_TYPE_Q5_0 = 6,
    GGML_TYPE_Q5_1 = 7,
    GGML_TYPE_Q8_0 = 8,
    GGML_TYPE_Q8_1 = 9,
    GGML_TYPE_Q2_K = 10,
    GGML_TYPE_Q3_K = 11,
    GGML_TYPE_Q4_K = 12,
    GGML_TYPE_Q5_K = 13,
    GGML_TYPE_Q6_K = 14,
    GGML_TYPE_Q8_K = 15,
    GGML_TYPE_I8   = 24,
    GGML_TYPE_I16  = 25,
    GGML_TYPE_I32  = 26,
    GGML_TYPE_I64  = 27,
    GGML_TYPE_F64  = 28,
    GGML_TYPE_BF16 = 30,
    GGML_TYPE_COUNT = 40,
};

typedef uint16_t ggml_fp16_t;
typedef struct { uint16_t bits; } ggml_bf16_t;

enum ggml_op {
    GGML_OP_NONE = 0,
    GGML_OP_DUP,
    GGML_OP_ADD,
    GGML_OP_SUB,
    GGML_OP_MUL,
    GGML_OP_DIV,
    GGML_OP_SCALE,
    GGML_OP_SUM,
    GGML_OP_SUM_ROWS,
    GGML_OP_ARGMAX,
    GGML_OP_REPEAT,
    GGML_OP_CONCAT,
    GGML_OP_NORM,
    GGML_OP_RMS_NORM,
    GGML_OP_MUL_MAT,
    GGML_OP_CPY,
    GGML_OP_CONT,
    GGML_OP_RESHAPE,
    GGML_OP_VIEW,
    GGML_OP_PERMUTE,
    GGML_OP_TRANSPOSE,
    GGML_OP_GET_ROWS,
    GGML_OP_SET_ROWS,
    GGML_OP_SOFT_MAX,
    GGML_OP_CLAMP,
    GGML_OP_CONV_TRANSPOSE_1D,
    GGML_OP_IM2COL,
    GGML_OP_PAD,
    GGML_OP_ARANGE,
    GGML_OP_TIMESTEP_EMBEDDING,
    GGML_OP_ARGSORT,
    GGML_OP_TOP_K,
    GGML_OP_UNARY,
    GGML_OP_COUNT,
};

enum ggml_unary_op {
    GGML_UNARY_OP_NEG = 0,
    GGML_UNARY_OP_ELU,
    GGML_UNARY_OP_GELU,
    GGML_UNARY_OP_SILU,
    GGML_UNARY_OP_RELU,
    GGML_UNARY_OP_TANH,
    GGML_UNARY_OP_SIGMOID,
    GGML_UNARY_OP_EXP,
    GGML_UNARY_OP_COUNT,
};

enum ggml_sort_order {
    GGML_SORT_ORDER_ASC,
    GGML_SORT_ORDER_DESC,
};

enum ggml_tensor_flag {
    GGML_TENSOR_FLAG_INPUT  = 1,
    GGML_TENSOR_FLAG_OUTPUT = 2,
    GGML_TENSOR_FLAG_PARAM  = 4,
};

struct ggml_context;
struct ggml_cgraph;
struct ggml_backend_buffer;

// n-dimensional tensor; ne[0] is the fastest dimension, nb[] are byte strides
struct ggml_tensor {
    enum ggml_type type;
    struct ggml_backend_buffer * buffer;
    int64_t ne[GGML_MAX_DIMS];
    size_t  nb[GGML_MAX_DIMS];
    enum ggml_op op;
    int32_t op_params[GGML_MAX_OP_PARAMS / sizeof(int32_t)];
    int32_t flags;
    struct ggml_tensor * src[GGML_MAX_SRC];
    struct ggml_tensor * view_src;   // root tensor when this is a view
    size_t               view_offs;  // byte offset into view_src (may wrap: negative offsets, transformer.h:200-209)
    void * data;
    char name[GGML_MAX_NAME];
    void * extra;
    char padding[8];
};

// brace-initialised by the reference: ggml_init({mem_size, NULL, no_alloc}) (src/context.h:281-285)
struct ggml_init_params {
    size_t mem_size;
    void * mem_buffer;
    bool   no_alloc;
};

// ---- misc ----------------------------------------------------------------------------------------
GGML_API int64_t ggml_time_ms(void);   // tools/moshi-tts.cpp:749
GGML_API int64_t ggml_time_us(void);   // tools/moshi-sts.cpp:731
GGML_API void    ggml_time_init(void);

GGML_API const char * ggml_type_name(enum ggml_type type);     // src/replay_ops.h:32-65
GGML_API const char * ggml_op_name  (enum ggml_op op);
GGML_API const char * ggml_op_desc  (const struct ggml_tensor * t);
GGML_API const char * ggml_status_to_string(enum ggml_status status);
GGML_API int64_t ggml_blck_size(enum ggml_type type);
GGML_API size_t  ggml_type_size(enum ggml_type type);
GGML_API size_t  ggml_row_size (enum ggml_type type, int64_t ne);   // src/loader.h:286
GGML_API bool    ggml_is_quantized(enum ggml_type type);

GGML_API int64_t ggml_nelements(const struct ggml_tensor * tensor);  // src/context.h:184
GGML_API int64_t ggml_nrows    (const struct ggml_tensor * tensor);
GGML_API size_t  ggml_nbytes   (const struct ggml_tensor * tensor);  // src/context.h:138
GGML_API size_t  ggml_element_size(const struct ggml_tensor * tensor);
GGML_API int     ggml_n_dims   (const struct ggml_tensor * tensor);
GGML_API bool    ggml_is_contiguous(const struct ggml_tensor * tensor);
GGML_API bool    ggml_is_transposed(const struct ggml_tensor * tensor);
GGML_API bool    ggml_is_permuted  (const struct ggml_tensor * tensor);
GGML_API bool    ggml_are_same_shape(const struct ggml_tensor * a, const struct ggml_tensor * b);

GGML_API float       ggml_fp16_to_fp32(ggml_fp16_t x);
GGML_API ggml_fp16_t ggml_fp32_to_fp16(float x);
GGML_API float       ggml_bf16_to_fp32(ggml_bf16_t x);
GGML_API ggml_bf16_t ggml_fp32_to_bf16(float x);

// ---- contexts ------------------------------------------------------------------------------------
GGML_API struct ggml_context * ggml_init (struct ggml_init_params params);  // src/context.h:281
GGML_API void                  ggml_reset(struct ggml_context * ctx);       // src/context.h:623
GGML_API void                  ggml_free (struct ggml_context * ctx);       // src/context.h:293
GGML_API size_t ggml_used_mem      (const struct ggml_context * ctx);
GGML_API bool   ggml_get_no_alloc  (struct ggml_context * ctx);
GGML_API void   ggml_set_no_alloc  (struct ggml_context * ctx, bool no_alloc);
GGML_API size_t ggml_tensor_overhead(void);   // src/context.h:206, 750
GGML_API size_t ggml_graph_overhead (void);
GGML_API size_t ggml_graph_overhead_custom(size_t size, bool grads);

GGML_API struct ggml_tensor * ggml_new_tensor   (struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne); // src/context.h:211
GGML_API struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0);                    // src/context.h:331
GGML_API struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1);       // compression.h:160
GGML_API struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2);
GGML_API struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3); // src/context.h:75
GGML_API struct ggml_tensor * ggml_dup_tensor   (struct ggml_context * ctx, const struct ggml_tensor * src);  // transformer.h:1303
GGML_API struct ggml_tensor * ggml_view_tensor  (struct ggml_context * ctx, struct ggml_tensor * src);

GGML_API struct ggml_tensor * ggml_get_first_tensor(const struct ggml_context * ctx);                         // src/loader.h:229
GGML_API struct ggml_tensor * ggml_get_next_tensor (const struct ggml_context * ctx, struct ggml_tensor * t); // src/loader.h:230
GGML_API struct ggml_tensor * ggml_get_tensor      (struct ggml_context * ctx, const char * name);            // src/loader.h:249

GGML_API const char *         ggml_get_name  (const struct ggml_tensor * tensor);
GGML_API struct ggml_tensor * ggml_set_name  (struct ggml_tensor * tensor, const char * name);  // src/loader.h:296
GGML_API struct ggml_tensor * ggml_format_name(struct ggml_tensor * tensor, const char * fmt, ...);
GGML_API void ggml_set_input (struct ggml_tensor * tensor);
GGML_API void ggml_set_output(struct ggml_tensor * tensor);   // src/ggml_cap.h (capture only)

GGML_API enum ggml_unary_op ggml_get_unary_op(const struct ggml_tensor * tensor);

// ---- op builders (each returns a new node in ctx; nothing is computed here) -----------------------
// signatures as spelled out by the reference's wrappers, src/ggml_wrap.h:106-614
GGML_API struct ggml_tensor * ggml_dup (struct ggml_context * ctx, struct ggml_tensor * a);
GGML_API struct ggml_tensor * ggml_add (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);          // transformer.h:934
GGML_API struct ggml_tensor * ggml_add_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);   // conv.h:290
// in-place variants the reference only reaches through its capture / replay tooling (src/replay_ops.h:293-308, CLASS_OP_INPLACE)
GGML_API struct ggml_tensor * ggml_sub_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
GGML_API struct ggml_tensor * ggml_mul_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
GGML_API struct ggml_tensor * ggml_div_inplace(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
GGML_API struct ggml_tensor * ggml_sub (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);          // rope.h:104
GGML_API struct ggml_tensor * ggml_mul (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);          // transformer.h:22
GGML_API struct ggml_tensor * ggml_div (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);          // sampling.h:13
GGML_API struct ggml_tensor * ggml_neg (struct ggml_context * ctx, struct ggml_tensor * a);                                  // transformer.h:276
GGML_API struct ggml_tensor * ggml_scale(struct ggml_context * ctx, struct ggml_tensor * a, float s);                        // sampling.h:58
GGML_API struct ggml_tensor * ggml_scale_inplace(struct ggml_context * ctx, struct ggml_tensor * a, float s);
GGML_API struct ggml_tensor * ggml_clamp(struct ggml_context * ctx, struct ggml_tensor * a, float min, float max);           // core_vq.h:77
GGML_API struct ggml_tensor * ggml_sum (struct ggml_context * ctx, struct ggml_tensor * a);                                  // src/context.h:508
GGML_API struct ggml_tensor * ggml_sum_rows(struct ggml_context * ctx, struct ggml_tensor * a);                              // core_vq.h:48
GGML_API struct ggml_tensor * ggml_argmax(struct ggml_context * ctx, struct ggml_tensor * a);                                // sampling.h:63
GGML_API struct ggml_tensor * ggml_argsort(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_sort_order order);
GGML_API struct ggml_tensor * ggml_argsort_top_k(struct ggml_context * ctx, struct ggml_tensor * a, int k);                  // sampling.h:35
GGML_API struct ggml_tensor * ggml_top_k (struct ggml_context * ctx, struct ggml_tensor * a, int k);                         // src/ggml_wrap.h:580
GGML_API struct ggml_tensor * ggml_arange(struct ggml_context * ctx, float start, float stop, float step);                   // src/context.h:452
GGML_API struct ggml_tensor * ggml_repeat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);
GGML_API struct ggml_tensor * ggml_repeat_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3); // core_vq.h:40
GGML_API struct ggml_tensor * ggml_concat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, int dim); // rope.h:123
GGML_API struct ggml_tensor * ggml_pad   (struct ggml_context * ctx, struct ggml_tensor * a, int p0, int p1, int p2, int p3);  // conv.h:25

GGML_API struct ggml_tensor * ggml_silu(struct ggml_context * ctx, struct ggml_tensor * a);   // gating.h:32
GGML_API struct ggml_tensor * ggml_gelu(struct ggml_context * ctx, struct ggml_tensor * a);   // transformer.h:957
GGML_API struct ggml_tensor * ggml_elu (struct ggml_context * ctx, struct ggml_tensor * a);   // seanet.h:20

GGML_API struct ggml_tensor * ggml_norm    (struct ggml_context * ctx, struct ggml_tensor * a, float eps);   // torch.h:55
GGML_API struct ggml_tensor * ggml_rms_norm(struct ggml_context * ctx, struct ggml_tensor * a, float eps);   // transformer.h:21

// a:[K,M,..] b:[K,N,..] -> [M,N,..] (torch.h:83, 232, 235)
GGML_API struct ggml_tensor * ggml_mul_mat(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);

GGML_API struct ggml_tensor * ggml_soft_max    (struct ggml_context * ctx, struct ggml_tensor * a);          // sampling.h:59
GGML_API struct ggml_tensor * ggml_soft_max_ext(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * mask, float scale, float max_bias); // torch.h:233

GGML_API struct ggml_tensor * ggml_cast(struct ggml_context * ctx, struct ggml_tensor * a, enum ggml_type type);   // lm.h:466
GGML_API struct ggml_tensor * ggml_cpy (struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b); // lm.h:510 (result aliases b)
GGML_API struct ggml_tensor * ggml_cont(struct ggml_context * ctx, struct ggml_tensor * a);                         // torch.h:221

GGML_API struct ggml_tensor * ggml_reshape_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0);
GGML_API struct ggml_tensor * ggml_reshape_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1);                           // sampling.h:9
GGML_API struct ggml_tensor * ggml_reshape_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2);              // transformer.h:573
GGML_API struct ggml_tensor * ggml_reshape_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3); // transformer.h:516

GGML_API struct ggml_tensor * ggml_view_1d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, size_t offset);                                              // lm.h:509
GGML_API struct ggml_tensor * ggml_view_2d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset);                      // torch.h:110
GGML_API struct ggml_tensor * ggml_view_3d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset); // transformer.h:498
GGML_API struct ggml_tensor * ggml_view_4d(struct ggml_context * ctx, struct ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset); // gating.h:18

// axis i of the source becomes axis ax_i of the result (transformer.h:522, rope.h:64)
GGML_API struct ggml_tensor * ggml_permute  (struct ggml_context * ctx, struct ggml_tensor * a, int axis0, int axis1, int axis2, int axis3);
GGML_API struct ggml_tensor * ggml_transpose(struct ggml_context * ctx, struct ggml_tensor * a);   // torch.h:234

GGML_API struct ggml_tensor * ggml_get_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b);                          // lm_utils.h:165
GGML_API struct ggml_tensor * ggml_set_rows(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, struct ggml_tensor * c);  // transformer.h:246 (result aliases a)

GGML_API struct ggml_tensor * ggml_im2col(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b,
                                          int s0, int s1, int p0, int p1, int d0, int d1, bool is_2D, enum ggml_type dst_type);
GGML_API struct ggml_tensor * ggml_conv_1d(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, int s0, int p0, int d0);            // conv.h:81
GGML_API struct ggml_tensor * ggml_conv_transpose_1d(struct ggml_context * ctx, struct ggml_tensor * a, struct ggml_tensor * b, int s0, int p0, int d0);  // conv.h:260

GGML_API struct ggml_tensor * ggml_timestep_embedding(struct ggml_context * ctx, struct ggml_tensor * timesteps, int dim, int max_period);  // rope.h:17

// ---- graphs --------------------------------------------------------------------------------------
GGML_API struct ggml_cgraph * ggml_new_graph       (struct ggml_context * ctx);
GGML_API struct ggml_cgraph * ggml_new_graph_custom(struct ggml_context * ctx, size_t size, bool grads);   // src/context.h:489
GGML_API void ggml_build_forward_expand(struct ggml_cgraph * cgraph, struct ggml_tensor * tensor);         // src/context.h:494
GGML_API void ggml_graph_clear(struct ggml_cgraph * cgraph);
GGML_API int  ggml_graph_size   (struct ggml_cgraph * cgraph);
GGML_API int  ggml_graph_n_nodes(struct ggml_cgraph * cgraph);
GGML_API struct ggml_tensor *  ggml_graph_node (struct ggml_cgraph * cgraph, int i);   // i < 0 counts from the end
GGML_API struct ggml_tensor ** ggml_graph_nodes(struct ggml_cgraph * cgraph);
GGML_API int  ggml_graph_n_leafs(struct ggml_cgraph * cgraph);
GGML_API struct ggml_tensor *  ggml_graph_leaf (struct ggml_cgraph * cgraph, int i);
GGML_API void ggml_graph_print(const struct ggml_cgraph * cgraph);

// ---- block-quantised storage formats ([ggml-upstream]) -------------------------------
#define QK4_0 32
#define QK8_0 32
#define QK_K  256
#define K_SCALE_SIZE 12

typedef struct { ggml_fp16_t d; uint8_t qs[QK4_0 / 2]; } block_q4_0;                       // 18 B / 32 w
typedef struct { ggml_fp16_t d; int8_t  qs[QK8_0];     } block_q8_0;                       // 34 B / 32 w
typedef struct { ggml_fp16_t d; ggml_fp16_t dmin; uint8_t scales[K_SCALE_SIZE]; uint8_t qs[QK_K / 2]; } block_q4_K;  // 144 B / 256 w
typedef struct { float d; int8_t qs[QK_K]; int16_t bsums[QK_K / 16]; } block_q8_K;          // 292 B / 256 w

// row (de)quantisation helpers: host-side, used by loaders and tests
GGML_API void ggml_quantize_row(enum ggml_type type, const float * x, void * y, int64_t k);
GGML_API void ggml_dequantize_row(enum ggml_type type, const void * x, float * y, int64_t k);

#ifdef __cplusplus
}
#endif
