// ggml_core.cpp — host-side object model behind include/ggml.h: context arenas, tensor/view
// bookkeeping, op builders (shape inference only, nothing is computed here) and graph expansion.
// Written from the call sites in the reference and the documented ggml semantics
// (SURVEY.md Appendix B); ggml itself is not available in this build environment.
#include "ggml_impl.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <time.h>

// ---------------------------------------------------------------------------------------------------
// abort / time
// ---------------------------------------------------------------------------------------------------
extern "C" void ggml_abort(const char * file, int line, const char * fmt, ...) {
    fflush(stdout);
    fprintf(stderr, "%s:%d: ", file, line);
    va_list args;
    va_start(args, fmt);
    vfprintf(stderr, fmt, args);
    va_end(args);
    fprintf(stderr, "\n");
    abort();
}

extern "C" void ggml_time_init(void) {}
extern "C" int64_t ggml_time_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t) ts.tv_sec * 1000000 + (int64_t) ts.tv_nsec / 1000;
}
extern "C" int64_t ggml_time_ms(void) { return ggml_time_us() / 1000; }

// ---------------------------------------------------------------------------------------------------
// type traits
// ---------------------------------------------------------------------------------------------------
struct type_traits_t { const char * name; int64_t blck; size_t size; bool quantized; };

static const type_traits_t * type_traits(enum ggml_type t) {
    static type_traits_t tr[GGML_TYPE_COUNT];
    static bool init = false;
    if (!init) {
        for (auto & e : tr) e = { NULL, 0, 0, false };
        tr[GGML_TYPE_F32]  = { "f32",  1, 4, false };
        tr[GGML_TYPE_F16]  = { "f16",  1, 2, false };
        tr[GGML_TYPE_BF16] = { "bf16", 1, 2, false };
        tr[GGML_TYPE_F64]  = { "f64",  1, 8, false };
        tr[GGML_TYPE_I8]   = { "i8",   1, 1, false };
        tr[GGML_TYPE_I16]  = { "i16",  1, 2, false };
        tr[GGML_TYPE_I32]  = { "i32",  1, 4, false };
        tr[GGML_TYPE_I64]  = { "i64",  1, 8, false };
        tr[GGML_TYPE_Q4_0] = { "q4_0", 32, 18, true };
        tr[GGML_TYPE_Q4_1] = { "q4_1", 32, 20, true };
        tr[GGML_TYPE_Q5_0] = { "q5_0", 32, 22, true };
        tr[GGML_TYPE_Q5_1] = { "q5_1", 32, 24, true };
        tr[GGML_TYPE_Q8_0] = { "q8_0", 32, 34, true };
        tr[GGML_TYPE_Q8_1] = { "q8_1", 32, 36, true };
        tr[GGML_TYPE_Q2_K] = { "q2_K", 256, 84, true };
        tr[GGML_TYPE_Q3_K] = { "q3_K", 256, 110, true };
        tr[GGML_TYPE_Q4_K] = { "q4_K", 256, 144, true };
        tr[GGML_TYPE_Q5_K] = { "q5_K", 256, 176, true };
        tr[GGML_TYPE_Q6_K] = { "q6_K", 256, 210, true };
        tr[GGML_TYPE_Q8_K] = { "q8_K", 256, 292, true };
        init = true;
    }
    GGML_ASSERT((int) t >= 0 && t < GGML_TYPE_COUNT);
    return &tr[t];
}

extern "C" const char * ggml_type_name(enum ggml_type type) {
    if ((int) type < 0 || type >= GGML_TYPE_COUNT) return "NONE";
    const char * n = type_traits(type)->name;
    return n ? n : "NONE";
}
extern "C" int64_t ggml_blck_size(enum ggml_type type) { return type_traits(type)->blck; }
extern "C" size_t  ggml_type_size(enum ggml_type type) { return type_traits(type)->size; }
extern "C" bool    ggml_is_quantized(enum ggml_type type) { return type_traits(type)->quantized; }
extern "C" size_t  ggml_row_size(enum ggml_type type, int64_t ne) {
    const type_traits_t * tr = type_traits(type);
    GGML_ASSERT(tr->blck > 0 && ne % tr->blck == 0);
    return tr->size * (size_t) (ne / tr->blck);
}

static const char * OP_NAMES[GGML_OP_COUNT] = {
    "NONE", "DUP", "ADD", "SUB", "MUL", "DIV", "SCALE", "SUM", "SUM_ROWS", "ARGMAX", "REPEAT", "CONCAT",
    "NORM", "RMS_NORM", "MUL_MAT", "CPY", "CONT", "RESHAPE", "VIEW", "PERMUTE", "TRANSPOSE", "GET_ROWS",
    "SET_ROWS", "SOFT_MAX", "CLAMP", "CONV_TRANSPOSE_1D", "IM2COL", "PAD", "ARANGE", "TIMESTEP_EMBEDDING",
    "ARGSORT", "TOP_K", "UNARY",
};
static const char * UNARY_NAMES[GGML_UNARY_OP_COUNT] = { "NEG", "ELU", "GELU", "SILU", "RELU", "TANH", "SIGMOID", "EXP" };

extern "C" const char * ggml_op_name(enum ggml_op op) {
    return ((int) op >= 0 && op < GGML_OP_COUNT) ? OP_NAMES[op] : "?";
}
extern "C" enum ggml_unary_op ggml_get_unary_op(const struct ggml_tensor * t) {
    GGML_ASSERT(t->op == GGML_OP_UNARY);
    return (enum ggml_unary_op) t->op_params[0];
}
extern "C" const char * ggml_op_desc(const struct ggml_tensor * t) {
    if (t->op == GGML_OP_UNARY) return UNARY_NAMES[ggml_get_unary_op(t)];
    return ggml_op_name(t->op);
}
extern "C" const char * ggml_status_to_string(enum ggml_status s) {
    switch (s) {
        case GGML_STATUS_ALLOC_FAILED: return "GGML status: error (failed to allocate memory)";
        case GGML_STATUS_FAILED:       return "GGML status: error (operation failed)";
        case GGML_STATUS_SUCCESS:      return "GGML status: success";
        case GGML_STATUS_ABORTED:      return "GGML status: warning (operation aborted)";
    }
    return "GGML status: unknown";
}

// ---------------------------------------------------------------------------------------------------
// half / bfloat conversions (bit-exact, round-to-nearest-even)
// ---------------------------------------------------------------------------------------------------
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

extern "C" float ggml_fp16_to_fp32(ggml_fp16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t man = h & 0x3ff;
    if (exp == 0) {
        if (man == 0) return bits_f32(sign);
        // subnormal: normalise
        int e = -1;
        do { e++; man <<= 1; } while ((man & 0x400) == 0);
        man &= 0x3ff;
        return bits_f32(sign | (uint32_t) (127 - 15 - e) << 23 | man << 13);
    }
    if (exp == 31) return bits_f32(sign | 0x7f800000u | man << 13);
    return bits_f32(sign | (exp + 127 - 15) << 23 | man << 13);
}

extern "C" ggml_fp16_t ggml_fp32_to_fp16(float f) {
    const uint32_t x = f32_bits(f);
    const uint32_t sign = (x >> 16) & 0x8000;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (ggml_fp16_t) (sign | 0x7c00 | (ax > 0x7f800000u ? 0x200 | ((ax >> 13) & 0x3ff) : 0));
    if (ax >= 0x477ff000u) return (ggml_fp16_t) (sign | 0x7c00);   // rounds to inf (>= 65520)
    if (ax < 0x33000001u) return (ggml_fp16_t) sign;               // rounds to zero (<= 2^-25)
    int32_t exp = (int32_t) (ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    uint32_t shift, hexp;
    if (exp < -14) { shift = (uint32_t) (13 + (-14 - exp)); hexp = 0; }
    else           { shift = 13; hexp = (uint32_t) (exp + 15); }
    uint32_t hman = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hman & 1))) hman++;
    // hman includes the implicit bit for normals; adding it to the exponent field handles carries
    uint32_t h = (hexp ? ((hexp - 1) << 10) : 0) + hman;
    return (ggml_fp16_t) (sign | h);
}

extern "C" float ggml_bf16_to_fp32(ggml_bf16_t h) { return bits_f32((uint32_t) h.bits << 16); }
extern "C" ggml_bf16_t ggml_fp32_to_bf16(float f) {
    ggml_bf16_t h;
    const uint32_t u = f32_bits(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) { h.bits = (uint16_t) ((u >> 16) | 64); return h; }  // quiet NaN
    h.bits = (uint16_t) ((u + (0x7fff + ((u >> 16) & 1))) >> 16);
    return h;
}

// ---------------------------------------------------------------------------------------------------
// contexts
// ---------------------------------------------------------------------------------------------------
enum object_type { OBJECT_TENSOR = 0, OBJECT_GRAPH = 1 };

struct ggml_object {
    size_t offs;
    size_t size;
    struct ggml_object * next;
    int    type;
    char   padding[4];
};

struct ggml_context {
    size_t mem_size;
    char * mem_buffer;
    bool   mem_buffer_owned;
    bool   no_alloc;
    int    n_objects;
    struct ggml_object * objects_begin;
    struct ggml_object * objects_end;
};

extern "C" struct ggml_context * ggml_init(struct ggml_init_params params) {
    struct ggml_context * ctx = (struct ggml_context *) calloc(1, sizeof(struct ggml_context));
    GGML_ASSERT(ctx);
    size_t mem_size = params.mem_buffer ? params.mem_size : GGML_PAD(params.mem_size, GGML_MEM_ALIGN);
    if (mem_size == 0) mem_size = GGML_MEM_ALIGN;
    ctx->mem_size = mem_size;
    if (params.mem_buffer) {
        ctx->mem_buffer = (char *) params.mem_buffer;
    } else {
        void * p = NULL;
        // pages are committed lazily, so the reference's 256 MB graph contexts cost nothing until used
        if (posix_memalign(&p, 64, mem_size) != 0) GGML_ABORT("ggml_init: failed to allocate %zu bytes", mem_size);
        ctx->mem_buffer = (char *) p;
        ctx->mem_buffer_owned = true;
    }
    ctx->no_alloc = params.no_alloc;
    return ctx;
}

extern "C" void ggml_reset(struct ggml_context * ctx) {
    if (!ctx) return;
    ctx->n_objects = 0;
    ctx->objects_begin = ctx->objects_end = NULL;
}

extern "C" void ggml_free(struct ggml_context * ctx) {
    if (!ctx) return;
    if (ctx->mem_buffer_owned) free(ctx->mem_buffer);
    free(ctx);
}

extern "C" size_t ggml_used_mem(const struct ggml_context * ctx) {
    return ctx->objects_end ? ctx->objects_end->offs + ctx->objects_end->size : 0;
}
extern "C" bool ggml_get_no_alloc(struct ggml_context * ctx) { return ctx->no_alloc; }
extern "C" void ggml_set_no_alloc(struct ggml_context * ctx, bool no_alloc) { ctx->no_alloc = no_alloc; }

extern "C" size_t ggml_tensor_overhead(void) { return sizeof(struct ggml_object) + sizeof(struct ggml_tensor); }

static struct ggml_object * new_object(struct ggml_context * ctx, int type, size_t size) {
    struct ggml_object * cur = ctx->objects_end;
    const size_t cur_end = cur ? cur->offs + cur->size : 0;
    const size_t size_needed = GGML_PAD(size, GGML_MEM_ALIGN);
    if (cur_end + size_needed + sizeof(struct ggml_object) > ctx->mem_size) {
        GGML_ABORT("not enough space in the context's memory pool (needed %zu, available %zu)",
                   cur_end + size_needed + sizeof(struct ggml_object), ctx->mem_size);
    }
    struct ggml_object * obj = (struct ggml_object *) (ctx->mem_buffer + cur_end);
    obj->offs = cur_end + sizeof(struct ggml_object);
    obj->size = size_needed;
    obj->next = NULL;
    obj->type = type;
    if (cur) cur->next = obj; else ctx->objects_begin = obj;
    ctx->objects_end = obj;
    ctx->n_objects++;
    return obj;
}

// ---------------------------------------------------------------------------------------------------
// tensor queries
// ---------------------------------------------------------------------------------------------------
extern "C" int64_t ggml_nelements(const struct ggml_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
extern "C" int64_t ggml_nrows(const struct ggml_tensor * t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
extern "C" size_t  ggml_element_size(const struct ggml_tensor * t) { return ggml_type_size(t->type); }

extern "C" size_t ggml_nbytes(const struct ggml_tensor * t) {
    for (int i = 0; i < GGML_MAX_DIMS; i++) if (t->ne[i] <= 0) return 0;
    const int64_t blck = ggml_blck_size(t->type);
    size_t nbytes;
    if (blck == 1) {
        nbytes = ggml_type_size(t->type);
        for (int i = 0; i < GGML_MAX_DIMS; i++) nbytes += (size_t) (t->ne[i] - 1) * t->nb[i];
    } else {
        nbytes = (size_t) t->ne[0] * t->nb[0] / (size_t) blck;
        for (int i = 1; i < GGML_MAX_DIMS; i++) nbytes += (size_t) (t->ne[i] - 1) * t->nb[i];
    }
    return nbytes;
}

extern "C" int ggml_n_dims(const struct ggml_tensor * t) {
    for (int i = GGML_MAX_DIMS - 1; i >= 1; --i) if (t->ne[i] > 1) return i + 1;
    return 1;
}

extern "C" bool ggml_is_contiguous(const struct ggml_tensor * t) {
    size_t next_nb = ggml_type_size(t->type);
    if (t->ne[0] != ggml_blck_size(t->type) && t->nb[0] != next_nb) return false;
    next_nb *= (size_t) (t->ne[0] / ggml_blck_size(t->type));
    for (int i = 1; i < GGML_MAX_DIMS; i++) {
        if (t->ne[i] != 1) {
            if (t->nb[i] != next_nb) return false;
            next_nb *= (size_t) t->ne[i];
        }
    }
    return true;
}
extern "C" bool ggml_is_transposed(const struct ggml_tensor * t) { return t->nb[0] > t->nb[1]; }
extern "C" bool ggml_is_permuted(const struct ggml_tensor * t) {
    return t->nb[0] > t->nb[1] || t->nb[1] > t->nb[2] || t->nb[2] > t->nb[3];
}
extern "C" bool ggml_are_same_shape(const struct ggml_tensor * a, const struct ggml_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
static bool is_matrix(const struct ggml_tensor * t) { return t->ne[2] == 1 && t->ne[3] == 1; }
static bool is_vector(const struct ggml_tensor * t) { return t->ne[1] == 1 && t->ne[2] == 1 && t->ne[3] == 1; }
static bool is_empty(const struct ggml_tensor * t) { return ggml_nelements(t) == 0; }
// b can be tiled to the shape of... t1 = n * t0 in every dim
static bool can_repeat(const struct ggml_tensor * t0, const struct ggml_tensor * t1) {
    if (is_empty(t0)) return is_empty(t1);
    return t1->ne[0] % t0->ne[0] == 0 && t1->ne[1] % t0->ne[1] == 0 && t1->ne[2] % t0->ne[2] == 0 && t1->ne[3] % t0->ne[3] == 0;
}
static bool rows_contiguous(const struct ggml_tensor * t) {
    return t->ne[0] == ggml_blck_size(t->type) || t->nb[0] == ggml_type_size(t->type);
}

// ---------------------------------------------------------------------------------------------------
// tensor creation
// ---------------------------------------------------------------------------------------------------
static struct ggml_tensor * new_tensor_impl(struct ggml_context * ctx, enum ggml_type type, int n_dims,
                                            const int64_t * ne, struct ggml_tensor * view_src, size_t view_offs) {
    GGML_ASSERT(type_traits(type)->blck > 0);
    GGML_ASSERT(n_dims >= 1 && n_dims <= GGML_MAX_DIMS);
    if (view_src != NULL && view_src->view_src != NULL) {   // resolve to the root
        view_offs += view_src->view_offs;
        view_src   = view_src->view_src;
    }
    size_t data_size = ggml_row_size(type, ne[0]);
    for (int i = 1; i < n_dims; i++) data_size *= (size_t) ne[i];

    void * data = view_src != NULL ? view_src->data : NULL;
    if (data != NULL) data = (char *) data + view_offs;

    size_t obj_alloc_size = 0;
    if (view_src == NULL && !ctx->no_alloc) obj_alloc_size = data_size;

    struct ggml_object * obj = new_object(ctx, OBJECT_TENSOR, sizeof(struct ggml_tensor) + obj_alloc_size);
    struct ggml_tensor * result = (struct ggml_tensor *) (ctx->mem_buffer + obj->offs);
    memset(result, 0, sizeof(struct ggml_tensor));
    result->type = type;
    result->op = GGML_OP_NONE;
    result->view_src = view_src;
    result->view_offs = view_offs;
    result->data = obj_alloc_size > 0 ? (void *) (result + 1) : data;
    for (int i = 0; i < GGML_MAX_DIMS; i++) result->ne[i] = 1;
    for (int i = 0; i < n_dims; i++) result->ne[i] = ne[i];
    result->nb[0] = ggml_type_size(type);
    result->nb[1] = result->nb[0] * (size_t) (result->ne[0] / ggml_blck_size(type));
    for (int i = 2; i < GGML_MAX_DIMS; i++) result->nb[i] = result->nb[i - 1] * (size_t) result->ne[i - 1];
    return result;
}

extern "C" struct ggml_tensor * ggml_new_tensor(struct ggml_context * ctx, enum ggml_type type, int n_dims, const int64_t * ne) {
    return new_tensor_impl(ctx, type, n_dims, ne, NULL, 0);
}
extern "C" struct ggml_tensor * ggml_new_tensor_1d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0) {
    return ggml_new_tensor(ctx, type, 1, &ne0);
}
extern "C" struct ggml_tensor * ggml_new_tensor_2d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1) {
    const int64_t ne[2] = { ne0, ne1 };
    return ggml_new_tensor(ctx, type, 2, ne);
}
extern "C" struct ggml_tensor * ggml_new_tensor_3d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2) {
    const int64_t ne[3] = { ne0, ne1, ne2 };
    return ggml_new_tensor(ctx, type, 3, ne);
}
extern "C" struct ggml_tensor * ggml_new_tensor_4d(struct ggml_context * ctx, enum ggml_type type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const int64_t ne[4] = { ne0, ne1, ne2, ne3 };
    return ggml_new_tensor(ctx, type, 4, ne);
}
extern "C" struct ggml_tensor * ggml_dup_tensor(struct ggml_context * ctx, const struct ggml_tensor * src) {
    return ggml_new_tensor(ctx, src->type, GGML_MAX_DIMS, src->ne);
}

extern "C" struct ggml_tensor * ggml_format_name(struct ggml_tensor * tensor, const char * fmt, ...) {
    va_list args;
    va_start(args, fmt);
    vsnprintf(tensor->name, sizeof(tensor->name), fmt, args);
    va_end(args);
    return tensor;
}
extern "C" const char * ggml_get_name(const struct ggml_tensor * tensor) { return tensor->name; }
extern "C" struct ggml_tensor * ggml_set_name(struct ggml_tensor * tensor, const char * name) {
    size_t i = 0;
    for (; i < sizeof(tensor->name) - 1 && name[i]; i++) tensor->name[i] = name[i];
    tensor->name[i] = 0;
    return tensor;
}
extern "C" void ggml_set_input (struct ggml_tensor * t) { t->flags |= GGML_TENSOR_FLAG_INPUT; }
extern "C" void ggml_set_output(struct ggml_tensor * t) { t->flags |= GGML_TENSOR_FLAG_OUTPUT; }

extern "C" struct ggml_tensor * ggml_view_tensor(struct ggml_context * ctx, struct ggml_tensor * src) {
    struct ggml_tensor * result = new_tensor_impl(ctx, src->type, GGML_MAX_DIMS, src->ne, src, 0);
    ggml_format_name(result, "%s (view)", src->name);
    for (int i = 0; i < GGML_MAX_DIMS; i++) result->nb[i] = src->nb[i];
    return result;
}

extern "C" struct ggml_tensor * ggml_get_first_tensor(const struct ggml_context * ctx) {
    for (struct ggml_object * obj = ctx->objects_begin; obj; obj = obj->next)
        if (obj->type == OBJECT_TENSOR) return (struct ggml_tensor *) (ctx->mem_buffer + obj->offs);
    return NULL;
}
extern "C" struct ggml_tensor * ggml_get_next_tensor(const struct ggml_context * ctx, struct ggml_tensor * tensor) {
    struct ggml_object * obj = (struct ggml_object *) ((char *) tensor - sizeof(struct ggml_object));
    for (obj = obj->next; obj; obj = obj->next)
        if (obj->type == OBJECT_TENSOR) return (struct ggml_tensor *) (ctx->mem_buffer + obj->offs);
    return NULL;
}
extern "C" struct ggml_tensor * ggml_get_tensor(struct ggml_context * ctx, const char * name) {
    for (struct ggml_tensor * t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t))
        if (strcmp(t->name, name) == 0) return t;
    return NULL;
}

// ---------------------------------------------------------------------------------------------------
// op builders
// ---------------------------------------------------------------------------------------------------
#define CTX struct ggml_context * ctx
#define T   struct ggml_tensor *

static T binary_impl(CTX, enum ggml_op op, T a, T b, bool inplace) {
    GGML_ASSERT(can_repeat(b, a));
    T result = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    result->op = op;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}
extern "C" T ggml_add(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_ADD, a, b, false); }
extern "C" T ggml_add_inplace(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_ADD, a, b, true); }
extern "C" T ggml_sub(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_SUB, a, b, false); }
extern "C" T ggml_sub_inplace(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_SUB, a, b, true); }
extern "C" T ggml_mul_inplace(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_MUL, a, b, true); }
extern "C" T ggml_div_inplace(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_DIV, a, b, true); }
extern "C" T ggml_mul(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_MUL, a, b, false); }
extern "C" T ggml_div(CTX, T a, T b) { return binary_impl(ctx, GGML_OP_DIV, a, b, false); }

extern "C" T ggml_dup(CTX, T a) {
    T result = ggml_dup_tensor(ctx, a);
    result->op = GGML_OP_DUP;
    result->src[0] = a;
    return result;
}

static T unary_impl(CTX, T a, enum ggml_unary_op op) {
    GGML_ASSERT(rows_contiguous(a));
    T result = ggml_dup_tensor(ctx, a);
    result->op = GGML_OP_UNARY;
    result->op_params[0] = (int32_t) op;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_neg (CTX, T a) { return unary_impl(ctx, a, GGML_UNARY_OP_NEG); }
extern "C" T ggml_silu(CTX, T a) { return unary_impl(ctx, a, GGML_UNARY_OP_SILU); }
extern "C" T ggml_gelu(CTX, T a) { return unary_impl(ctx, a, GGML_UNARY_OP_GELU); }
extern "C" T ggml_elu (CTX, T a) { return unary_impl(ctx, a, GGML_UNARY_OP_ELU); }

static T scale_impl(CTX, T a, float s, bool inplace) {
    GGML_ASSERT(rows_contiguous(a));
    T result = inplace ? ggml_view_tensor(ctx, a) : ggml_dup_tensor(ctx, a);
    result->op = GGML_OP_SCALE;
    ggml_set_op_params_f32(result, 0, s);
    ggml_set_op_params_f32(result, 1, 0.0f);   // bias
    result->src[0] = a;
    return result;
}
extern "C" T ggml_scale(CTX, T a, float s) { return scale_impl(ctx, a, s, false); }
extern "C" T ggml_scale_inplace(CTX, T a, float s) { return scale_impl(ctx, a, s, true); }

// NOTE: ggml_clamp operates in place: the result is a view of `a` (the reference relies on it,
// transformer.h:269-270)
extern "C" T ggml_clamp(CTX, T a, float min, float max) {
    T result = ggml_view_tensor(ctx, a);
    result->op = GGML_OP_CLAMP;
    ggml_set_op_params_f32(result, 0, min);
    ggml_set_op_params_f32(result, 1, max);
    result->src[0] = a;
    return result;
}

extern "C" T ggml_sum(CTX, T a) {
    T result = ggml_new_tensor_1d(ctx, a->type, 1);
    result->op = GGML_OP_SUM;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_sum_rows(CTX, T a) {
    int64_t ne[GGML_MAX_DIMS] = { 1, a->ne[1], a->ne[2], a->ne[3] };
    T result = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, ne);
    result->op = GGML_OP_SUM_ROWS;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_argmax(CTX, T a) {
    GGML_ASSERT(is_matrix(a));
    GGML_ASSERT(a->ne[0] <= INT32_MAX);
    T result = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, a->ne[1]);
    result->op = GGML_OP_ARGMAX;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_argsort(CTX, T a, enum ggml_sort_order order) {
    GGML_ASSERT(a->ne[0] <= INT32_MAX);
    T result = ggml_new_tensor(ctx, GGML_TYPE_I32, GGML_MAX_DIMS, a->ne);
    result->op = GGML_OP_ARGSORT;
    result->op_params[0] = (int32_t) order;
    result->src[0] = a;
    return result;
}
// indices of the k largest values per row, in descending order of value
extern "C" T ggml_argsort_top_k(CTX, T a, int k) {
    GGML_ASSERT(a->ne[0] >= k);
    T result = ggml_argsort(ctx, a, GGML_SORT_ORDER_DESC);
    return ggml_view_4d(ctx, result, k, result->ne[1], result->ne[2], result->ne[3],
                        result->nb[1], result->nb[2], result->nb[3], 0);
}
extern "C" T ggml_top_k(CTX, T a, int k) { return ggml_argsort_top_k(ctx, a, k); }

extern "C" T ggml_arange(CTX, float start, float stop, float step) {
    GGML_ASSERT(stop > start);
    const int64_t steps = (int64_t) ceilf((stop - start) / step);
    T result = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, steps);
    result->op = GGML_OP_ARANGE;
    ggml_set_op_params_f32(result, 0, start);
    ggml_set_op_params_f32(result, 1, stop);
    ggml_set_op_params_f32(result, 2, step);
    return result;
}

extern "C" T ggml_repeat(CTX, T a, T b) {
    GGML_ASSERT(can_repeat(a, b));
    T result = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, b->ne);
    result->op = GGML_OP_REPEAT;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_repeat_4d(CTX, T a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    const bool ok = is_empty(a) || (ne0 % a->ne[0] == 0 && ne1 % a->ne[1] == 0 && ne2 % a->ne[2] == 0 && ne3 % a->ne[3] == 0);
    GGML_ASSERT(ok);
    T result = ggml_new_tensor_4d(ctx, a->type, ne0, ne1, ne2, ne3);
    result->op = GGML_OP_REPEAT;
    result->src[0] = a;
    return result;
}

extern "C" T ggml_concat(CTX, T a, T b, int dim) {
    GGML_ASSERT(dim >= 0 && dim < GGML_MAX_DIMS);
    GGML_ASSERT(a->type == b->type);
    int64_t ne[GGML_MAX_DIMS];
    for (int d = 0; d < GGML_MAX_DIMS; ++d) {
        if (d == dim) { ne[d] = a->ne[d] + b->ne[d]; continue; }
        GGML_ASSERT(a->ne[d] == b->ne[d]);
        ne[d] = a->ne[d];
    }
    T result = ggml_new_tensor(ctx, a->type, GGML_MAX_DIMS, ne);
    result->op = GGML_OP_CONCAT;
    result->op_params[0] = dim;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}

extern "C" T ggml_pad(CTX, T a, int p0, int p1, int p2, int p3) {
    T result = ggml_new_tensor_4d(ctx, a->type, a->ne[0] + p0, a->ne[1] + p1, a->ne[2] + p2, a->ne[3] + p3);
    result->op = GGML_OP_PAD;
    result->op_params[0] = p0; result->op_params[1] = p1; result->op_params[2] = p2; result->op_params[3] = p3;
    result->src[0] = a;
    return result;
}

static T norm_impl(CTX, enum ggml_op op, T a, float eps) {
    T result = ggml_dup_tensor(ctx, a);
    result->op = op;
    ggml_set_op_params_f32(result, 0, eps);
    result->src[0] = a;
    return result;
}
extern "C" T ggml_norm(CTX, T a, float eps) { return norm_impl(ctx, GGML_OP_NORM, a, eps); }
extern "C" T ggml_rms_norm(CTX, T a, float eps) { return norm_impl(ctx, GGML_OP_RMS_NORM, a, eps); }

extern "C" T ggml_mul_mat(CTX, T a, T b) {
    const bool ok = a->ne[0] == b->ne[0] && b->ne[2] % a->ne[2] == 0 && b->ne[3] % a->ne[3] == 0;
    GGML_ASSERT(ok);
    GGML_ASSERT(!ggml_is_transposed(a));
    const int64_t ne[4] = { a->ne[1], b->ne[1], b->ne[2], b->ne[3] };
    T result = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    result->op = GGML_OP_MUL_MAT;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}

extern "C" T ggml_soft_max_ext(CTX, T a, T mask, float scale, float max_bias) {
    GGML_ASSERT(ggml_is_contiguous(a));
    if (mask) {
        GGML_ASSERT(mask->type == GGML_TYPE_F16 || mask->type == GGML_TYPE_F32);
        GGML_ASSERT(ggml_is_contiguous(mask));
        GGML_ASSERT(mask->ne[0] == a->ne[0]);
        GGML_ASSERT(mask->ne[1] >= a->ne[1]);
        GGML_ASSERT(a->ne[2] % mask->ne[2] == 0);
        GGML_ASSERT(a->ne[3] % mask->ne[3] == 0);
    }
    if (max_bias > 0.0f) GGML_ASSERT(mask);
    T result = ggml_dup_tensor(ctx, a);
    result->op = GGML_OP_SOFT_MAX;
    ggml_set_op_params_f32(result, 0, scale);
    ggml_set_op_params_f32(result, 1, max_bias);
    result->src[0] = a;
    result->src[1] = mask;
    return result;
}
extern "C" T ggml_soft_max(CTX, T a) { return ggml_soft_max_ext(ctx, a, NULL, 1.0f, 0.0f); }

extern "C" T ggml_cpy(CTX, T a, T b) {
    GGML_ASSERT(ggml_nelements(a) == ggml_nelements(b));
    T result = ggml_view_tensor(ctx, b);
    if (strlen(b->name) > 0) ggml_format_name(result, "%s (copy of %s)", b->name, a->name);
    else                     ggml_format_name(result, "%s (copy)", a->name);
    result->op = GGML_OP_CPY;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}
extern "C" T ggml_cast(CTX, T a, enum ggml_type type) {
    T result = ggml_new_tensor(ctx, type, GGML_MAX_DIMS, a->ne);
    ggml_format_name(result, "%s (copy)", a->name);
    result->op = GGML_OP_CPY;
    result->src[0] = a;
    result->src[1] = result;
    return result;
}
extern "C" T ggml_cont(CTX, T a) {
    T result = ggml_dup_tensor(ctx, a);
    ggml_format_name(result, "%s (cont)", a->name);
    result->op = GGML_OP_CONT;
    result->src[0] = a;
    return result;
}

static T reshape_impl(CTX, T a, int n_dims, const int64_t * ne) {
    GGML_ASSERT(ggml_is_contiguous(a));
    int64_t n = 1;
    for (int i = 0; i < n_dims; i++) n *= ne[i];
    GGML_ASSERT(ggml_nelements(a) == n);
    T result = new_tensor_impl(ctx, a->type, n_dims, ne, a, 0);
    ggml_format_name(result, "%s (reshaped)", a->name);
    result->op = GGML_OP_RESHAPE;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_reshape_1d(CTX, T a, int64_t ne0) { return reshape_impl(ctx, a, 1, &ne0); }
extern "C" T ggml_reshape_2d(CTX, T a, int64_t ne0, int64_t ne1) { const int64_t ne[2] = { ne0, ne1 }; return reshape_impl(ctx, a, 2, ne); }
extern "C" T ggml_reshape_3d(CTX, T a, int64_t ne0, int64_t ne1, int64_t ne2) { const int64_t ne[3] = { ne0, ne1, ne2 }; return reshape_impl(ctx, a, 3, ne); }
extern "C" T ggml_reshape_4d(CTX, T a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) { const int64_t ne[4] = { ne0, ne1, ne2, ne3 }; return reshape_impl(ctx, a, 4, ne); }

static T view_impl(CTX, T a, int n_dims, const int64_t * ne, size_t offset) {
    T result = new_tensor_impl(ctx, a->type, n_dims, ne, a, offset);
    ggml_format_name(result, "%s (view)", a->name);
    memcpy(result->op_params, &offset, sizeof(offset));
    result->op = GGML_OP_VIEW;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_view_1d(CTX, T a, int64_t ne0, size_t offset) { return view_impl(ctx, a, 1, &ne0, offset); }
extern "C" T ggml_view_2d(CTX, T a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset) {
    const int64_t ne[2] = { ne0, ne1 };
    T result = view_impl(ctx, a, 2, ne, offset);
    result->nb[1] = nb1;
    result->nb[2] = result->nb[1] * (size_t) ne1;
    result->nb[3] = result->nb[2];
    return result;
}
extern "C" T ggml_view_3d(CTX, T a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset) {
    const int64_t ne[3] = { ne0, ne1, ne2 };
    T result = view_impl(ctx, a, 3, ne, offset);
    result->nb[1] = nb1;
    result->nb[2] = nb2;
    result->nb[3] = result->nb[2] * (size_t) ne2;
    return result;
}
extern "C" T ggml_view_4d(CTX, T a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, size_t nb1, size_t nb2, size_t nb3, size_t offset) {
    const int64_t ne[4] = { ne0, ne1, ne2, ne3 };
    T result = view_impl(ctx, a, 4, ne, offset);
    result->nb[1] = nb1;
    result->nb[2] = nb2;
    result->nb[3] = nb3;
    return result;
}

extern "C" T ggml_permute(CTX, T a, int axis0, int axis1, int axis2, int axis3) {
    GGML_ASSERT(axis0 >= 0 && axis0 < GGML_MAX_DIMS && axis1 >= 0 && axis1 < GGML_MAX_DIMS);
    GGML_ASSERT(axis2 >= 0 && axis2 < GGML_MAX_DIMS && axis3 >= 0 && axis3 < GGML_MAX_DIMS);
    GGML_ASSERT(axis0 != axis1 && axis0 != axis2 && axis0 != axis3 && axis1 != axis2 && axis1 != axis3 && axis2 != axis3);
    T result = ggml_view_tensor(ctx, a);
    ggml_format_name(result, "%s (permuted)", a->name);
    const int ax[4] = { axis0, axis1, axis2, axis3 };
    int64_t ne[4]; size_t nb[4];
    for (int i = 0; i < 4; i++) { ne[ax[i]] = a->ne[i]; nb[ax[i]] = a->nb[i]; }
    for (int i = 0; i < 4; i++) { result->ne[i] = ne[i]; result->nb[i] = nb[i]; result->op_params[i] = ax[i]; }
    result->op = GGML_OP_PERMUTE;
    result->src[0] = a;
    return result;
}
extern "C" T ggml_transpose(CTX, T a) {
    T result = ggml_view_tensor(ctx, a);
    ggml_format_name(result, "%s (transposed)", a->name);
    result->ne[0] = a->ne[1]; result->ne[1] = a->ne[0];
    result->nb[0] = a->nb[1]; result->nb[1] = a->nb[0];
    result->op = GGML_OP_TRANSPOSE;
    result->src[0] = a;
    return result;
}

extern "C" T ggml_get_rows(CTX, T a, T b) {
    GGML_ASSERT(a->ne[2] == b->ne[1]);
    GGML_ASSERT(b->ne[3] == 1);
    GGML_ASSERT(b->type == GGML_TYPE_I32);
    const enum ggml_type type = a->type == GGML_TYPE_I32 ? a->type : GGML_TYPE_F32;
    T result = ggml_new_tensor_4d(ctx, type, a->ne[0], b->ne[0], b->ne[1], b->ne[2]);
    result->op = GGML_OP_GET_ROWS;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}

// a = destination, b = rows (F32), c = row indices; result is a view of a (transformer.h:246)
extern "C" T ggml_set_rows(CTX, T a, T b, T c) {
    GGML_ASSERT(a->ne[0] == b->ne[0]);
    GGML_ASSERT(a->ne[2] == b->ne[2]);
    GGML_ASSERT(a->ne[3] == b->ne[3]);
    GGML_ASSERT(b->ne[1] == c->ne[0]);
    GGML_ASSERT(b->ne[2] % c->ne[1] == 0);
    GGML_ASSERT(b->ne[3] % c->ne[2] == 0);
    GGML_ASSERT(c->ne[3] == 1);
    GGML_ASSERT(b->type == GGML_TYPE_F32);
    GGML_ASSERT(c->type == GGML_TYPE_I64 || c->type == GGML_TYPE_I32);
    GGML_ASSERT(rows_contiguous(a));
    GGML_ASSERT(rows_contiguous(b));
    T result = ggml_view_tensor(ctx, a);
    result->op = GGML_OP_SET_ROWS;
    result->src[0] = b;
    result->src[1] = c;
    result->src[2] = a;
    return result;
}

static int64_t conv_output_size(int64_t ins, int64_t ks, int s, int p, int d) {
    return (ins + 2 * p - d * (ks - 1) - 1) / s + 1;
}

// a: kernel [K, IC, OC] (1D), b: data [L, IC, N]; result [IC*K, OL, N] with column index ic*K + k
extern "C" T ggml_im2col(CTX, T a, T b, int s0, int s1, int p0, int p1, int d0, int d1, bool is_2D, enum ggml_type dst_type) {
    GGML_ASSERT(!is_2D);   // the moshi path only has 1-D convolutions (conv.h:81,157)
    GGML_ASSERT(a->ne[1] == b->ne[1]);
    GGML_ASSERT(b->ne[3] == 1);
    const int64_t OW = conv_output_size(b->ne[0], a->ne[0], s0, p0, d0);
    GGML_ASSERT(OW > 0);
    const int64_t ne[4] = { a->ne[1] * a->ne[0], OW, b->ne[2], 1 };
    T result = ggml_new_tensor(ctx, dst_type, 4, ne);
    const int32_t params[] = { s0, s1, p0, p1, d0, d1, is_2D ? 1 : 0 };
    memcpy(result->op_params, params, sizeof(params));
    result->op = GGML_OP_IM2COL;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}

// conv_1d = im2col (F16) followed by a matrix product, as ggml defines it
extern "C" T ggml_conv_1d(CTX, T a, T b, int s0, int p0, int d0) {
    T im2col = ggml_im2col(ctx, a, b, s0, 0, p0, 0, d0, 0, false, GGML_TYPE_F16);
    T result = ggml_mul_mat(ctx,
        ggml_reshape_2d(ctx, im2col, im2col->ne[0], im2col->ne[2] * im2col->ne[1]),
        ggml_reshape_2d(ctx, a, a->ne[0] * a->ne[1], a->ne[2]));
    return ggml_reshape_3d(ctx, result, im2col->ne[1], a->ne[2], im2col->ne[2]);
}

// a: kernel [K, OC, IC], b: data [L, IC]; result [(L-1)*s0 + K, OC]
extern "C" T ggml_conv_transpose_1d(CTX, T a, T b, int s0, int p0, int d0) {
    GGML_ASSERT(is_matrix(b));
    GGML_ASSERT(a->ne[2] == b->ne[1]);
    GGML_ASSERT(a->ne[3] == 1);
    GGML_ASSERT(p0 == 0);
    GGML_ASSERT(d0 == 1);
    const int64_t ne[4] = { (b->ne[0] - 1) * s0 + a->ne[0], a->ne[1], 1, 1 };
    T result = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
    result->op_params[0] = s0; result->op_params[1] = p0; result->op_params[2] = d0;
    result->op = GGML_OP_CONV_TRANSPOSE_1D;
    result->src[0] = a;
    result->src[1] = b;
    return result;
}

extern "C" T ggml_timestep_embedding(CTX, T timesteps, int dim, int max_period) {
    GGML_ASSERT(is_vector(timesteps) || timesteps->ne[1] == 1);
    T result = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, dim, timesteps->ne[0]);
    result->op_params[0] = dim;
    result->op_params[1] = max_period;
    result->op = GGML_OP_TIMESTEP_EMBEDDING;
    result->src[0] = timesteps;
    return result;
}

#undef CTX
#undef T

// ---------------------------------------------------------------------------------------------------
// graphs
// ---------------------------------------------------------------------------------------------------
static size_t graph_visited_size(size_t size) {
    size_t n = 16;
    while (n < size * 4) n <<= 1;   // <= 25% load: linear probing stays short
    return n;
}
static size_t graph_nbytes(size_t size) {
    return sizeof(struct ggml_cgraph) + 2 * size * sizeof(struct ggml_tensor *) + graph_visited_size(size) * sizeof(struct ggml_tensor *);
}
extern "C" size_t ggml_graph_overhead_custom(size_t size, bool grads) {
    GGML_UNUSED(grads);
    return sizeof(struct ggml_object) + GGML_PAD(graph_nbytes(size), GGML_MEM_ALIGN);
}
extern "C" size_t ggml_graph_overhead(void) { return ggml_graph_overhead_custom(GGML_DEFAULT_GRAPH_SIZE, false); }

extern "C" struct ggml_cgraph * ggml_new_graph_custom(struct ggml_context * ctx, size_t size, bool grads) {
    GGML_ASSERT(!grads);   // inference only (src/context.h:489 passes false)
    struct ggml_object * obj = new_object(ctx, OBJECT_GRAPH, graph_nbytes(size));
    struct ggml_cgraph * g = (struct ggml_cgraph *) (ctx->mem_buffer + obj->offs);
    g->size = (int) size;
    g->n_nodes = 0;
    g->n_leafs = 0;
    g->nodes = (struct ggml_tensor **) (g + 1);
    g->leafs = g->nodes + size;
    g->visited_size = graph_visited_size(size);
    g->visited = g->leafs + size;
    memset(g->visited, 0, g->visited_size * sizeof(struct ggml_tensor *));
    return g;
}
extern "C" struct ggml_cgraph * ggml_new_graph(struct ggml_context * ctx) {
    return ggml_new_graph_custom(ctx, GGML_DEFAULT_GRAPH_SIZE, false);
}
extern "C" void ggml_graph_clear(struct ggml_cgraph * g) {
    g->n_nodes = g->n_leafs = 0;
    memset(g->visited, 0, g->visited_size * sizeof(struct ggml_tensor *));
}

// returns true when t was already in the set
static bool visited_insert(struct ggml_cgraph * g, struct ggml_tensor * t) {
    const size_t mask = g->visited_size - 1;
    size_t h = ((uintptr_t) t >> 4) * 0x9E3779B97F4A7C15ull >> 20 & mask;
    while (g->visited[h]) {
        if (g->visited[h] == t) return true;
        h = (h + 1) & mask;
    }
    g->visited[h] = t;
    return false;
}

static void visit_parents(struct ggml_cgraph * g, struct ggml_tensor * node) {
    if (visited_insert(g, node)) return;
    for (int i = 0; i < GGML_MAX_SRC; ++i)
        if (node->src[i]) visit_parents(g, node->src[i]);
    if (node->op == GGML_OP_NONE && !(node->flags & GGML_TENSOR_FLAG_PARAM)) {
        GGML_ASSERT(g->n_leafs < g->size);
        if (strlen(node->name) == 0) ggml_format_name(node, "leaf_%d", g->n_leafs);
        g->leafs[g->n_leafs++] = node;
    } else {
        GGML_ASSERT(g->n_nodes < g->size);
        if (strlen(node->name) == 0) ggml_format_name(node, "node_%d", g->n_nodes);
        g->nodes[g->n_nodes++] = node;
    }
}

// appends the not-yet-visited ancestors of `tensor` in DFS post-order; execution order = node order
extern "C" void ggml_build_forward_expand(struct ggml_cgraph * g, struct ggml_tensor * tensor) {
    const int n0 = g->n_nodes;
    visit_parents(g, tensor);
    if (g->n_nodes > n0) GGML_ASSERT(g->nodes[g->n_nodes - 1] == tensor);
}

extern "C" int ggml_graph_size(struct ggml_cgraph * g) { return g->size; }
extern "C" int ggml_graph_n_nodes(struct ggml_cgraph * g) { return g->n_nodes; }
extern "C" struct ggml_tensor * ggml_graph_node(struct ggml_cgraph * g, int i) {
    if (i < 0) { GGML_ASSERT(g->n_nodes + i >= 0); return g->nodes[g->n_nodes + i]; }
    GGML_ASSERT(i < g->n_nodes);
    return g->nodes[i];
}
extern "C" struct ggml_tensor ** ggml_graph_nodes(struct ggml_cgraph * g) { return g->nodes; }
extern "C" int ggml_graph_n_leafs(struct ggml_cgraph * g) { return g->n_leafs; }
extern "C" struct ggml_tensor * ggml_graph_leaf(struct ggml_cgraph * g, int i) { GGML_ASSERT(i >= 0 && i < g->n_leafs); return g->leafs[i]; }

extern "C" void ggml_graph_print(const struct ggml_cgraph * g) {
    printf("=== GRAPH === n_nodes = %d, n_leafs = %d\n", g->n_nodes, g->n_leafs);
    for (int i = 0; i < g->n_nodes; i++) {
        const struct ggml_tensor * n = g->nodes[i];
        printf(" - %4d: [%6lld,%6lld,%5lld,%3lld] %-5s %-18s %s\n", i, (long long) n->ne[0], (long long) n->ne[1],
               (long long) n->ne[2], (long long) n->ne[3], ggml_type_name(n->type), ggml_op_desc(n), n->name);
    }
}

// ---------------------------------------------------------------------------------------------------
// row (de)quantisation — host utilities for loaders and tests (formats: SURVEY.md §8c)
// ---------------------------------------------------------------------------------------------------
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

static void get_scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// make_qkx2_quants [ggml-upstream, ggml-quants.c]: affine quantisation x ~ scale * l + min of n values to l in [0, nmax] that minimises the weighted squared
// error, searched over nstep + 1 candidate scales around nmax / (max - min) with a closed-form (scale, min) refit for each; returns scale, *the_min = -min
static float make_qkx2_quants(int n, int nmax, const float * x, const float * weights, uint8_t * L, float * the_min, uint8_t * Laux,
                              float rmin, float rdelta, int nstep) {
    float min = x[0], max = x[0], sum_w = weights[0], sum_x = sum_w * x[0];
    for (int i = 1; i < n; ++i) {
        if (x[i] < min) min = x[i];
        if (x[i] > max) max = x[i];
        const float w = weights[i];
        sum_w += w; sum_x += w * x[i];
    }
    if (min > 0) min = 0;
    if (max == min) { for (int i = 0; i < n; ++i) L[i] = 0; *the_min = -min; return 0.f; }
    float iscale = nmax / (max - min), scale = 1 / iscale, best_mad = 0;
    for (int i = 0; i < n; ++i) {
        int l = nearest_int(iscale * (x[i] - min));
        L[i] = (uint8_t) (l < 0 ? 0 : l > nmax ? nmax : l);
        float diff = scale * L[i] + min - x[i];
        diff = diff * diff;
        best_mad += weights[i] * diff;
    }
    for (int is = 0; is <= nstep; ++is) {
        iscale = (rmin + rdelta * is + nmax) / (max - min);
        float sum_l = 0, sum_l2 = 0, sum_xl = 0;
        for (int i = 0; i < n; ++i) {
            int l = nearest_int(iscale * (x[i] - min));
            l = l < 0 ? 0 : l > nmax ? nmax : l;
            Laux[i] = (uint8_t) l;
            const float w = weights[i];
            sum_l += w * l; sum_l2 += w * l * l; sum_xl += w * l * x[i];
        }
        const float D = sum_w * sum_l2 - sum_l * sum_l;
        if (D > 0) {
            float this_scale = (sum_w * sum_xl - sum_x * sum_l) / D, this_min = (sum_l2 * sum_x - sum_l * sum_xl) / D;
            if (this_min > 0) { this_min = 0; this_scale = sum_xl / sum_l2; }
            float mad = 0;
            for (int i = 0; i < n; ++i) { float diff = this_scale * Laux[i] + this_min - x[i]; diff = diff * diff; mad += weights[i] * diff; }
            if (mad < best_mad) { for (int i = 0; i < n; ++i) L[i] = Laux[i]; best_mad = mad; scale = this_scale; min = this_min; }
        }
    }
    *the_min = -min;
    return scale;
}

// quantize_row_q4_K_ref [ggml-upstream], one super-block: w = d * sc * q - dmin * m with 6-bit (sc, m) per 32-weight sub-block
static void quantize_q4_K_block(const float * x, block_q4_K * y) {
    uint8_t L[256], Laux[32];
    float weights[32], mins[8], scales[8];
    float max_scale = 0, max_min = 0;
    for (int j = 0; j < 8; j++) {
        float sum_x2 = 0;
        for (int l = 0; l < 32; ++l) sum_x2 += x[32 * j + l] * x[32 * j + l];
        const float av_x = sqrtf(sum_x2 / 32);
        for (int l = 0; l < 32; ++l) weights[l] = av_x + fabsf(x[32 * j + l]);
        scales[j] = make_qkx2_quants(32, 15, x + 32 * j, weights, L + 32 * j, &mins[j], Laux, -1.f, 0.1f, 20);
        if (scales[j] > max_scale) max_scale = scales[j];
        if (mins[j] > max_min) max_min = mins[j];
    }
    const float inv_scale = max_scale > 0 ? 63.f / max_scale : 0.f;
    const float inv_min   = max_min   > 0 ? 63.f / max_min   : 0.f;
    memset(y->scales, 0, K_SCALE_SIZE);
    for (int j = 0; j < 8; j++) {
        uint8_t ls = (uint8_t) nearest_int(inv_scale * scales[j]);
        uint8_t lm = (uint8_t) nearest_int(inv_min * mins[j]);
        if (ls > 63) ls = 63;
        if (lm > 63) lm = 63;
        if (j < 4) { y->scales[j] = ls; y->scales[j + 4] = lm; }
        else {
            y->scales[j + 4] = (ls & 0xF) | ((lm & 0xF) << 4);
            y->scales[j - 4] |= (uint8_t) ((ls >> 4) << 6);
            y->scales[j]     |= (uint8_t) ((lm >> 4) << 6);
        }
    }
    y->d    = ggml_fp32_to_fp16(max_scale / 63.f);
    y->dmin = ggml_fp32_to_fp16(max_min / 63.f);
    for (int j = 0; j < 8; j++) {
        uint8_t sc, m;
        get_scale_min_k4(j, y->scales, &sc, &m);
        const float d = ggml_fp16_to_fp32(y->d) * sc;
        if (!d) continue;                                   // (the search's own L stays)
        const float dm = ggml_fp16_to_fp32(y->dmin) * m;
        for (int l = 0; l < 32; l++) {
            int q = nearest_int((x[32 * j + l] + dm) / d);
            L[32 * j + l] = (uint8_t) (q < 0 ? 0 : q > 15 ? 15 : q);
        }
    }
    uint8_t * q = y->qs;
    for (int j = 0; j < 256; j += 64) {
        for (int l = 0; l < 32; l++) q[l] = L[j + l] | (L[j + l + 32] << 4);
        q += 32;
    }
}

extern "C" void ggml_quantize_row(enum ggml_type type, const float * x, void * vy, int64_t k) {
    switch (type) {
        case GGML_TYPE_F32: memcpy(vy, x, (size_t) k * 4); break;
        case GGML_TYPE_F16: { ggml_fp16_t * y = (ggml_fp16_t *) vy; for (int64_t i = 0; i < k; i++) y[i] = ggml_fp32_to_fp16(x[i]); } break;
        case GGML_TYPE_BF16: { ggml_bf16_t * y = (ggml_bf16_t *) vy; for (int64_t i = 0; i < k; i++) y[i] = ggml_fp32_to_bf16(x[i]); } break;
        case GGML_TYPE_Q8_0: {
            block_q8_0 * y = (block_q8_0 *) vy;
            for (int64_t i = 0; i < k / 32; i++) {
                float amax = 0;
                for (int j = 0; j < 32; j++) { float v = fabsf(x[i * 32 + j]); if (v > amax) amax = v; }
                const float d = amax / 127.f, id = d ? 1.f / d : 0.f;
                y[i].d = ggml_fp32_to_fp16(d);
                for (int j = 0; j < 32; j++) y[i].qs[j] = (int8_t) roundf(x[i * 32 + j] * id);
            }
        } break;
        case GGML_TYPE_Q4_0: {
            block_q4_0 * y = (block_q4_0 *) vy;
            for (int64_t i = 0; i < k / 32; i++) {
                float amax = 0, max = 0;
                for (int j = 0; j < 32; j++) { float v = x[i * 32 + j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
                const float d = max / -8.f, id = d ? 1.f / d : 0.f;
                y[i].d = ggml_fp32_to_fp16(d);
                for (int j = 0; j < 16; j++) {
                    const float x0 = x[i * 32 + j] * id, x1 = x[i * 32 + 16 + j] * id;
                    int a = (int) (x0 + 8.5f), b = (int) (x1 + 8.5f);
                    const uint8_t xi0 = (uint8_t) (a > 15 ? 15 : a), xi1 = (uint8_t) (b > 15 ? 15 : b);
                    y[i].qs[j] = xi0 | (xi1 << 4);
                }
            }
        } break;
        case GGML_TYPE_Q4_K: {
            block_q4_K * y = (block_q4_K *) vy;
            for (int64_t i = 0; i < k / 256; i++) quantize_q4_K_block(x + i * 256, y + i);
        } break;
        case GGML_TYPE_Q8_K: {
            block_q8_K * y = (block_q8_K *) vy;
            for (int64_t i = 0; i < k / 256; i++) {
                float max = 0, amax = 0;
                for (int j = 0; j < 256; j++) { float ax = fabsf(x[i * 256 + j]); if (ax > amax) { amax = ax; max = x[i * 256 + j]; } }
                if (!amax) { y[i].d = 0; memset(y[i].qs, 0, 256); memset(y[i].bsums, 0, sizeof(y[i].bsums)); continue; }
                const float iscale = -127.f / max;
                for (int j = 0; j < 256; j++) { int v = nearest_int(iscale * x[i * 256 + j]); y[i].qs[j] = (int8_t) (v > 127 ? 127 : v); }
                for (int j = 0; j < 16; j++) { int s = 0; for (int l = 0; l < 16; l++) s += y[i].qs[j * 16 + l]; y[i].bsums[j] = (int16_t) s; }
                y[i].d = 1.f / iscale;
            }
        } break;
        default: GGML_ABORT("ggml_quantize_row: unsupported type %s", ggml_type_name(type));
    }
}

extern "C" void ggml_dequantize_row(enum ggml_type type, const void * vx, float * y, int64_t k) {
    switch (type) {
        case GGML_TYPE_F32: memcpy(y, vx, (size_t) k * 4); break;
        case GGML_TYPE_F16: { const ggml_fp16_t * x = (const ggml_fp16_t *) vx; for (int64_t i = 0; i < k; i++) y[i] = ggml_fp16_to_fp32(x[i]); } break;
        case GGML_TYPE_BF16: { const ggml_bf16_t * x = (const ggml_bf16_t *) vx; for (int64_t i = 0; i < k; i++) y[i] = ggml_bf16_to_fp32(x[i]); } break;
        case GGML_TYPE_Q8_0: {
            const block_q8_0 * x = (const block_q8_0 *) vx;
            for (int64_t i = 0; i < k / 32; i++) { const float d = ggml_fp16_to_fp32(x[i].d); for (int j = 0; j < 32; j++) y[i * 32 + j] = x[i].qs[j] * d; }
        } break;
        case GGML_TYPE_Q4_0: {
            const block_q4_0 * x = (const block_q4_0 *) vx;
            for (int64_t i = 0; i < k / 32; i++) {
                const float d = ggml_fp16_to_fp32(x[i].d);
                for (int j = 0; j < 16; j++) {
                    y[i * 32 + j]      = ((x[i].qs[j] & 0x0F) - 8) * d;
                    y[i * 32 + j + 16] = ((x[i].qs[j] >> 4) - 8) * d;
                }
            }
        } break;
        case GGML_TYPE_Q4_K: {
            const block_q4_K * x = (const block_q4_K *) vx;
            for (int64_t i = 0; i < k / 256; i++) {
                const uint8_t * q = x[i].qs;
                const float d = ggml_fp16_to_fp32(x[i].d), min = ggml_fp16_to_fp32(x[i].dmin);
                int is = 0;
                uint8_t sc, m;
                for (int j = 0; j < 256; j += 64) {
                    get_scale_min_k4(is + 0, x[i].scales, &sc, &m);
                    const float d1 = d * sc, m1 = min * m;
                    get_scale_min_k4(is + 1, x[i].scales, &sc, &m);
                    const float d2 = d * sc, m2 = min * m;
                    for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
                    for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
                    q += 32; is += 2;
                }
            }
        } break;
        default: GGML_ABORT("ggml_dequantize_row: unsupported type %s", ggml_type_name(type));
    }
}
