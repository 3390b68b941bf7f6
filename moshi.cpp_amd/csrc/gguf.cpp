// gguf.cpp — GGUF v3 container reader / writer behind include/gguf.h (the subset WeightLoader uses,
// src/loader.h:85-99, 227-270 in the reference). Format: magic, version, n_tensors, n_kv, kv pairs,
// tensor infos (name, dims, type, offset), padding to general.alignment, tensor data.
#include "gguf.h"
#include "ggml-backend.h"

#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

enum gguf_type {
    GGUF_TYPE_UINT8 = 0, GGUF_TYPE_INT8 = 1, GGUF_TYPE_UINT16 = 2, GGUF_TYPE_INT16 = 3, GGUF_TYPE_UINT32 = 4, GGUF_TYPE_INT32 = 5,
    GGUF_TYPE_FLOAT32 = 6, GGUF_TYPE_BOOL = 7, GGUF_TYPE_STRING = 8, GGUF_TYPE_ARRAY = 9, GGUF_TYPE_UINT64 = 10, GGUF_TYPE_INT64 = 11,
    GGUF_TYPE_FLOAT64 = 12, GGUF_TYPE_COUNT,
};
static size_t gguf_type_size(int t) {
    static const size_t sz[GGUF_TYPE_COUNT] = { 1, 1, 2, 2, 4, 4, 4, 1, 0, 0, 8, 8, 8 };
    return t >= 0 && t < GGUF_TYPE_COUNT ? sz[t] : 0;
}

struct gguf_kv { std::string key; int type; std::vector<uint8_t> raw; std::string str; int arr_type = 0; uint64_t arr_n = 0; std::vector<std::string> arr_str; };
struct gguf_tensor_info {
    std::string name;
    uint32_t n_dims;
    int64_t ne[GGML_MAX_DIMS];
    enum ggml_type type;
    uint64_t offset;
    size_t size;
    const struct ggml_tensor * tensor;   // for writing
};
struct gguf_context {
    uint32_t version = GGUF_VERSION;
    size_t alignment = GGUF_DEFAULT_ALIGNMENT;
    size_t data_offset = 0;
    std::vector<gguf_kv> kv;
    std::vector<gguf_tensor_info> infos;
};

static bool rd(FILE * f, void * p, size_t n) { return fread(p, 1, n, f) == n; }
static bool rd_str(FILE * f, std::string & s) {
    uint64_t n;
    if (!rd(f, &n, 8) || n > (1u << 30)) return false;
    s.resize((size_t) n);
    return n == 0 || rd(f, &s[0], (size_t) n);
}

extern "C" struct gguf_context * gguf_init_empty(void) { return new gguf_context; }
extern "C" void gguf_free(struct gguf_context * ctx) { delete ctx; }

extern "C" struct gguf_context * gguf_init_from_file(const char * fname, struct gguf_init_params params) {
    FILE * f = fopen(fname, "rb");
    if (!f) return NULL;
    gguf_context * g = new gguf_context;
    auto fail = [&]() { fclose(f); delete g; return (gguf_context *) NULL; };
    char magic[4];
    uint64_t n_tensors = 0, n_kv = 0;
    if (!rd(f, magic, 4) || memcmp(magic, GGUF_MAGIC, 4) != 0) return fail();
    if (!rd(f, &g->version, 4) || g->version < 2 || g->version > 3) return fail();
    if (!rd(f, &n_tensors, 8) || !rd(f, &n_kv, 8)) return fail();
    for (uint64_t i = 0; i < n_kv; i++) {
        gguf_kv kv;
        uint32_t t;
        if (!rd_str(f, kv.key) || !rd(f, &t, 4)) return fail();
        kv.type = (int) t;
        if (t == GGUF_TYPE_STRING) { if (!rd_str(f, kv.str)) return fail(); }
        else if (t == GGUF_TYPE_ARRAY) {
            uint32_t at;
            if (!rd(f, &at, 4) || !rd(f, &kv.arr_n, 8)) return fail();
            kv.arr_type = (int) at;
            if (at == GGUF_TYPE_STRING) { kv.arr_str.resize((size_t) kv.arr_n); for (auto & s : kv.arr_str) if (!rd_str(f, s)) return fail(); }
            else {
                const size_t es = gguf_type_size((int) at);
                if (es == 0) return fail();
                kv.raw.resize(es * (size_t) kv.arr_n);
                if (!kv.raw.empty() && !rd(f, kv.raw.data(), kv.raw.size())) return fail();
            }
        } else {
            const size_t es = gguf_type_size((int) t);
            if (es == 0) return fail();
            kv.raw.resize(es);
            if (!rd(f, kv.raw.data(), es)) return fail();
        }
        if (kv.key == "general.alignment" && kv.type == GGUF_TYPE_UINT32) { uint32_t a; memcpy(&a, kv.raw.data(), 4); if (a) g->alignment = a; }
        g->kv.push_back(std::move(kv));
    }
    for (uint64_t i = 0; i < n_tensors; i++) {
        gguf_tensor_info ti = {};
        uint32_t type;
        if (!rd_str(f, ti.name) || !rd(f, &ti.n_dims, 4) || ti.n_dims > GGML_MAX_DIMS) return fail();
        for (int d = 0; d < GGML_MAX_DIMS; d++) ti.ne[d] = 1;
        for (uint32_t d = 0; d < ti.n_dims; d++) { uint64_t v; if (!rd(f, &v, 8)) return fail(); ti.ne[d] = (int64_t) v; }
        if (!rd(f, &type, 4) || !rd(f, &ti.offset, 8)) return fail();
        if (type >= GGML_TYPE_COUNT || ggml_blck_size((enum ggml_type) type) == 0) return fail();
        ti.type = (enum ggml_type) type;
        if (ti.ne[0] % ggml_blck_size(ti.type) != 0) return fail();
        ti.size = ggml_row_size(ti.type, ti.ne[0]) * (size_t) (ti.ne[1] * ti.ne[2] * ti.ne[3]);
        g->infos.push_back(ti);
    }
    const long pos = ftell(f);
    g->data_offset = GGML_PAD((size_t) pos, g->alignment);

    if (params.ctx != NULL) {
        size_t mem = (g->infos.size() + 1) * ggml_tensor_overhead();
        if (!params.no_alloc) for (auto & ti : g->infos) mem += GGML_PAD(ti.size, GGML_MEM_ALIGN);
        struct ggml_init_params ip = { mem, NULL, params.no_alloc };
        struct ggml_context * ctx = ggml_init(ip);
        for (auto & ti : g->infos) {
            struct ggml_tensor * t = ggml_new_tensor(ctx, ti.type, GGML_MAX_DIMS, ti.ne);
            ggml_set_name(t, ti.name.c_str());
            if (!params.no_alloc) {
                if (fseek(f, (long) (g->data_offset + ti.offset), SEEK_SET) != 0 || !rd(f, t->data, ti.size)) { ggml_free(ctx); return fail(); }
            }
        }
        *params.ctx = ctx;
    }
    fclose(f);
    return g;
}

extern "C" uint32_t gguf_get_version(const struct gguf_context * ctx) { return ctx->version; }
extern "C" size_t gguf_get_alignment(const struct gguf_context * ctx) { return ctx->alignment; }
extern "C" size_t gguf_get_data_offset(const struct gguf_context * ctx) { return ctx->data_offset; }

extern "C" int64_t gguf_get_n_kv(const struct gguf_context * ctx) { return (int64_t) ctx->kv.size(); }
extern "C" int64_t gguf_find_key(const struct gguf_context * ctx, const char * key) {
    for (size_t i = 0; i < ctx->kv.size(); i++) if (ctx->kv[i].key == key) return (int64_t) i;
    return -1;
}
extern "C" const char * gguf_get_key(const struct gguf_context * ctx, int64_t id) { return ctx->kv[(size_t) id].key.c_str(); }
extern "C" const char * gguf_get_val_str(const struct gguf_context * ctx, int64_t id) {
    GGML_ASSERT(ctx->kv[(size_t) id].type == GGUF_TYPE_STRING);
    return ctx->kv[(size_t) id].str.c_str();
}
extern "C" uint32_t gguf_get_val_u32(const struct gguf_context * ctx, int64_t id) {
    GGML_ASSERT(ctx->kv[(size_t) id].type == GGUF_TYPE_UINT32);
    uint32_t v; memcpy(&v, ctx->kv[(size_t) id].raw.data(), 4); return v;
}
static gguf_kv & kv_slot(struct gguf_context * ctx, const char * key) {
    const int64_t id = gguf_find_key(ctx, key);
    if (id >= 0) return ctx->kv[(size_t) id];
    ctx->kv.push_back(gguf_kv());
    ctx->kv.back().key = key;
    return ctx->kv.back();
}
extern "C" void gguf_set_val_u32(struct gguf_context * ctx, const char * key, uint32_t val) {
    gguf_kv & kv = kv_slot(ctx, key);
    kv.type = GGUF_TYPE_UINT32; kv.raw.resize(4); memcpy(kv.raw.data(), &val, 4);
    if (kv.key == "general.alignment" && val) ctx->alignment = val;
}
extern "C" void gguf_set_val_str(struct gguf_context * ctx, const char * key, const char * val) {
    gguf_kv & kv = kv_slot(ctx, key);
    kv.type = GGUF_TYPE_STRING; kv.str = val;
}

extern "C" int64_t gguf_get_n_tensors(const struct gguf_context * ctx) { return (int64_t) ctx->infos.size(); }
extern "C" int64_t gguf_find_tensor(const struct gguf_context * ctx, const char * name) {
    for (size_t i = 0; i < ctx->infos.size(); i++) if (ctx->infos[i].name == name) return (int64_t) i;
    return -1;
}
extern "C" const char * gguf_get_tensor_name(const struct gguf_context * ctx, int64_t id) { return ctx->infos[(size_t) id].name.c_str(); }
extern "C" enum ggml_type gguf_get_tensor_type(const struct gguf_context * ctx, int64_t id) { return ctx->infos[(size_t) id].type; }
extern "C" size_t gguf_get_tensor_offset(const struct gguf_context * ctx, int64_t id) { return (size_t) ctx->infos[(size_t) id].offset; }
extern "C" size_t gguf_get_tensor_size(const struct gguf_context * ctx, int64_t id) { return ctx->infos[(size_t) id].size; }

extern "C" void gguf_add_tensor(struct gguf_context * ctx, const struct ggml_tensor * tensor) {
    GGML_ASSERT(gguf_find_tensor(ctx, tensor->name) < 0 && "duplicate tensor name");
    gguf_tensor_info ti = {};
    ti.name = tensor->name;
    ti.n_dims = (uint32_t) ggml_n_dims(tensor);
    for (int d = 0; d < GGML_MAX_DIMS; d++) ti.ne[d] = tensor->ne[d];
    ti.type = tensor->type;
    ti.size = ggml_nbytes(tensor);
    ti.offset = ctx->infos.empty() ? 0 : ctx->infos.back().offset + GGML_PAD(ctx->infos.back().size, ctx->alignment);
    ti.tensor = tensor;
    ctx->infos.push_back(ti);
}

static void wr(FILE * f, const void * p, size_t n) { if (n) fwrite(p, 1, n, f); }
static void wr_str(FILE * f, const std::string & s) { uint64_t n = s.size(); wr(f, &n, 8); wr(f, s.data(), s.size()); }

extern "C" bool gguf_write_to_file(const struct gguf_context * ctx, const char * fname, bool only_meta) {
    FILE * f = fopen(fname, "wb");
    if (!f) return false;
    const uint32_t version = GGUF_VERSION;
    const uint64_t n_tensors = ctx->infos.size(), n_kv = ctx->kv.size();
    wr(f, GGUF_MAGIC, 4); wr(f, &version, 4); wr(f, &n_tensors, 8); wr(f, &n_kv, 8);
    for (auto & kv : ctx->kv) {
        wr_str(f, kv.key);
        const uint32_t t = (uint32_t) kv.type;
        wr(f, &t, 4);
        if (kv.type == GGUF_TYPE_STRING) wr_str(f, kv.str);
        else if (kv.type == GGUF_TYPE_ARRAY) {
            const uint32_t at = (uint32_t) kv.arr_type; wr(f, &at, 4); wr(f, &kv.arr_n, 8);
            if (kv.arr_type == GGUF_TYPE_STRING) for (auto & s : kv.arr_str) wr_str(f, s); else wr(f, kv.raw.data(), kv.raw.size());
        } else wr(f, kv.raw.data(), kv.raw.size());
    }
    for (auto & ti : ctx->infos) {
        wr_str(f, ti.name);
        wr(f, &ti.n_dims, 4);
        for (uint32_t d = 0; d < ti.n_dims; d++) { const uint64_t v = (uint64_t) ti.ne[d]; wr(f, &v, 8); }
        const uint32_t type = (uint32_t) ti.type;
        wr(f, &type, 4); wr(f, &ti.offset, 8);
    }
    if (!only_meta) {
        std::vector<uint8_t> buf;
        const uint8_t zeros[64] = { 0 };
        long pos = ftell(f);
        size_t pad = GGML_PAD((size_t) pos, ctx->alignment) - (size_t) pos;
        while (pad) { const size_t n = pad < sizeof(zeros) ? pad : sizeof(zeros); wr(f, zeros, n); pad -= n; }
        for (auto & ti : ctx->infos) {
            GGML_ASSERT(ti.tensor && "tensor data unavailable for writing");
            const struct ggml_tensor * t = ti.tensor;
            if (t->buffer == NULL || ggml_backend_buffer_is_host(t->buffer)) wr(f, t->data, ti.size);
            else { buf.resize(ti.size); ggml_backend_tensor_get(t, buf.data(), 0, ti.size); wr(f, buf.data(), ti.size); }
            pad = GGML_PAD(ti.size, ctx->alignment) - ti.size;
            while (pad) { const size_t n = pad < sizeof(zeros) ? pad : sizeof(zeros); wr(f, zeros, n); pad -= n; }
        }
    }
    const bool ok = ferror(f) == 0;
    fclose(f);
    return ok;
}
