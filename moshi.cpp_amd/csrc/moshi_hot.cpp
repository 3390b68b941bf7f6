// moshi_hot.cpp — host-side driver of the streaming-decode hot path behind include/moshi_hot.h.
//
// Restates, over the ggml C-ABI only, the graph construction and per-frame protocol of the reference's
// libmoshi for this path: Temporal transformer graph (lm.h:659-677, 853-879), chained Depth transformer
// graph (lm.h:446-553), Mimi decode / encode graphs (compression.h:149-204, 277-325) and the delay-ring
// frame driver (lm.h:778-979). The op sequences are the reference's (they are what the backend's fusion
// matchers key on and what the oracle executes node by node); the C++ around them is this project's own.
// Weights are synthetic and deterministic (no model files can reach the build or the GPU box).
#include "moshi_hot.h"
#include "gguf.h"
#include "ggml-cpu.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

typedef struct ggml_tensor * T;

namespace {

// ---------------------------------------------------------------------------------------------------
// deterministic generator (splitmix64) for synthetic weights
// ---------------------------------------------------------------------------------------------------
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    float uniform() { return (float) ((next() >> 40) + 1) * (1.0f / 16777217.0f); }   // (0, 1)
    float normal() { const float u1 = uniform(), u2 = uniform(); return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2); }
};
uint64_t name_seed(uint64_t seed, const std::string & name) {
    uint64_t h = seed ^ 0xcbf29ce484222325ull;
    for (char c : name) h = (h ^ (uint8_t) c) * 0x100000001b3ull;
    return h;
}

// ---------------------------------------------------------------------------------------------------
// graph context: persistent (built once, replayed) or scratch (rebuilt every frame)
// mirrors the upload / alloc / compute protocol of src/context.h:227-653
// ---------------------------------------------------------------------------------------------------
struct Builder {
    ggml_backend_t be;
    struct ggml_context * ctx;
    struct ggml_cgraph * gf = nullptr;
    ggml_backend_buffer_t buf = nullptr;
    struct upload { T t; std::vector<uint8_t> data; };
    std::vector<upload> consts;
    struct noise { T t; float lambd; };
    std::vector<noise> exponentials;
    std::vector<float> noise_tmp;
    struct download { T t; void * dst; size_t n; };
    std::vector<download> readbacks;   // ScratchContext::build_forward_expand(tensor, dst): read back after compute, before the buffer is freed (src/context.h:588-606)

    Builder(ggml_backend_t be_, size_t mb) : be(be_) { ctx = ggml_init({ mb * 1024 * 1024, NULL, true }); }
    ~Builder() { if (buf) ggml_backend_buffer_free(buf); ggml_free(ctx); }
    operator struct ggml_context * () { return ctx; }

    T tensor(enum ggml_type type, int64_t n0, int64_t n1 = 1, int64_t n2 = 1, int64_t n3 = 1) { return ggml_new_tensor_4d(ctx, type, n0, n1, n2, n3); }
    T constant(T t, const void * data) {
        consts.push_back({ t, std::vector<uint8_t>((const uint8_t *) data, (const uint8_t *) data + ggml_nbytes(t)) });
        return t;
    }
    T f32(float v) { return constant(ggml_new_tensor_1d(ctx, GGML_TYPE_F32, 1), &v); }
    T i32s(const std::vector<int32_t> & v) { return constant(ggml_new_tensor_1d(ctx, GGML_TYPE_I32, (int64_t) v.size()), v.data()); }
    T arange(int n) { std::vector<float> v((size_t) n); for (int i = 0; i < n; i++) v[(size_t) i] = (float) i; return constant(ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n), v.data()); }
    T fill(const int64_t * ne, float val) {
        T t = ggml_new_tensor(ctx, GGML_TYPE_F32, 4, ne);
        std::vector<float> v((size_t) ggml_nelements(t), val);
        return constant(t, v.data());
    }
    T exponential(int64_t n0, int64_t n1, float lambd) { T t = tensor(GGML_TYPE_F32, n0, n1); exponentials.push_back({ t, lambd }); return t; }
    // deferred scalar sets of a scratch context (src/context.h:562-586)
    void set_later(T t, const void * data, size_t n) { consts.push_back({ t, std::vector<uint8_t>((const uint8_t *) data, (const uint8_t *) data + n) }); }

    void expand(T t) {
        if (!gf) gf = ggml_new_graph_custom(ctx, GGML_DEFAULT_GRAPH_SIZE * 4, false);
        ggml_build_forward_expand(gf, t);
    }
    void alloc() {
        buf = ggml_backend_alloc_ctx_tensors(ctx, be);
        GGML_ASSERT(buf);
        for (auto & c : consts) ggml_backend_tensor_set(c.t, c.data.data(), 0, c.data.size());
    }
    void upload_noise() {
        for (auto & e : exponentials) {
            const int64_t n = ggml_nelements(e.t);
            noise_tmp.resize((size_t) n);
            for (int64_t i = 0; i < n; i++) noise_tmp[(size_t) i] = -logf(rand() / (float) RAND_MAX) / e.lambd;   // src/context.h:475-476
            ggml_backend_tensor_set(e.t, noise_tmp.data(), 0, (size_t) n * 4);
        }
    }
    // parity probes: every byte of the graph buffer reads 0xFF (NaN as F32 / BF16 / F16) until a kernel or an upload writes it
    void alloc_poisoned() {
        buf = ggml_backend_alloc_ctx_tensors(ctx, be);
        GGML_ASSERT(buf);
        ggml_backend_buffer_clear(buf, 0xFF);
        for (auto & c : consts) ggml_backend_tensor_set(c.t, c.data.data(), 0, c.data.size());
    }
    void release_scratch() {
        readbacks.clear(); consts.clear(); exponentials.clear();
        ggml_backend_buffer_free(buf);
        buf = nullptr;
        ggml_reset(ctx);
        gf = nullptr;
    }
    void compute() { upload_noise(); if (gf) ggml_backend_graph_compute(be, gf); }
    // scratch protocol: alloc -> upload -> compute -> free -> reset (src/context.h:628-653)
    void expand_read(T t, void * dst) { expand(t); readbacks.push_back({ t, dst, ggml_nbytes(t) }); }
    void compute_scratch() {
        alloc();
        compute();
        for (auto & r : readbacks) ggml_backend_tensor_get(r.t, r.dst, 0, r.n);
        readbacks.clear();
        consts.clear();
        exponentials.clear();
        ggml_backend_buffer_free(buf);
        buf = nullptr;
        ggml_reset(ctx);
        gf = nullptr;
    }
};

// ---------------------------------------------------------------------------------------------------
// weights: one context + one backend buffer, filled after allocation
// ---------------------------------------------------------------------------------------------------
struct Weights {
    ggml_backend_t be;
    struct ggml_context * ctx;
    ggml_backend_buffer_t buf = nullptr;
    uint64_t seed;
    struct pending { T t; std::function<void(T, Rng &, std::vector<uint8_t> &)> gen; std::string name; };
    std::vector<pending> todo;
    std::map<std::string, T> by_name;
    std::set<std::string> file_names;   // the names as a GGUF file carries them (file_name): unique, checked in add()
    size_t bytes[6] = { 0, 0, 0, 0, 0, 0 };   // by part; [5] = the tensor-parallel slices (also counted in their part)
    int part = 0;

    // the name a tensor carries inside the context and inside a GGUF file: the checkpoint name when it fits ggml's name field, else an 8-digit hex
    // digest of it (WeightLoader::tensor_name, loader.h:120-137: names longer than GGML_MAX_NAME are replaced by a CRC) - unique either way
    // Byte-identical to the reference: the digest is the IEEE 802.3 CRC-32 of the name (src/crc-bbf.h: width 32, poly 0x04c11db7, reflected in and
    // out, xor-in / xor-out 0xffffffff - i.e. zlib's crc32, restated here in its reflected bit-serial form) and the eight characters are what
    // loader.h:128-135 prints: character i is the LOW nibble of byte i of the 64-bit crc_t (the high-nibble assignment of that loop is overwritten),
    // so a name becomes four hex digits followed by "0000". Files written by the reference's tools therefore resolve here and vice versa.
    static uint32_t name_crc32(const std::string & name) {
        uint32_t crc = 0xffffffffu;
        for (unsigned char ch : name) {
            crc ^= ch;
            for (int b = 0; b < 8; b++) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1u)));
        }
        return crc ^ 0xffffffffu;
    }
    static std::string file_name(const std::string & name) {
        if (name.size() < GGML_MAX_NAME) return name;
        uint64_t crc = name_crc32(name);
        std::string out(8, '0');
        for (int i = 0; i < 8; i++) { out[(size_t) i] = "0123456789abcdef"[crc & 0xf]; crc >>= 8; }
        return out;
    }
    std::string gguf_path;     // non-empty: load() reads every tensor's bytes from this file (WeightLoader::load_gguf, loader.h:235-271) instead of generating them

    Weights(ggml_backend_t be_, uint64_t seed_, int max_tensors) : be(be_), seed(seed_) {
        ctx = ggml_init({ ggml_tensor_overhead() * (size_t) max_tensors, NULL, true });
    }
    ~Weights() { if (buf) ggml_backend_buffer_free(buf); ggml_free(ctx); }

    T add(const std::string & name, enum ggml_type type, int64_t n0, int64_t n1, int64_t n2, std::function<void(T, Rng &, std::vector<uint8_t> &)> gen) {
        T t = ggml_new_tensor_3d(ctx, type, n0, n1, n2);
        const std::string fname = file_name(name);
        // The reference keeps 16 bits of the CRC (loader.h:128-135), so two names of >= GGML_MAX_NAME characters share a file name once in 65 536 pairs:
        // gguf_add_tensor refuses a duplicate name and ggml_get_tensor could not tell the two apart - fail here, by name, rather than at save time
        if (!file_names.insert(fname).second)
            GGML_ABORT("moshi_hot: tensor '%s' has the same GGUF name '%s' as an earlier tensor (the reference's 16-bit digest of long names collides)", name.c_str(), fname.c_str());
        ggml_set_name(t, fname.c_str());
        todo.push_back({ t, gen, name });
        by_name[name] = t;   // full checkpoint name (ggml names are cut to GGML_MAX_NAME, which makes encoder/decoder tails collide)
        bytes[part] += ggml_nbytes(t);
        return t;
    }
    // a SLICE of a synthetic matrix [K, M]: output row j is rows[j] of the full matrix, restricted to the K-range [k0, k0 + Ks) (whole blocks).
    // The full matrix is generated exactly as add() would (same name, same seed), so N slices sum up to the unsplit layer (tensor parallelism).
    struct slice { T t; std::function<void(T, Rng &, std::vector<uint8_t> &)> gen; std::string name; int64_t K, M; std::vector<int64_t> rows; int64_t k0; };
    std::vector<slice> sliced;
    T add_slice(const std::string & name, enum ggml_type type, int64_t K, int64_t M, std::function<void(T, Rng &, std::vector<uint8_t> &)> gen,
                const std::vector<int64_t> & rows, int64_t k0, int64_t Ks) {
        GGML_ASSERT(k0 % ggml_blck_size(type) == 0 && Ks % ggml_blck_size(type) == 0 && k0 + Ks <= K);
        T t = ggml_new_tensor_2d(ctx, type, Ks, (int64_t) rows.size());
        const std::string nm = name + ".slice";
        ggml_set_name(t, nm.size() < GGML_MAX_NAME ? nm.c_str() : nm.substr(nm.size() - GGML_MAX_NAME + 1).c_str());
        sliced.push_back({ t, gen, name, K, M, rows, k0 });
        by_name[nm] = t;
        bytes[part] += ggml_nbytes(t);
        bytes[5] += ggml_nbytes(t);
        return t;
    }
    void load_gguf();
    void load() {
        buf = ggml_backend_alloc_ctx_tensors(ctx, be);
        GGML_ASSERT(buf);
        std::vector<uint8_t> tmp;
        for (auto & sl : sliced) {
            struct ggml_context * fc = ggml_init({ ggml_tensor_overhead() * 2, NULL, true });
            T full = ggml_new_tensor_2d(fc, sl.t->type, sl.K, sl.M);
            Rng r(name_seed(seed, sl.name));
            tmp.resize(ggml_nbytes(full));
            sl.gen(full, r, tmp);
            const size_t rb_full = ggml_row_size(sl.t->type, sl.K), rb = ggml_row_size(sl.t->type, sl.t->ne[0]), off = ggml_row_size(sl.t->type, sl.k0);
            std::vector<uint8_t> out(ggml_nbytes(sl.t));
            for (size_t j = 0; j < sl.rows.size(); j++) memcpy(out.data() + j * rb, tmp.data() + (size_t) sl.rows[j] * rb_full + off, rb);
            ggml_backend_tensor_set(sl.t, out.data(), 0, out.size());
            ggml_free(fc);
        }
        sliced.clear();
        if (!gguf_path.empty()) { load_gguf(); todo.clear(); return; }
        for (auto & p : todo) {
            Rng r(name_seed(seed, p.name));
            tmp.resize(ggml_nbytes(p.t));
            p.gen(p.t, r, tmp);
            ggml_backend_tensor_set(p.t, tmp.data(), 0, tmp.size());
        }
        todo.clear();
    }
};

// WeightLoader::load_gguf (loader.h:235-271): every tensor of the file is looked up by name in the model's weight context, its bytes are read at
// data_offset + tensor_offset and uploaded with ggml_backend_tensor_set. Type and size must be the ones the model asked for.
void Weights::load_gguf() {
    struct ggml_context * meta = nullptr;
    gguf_init_params params = { true, &meta };
    struct gguf_context * gg = gguf_init_from_file(gguf_path.c_str(), params);
    GGML_ASSERT(gg && "moshi_hot: cannot read the GGUF file");
    FILE * f = fopen(gguf_path.c_str(), "rb");
    GGML_ASSERT(f);
    const size_t data_offset = gguf_get_data_offset(gg);
    const int n_tensors = (int) gguf_get_n_tensors(gg);
    std::vector<uint8_t> data;
    int found = 0;
    // Only the tensors this load still has to fill count (tensor-parallel slices were cut from generated matrices above and are not in `todo`). File names
    // are unique inside a model (Weights::add refuses the 1-in-65 536 case of two long names sharing the reference's 16-bit digest, loader.h:128-135).
    std::map<std::string, std::deque<T>> want;
    for (auto & p : todo) want[ggml_get_name(p.t)].push_back(p.t);
    for (int i = 0; i < n_tensors; i++) {
        const char * name = gguf_get_tensor_name(gg, i);
        auto it = want.find(name);
        if (it == want.end() || it->second.empty()) continue;   // (a file may hold more than this configuration uses)
        T t = it->second.front();
        it->second.pop_front();
        const size_t nbytes = gguf_get_tensor_size(gg, i);
        GGML_ASSERT(gguf_get_tensor_type(gg, i) == t->type && nbytes == ggml_nbytes(t) && "GGUF tensor does not match the configuration");
        data.resize(nbytes);
        GGML_ASSERT(fseek(f, (long) (data_offset + gguf_get_tensor_offset(gg, i)), SEEK_SET) == 0 && fread(data.data(), nbytes, 1, f) == 1);
        ggml_backend_tensor_set(t, data.data(), 0, nbytes);
        found++;
    }
    fclose(f);
    GGML_ASSERT(found == (int) todo.size() && "the GGUF file lacks tensors this configuration needs");
    gguf_free(gg);
    if (meta) ggml_free(meta);
}

// generators ------------------------------------------------------------------------------------------------
void gen_const(T t, float v, std::vector<uint8_t> & out) { float * f = (float *) out.data(); for (int64_t i = 0; i < ggml_nelements(t); i++) f[i] = v; }

// N(0, std) in F32 / F16 / BF16
void gen_normal(T t, Rng & r, std::vector<uint8_t> & out, float std_) {
    const int64_t n = ggml_nelements(t);
    if (t->type == GGML_TYPE_F32) { float * f = (float *) out.data(); for (int64_t i = 0; i < n; i++) f[i] = r.normal() * std_; }
    else if (t->type == GGML_TYPE_F16) { ggml_fp16_t * h = (ggml_fp16_t *) out.data(); for (int64_t i = 0; i < n; i++) h[i] = ggml_fp32_to_fp16(r.normal() * std_); }
    else if (t->type == GGML_TYPE_BF16) { ggml_bf16_t * h = (ggml_bf16_t *) out.data(); for (int64_t i = 0; i < n; i++) h[i] = ggml_fp32_to_bf16(r.normal() * std_); }
    else GGML_ABORT("gen_normal: type");
}

// block-quantised weights with zero-mean values of standard deviation ~std_ (DESIGN.md "synthetic weights"):
//   Q4_K: w = d*sc*q - dmin*m with dmin = 7.5 d and m = sc  =>  w = d*sc*(q - 7.5), q uniform[0,15], sc uniform[1,63]
//   Q4_0: w = d*(q - 8);  Q8_0: w = d*q, q uniform[-127,127]
void gen_quant(T t, Rng & r, std::vector<uint8_t> & out, float std_) {
    const int64_t n = ggml_nelements(t);
    if (t->type == GGML_TYPE_Q4_K) {
        block_q4_K * b = (block_q4_K *) out.data();
        const float base = std_ / (4.61f * 36.9f);   // std(q-7.5) = 4.61, rms(sc) = 36.9
        for (int64_t i = 0; i < n / 256; i++) {
            const float d = fabsf(r.normal()) * base * 1.25f;   // E|N| = 0.8
            b[i].d = ggml_fp32_to_fp16(d);
            b[i].dmin = ggml_fp32_to_fp16(7.5f * ggml_fp16_to_fp32(b[i].d));
            uint8_t sc[8];
            const uint64_t z = r.next();
            for (int j = 0; j < 8; j++) sc[j] = (uint8_t) (1 + ((z >> (8 * j)) & 0xff) % 63);
            for (int j = 0; j < 4; j++) {
                b[i].scales[j]     = (uint8_t) ((sc[j] & 63) | ((sc[j + 4] >> 4) << 6));
                b[i].scales[j + 4] = (uint8_t) ((sc[j] & 63) | ((sc[j + 4] >> 4) << 6));   // mins = scales
                b[i].scales[j + 8] = (uint8_t) ((sc[j + 4] & 0xF) | ((sc[j + 4] & 0xF) << 4));
            }
            uint64_t * q = (uint64_t *) b[i].qs;
            for (int j = 0; j < 16; j++) q[j] = r.next();
        }
    } else if (t->type == GGML_TYPE_Q4_0) {
        block_q4_0 * b = (block_q4_0 *) out.data();
        for (int64_t i = 0; i < n / 32; i++) {
            b[i].d = ggml_fp32_to_fp16(fabsf(r.normal()) * std_ * 1.25f / 4.61f);
            uint64_t z0 = r.next(), z1 = r.next();
            memcpy(b[i].qs, &z0, 8); memcpy(b[i].qs + 8, &z1, 8);
        }
    } else if (t->type == GGML_TYPE_Q8_0) {
        block_q8_0 * b = (block_q8_0 *) out.data();
        for (int64_t i = 0; i < n / 32; i++) {
            b[i].d = ggml_fp32_to_fp16(fabsf(r.normal()) * std_ * 1.25f / 73.3f);
            for (int j = 0; j < 4; j++) { uint64_t z = r.next(); for (int k = 0; k < 8; k++) { int v = (int) ((z >> (8 * k)) & 0xff) - 128; b[i].qs[j * 8 + k] = (int8_t) (v < -127 ? -127 : v); } }
        }
    } else gen_normal(t, r, out, std_);
}

// ---------------------------------------------------------------------------------------------------
// modules (each builder emits the reference's op sequence; citations are to /root/reference/src)
// ---------------------------------------------------------------------------------------------------
struct Norm { bool rms; float eps; T w = nullptr, b = nullptr; };

// moshi_rms_norm (moshi/modules/transformer.h:15-23) / torch_nn_layer_norm (torch.h:49-60)
T apply_norm(Builder & c, const Norm & n, T x) {
    if (n.rms) {   // moshi_rms_norm (transformer.h:16-23) multiplies alpha * y, which ggml only broadcasts for one column; the batched
                   // prefill (T > 1, not a reference path) swaps the operands — same products
        T y = ggml_rms_norm(c, x, n.eps);
        return x->ne[1] > 1 ? ggml_mul(c, y, n.w) : ggml_mul(c, n.w, y);
    }
    x = ggml_norm(c, x, n.eps);
    x = ggml_mul(c, x, n.w);
    if (n.b) x = ggml_add(c, x, n.b);
    return x;
}

T linear(Builder & c, T w, T x) { return ggml_mul_mat(c, w, x); }   // torch_nn_linear, no bias on this path (torch.h:79-87)

struct Rot { T rotr = nullptr, roti = nullptr; };

// moshi_get_timestep_embedding (moshi/modules/rope.h:8-20)
Rot timestep_embedding(Builder & c, int Tn, int D, T offset, int max_period) {
    T ts = c.arange(Tn);
    ts = ggml_add(c, ts, offset);
    T rot = ggml_timestep_embedding(c, ts, D, max_period);
    Rot r;
    r.rotr = ggml_view_2d(c, rot, D / 2, Tn, rot->nb[1], 0);
    r.roti = ggml_view_2d(c, rot, D / 2, Tn, rot->nb[1], rot->nb[0] * (size_t) (D / 2));
    return r;
}

// one operand of moshi_apply_rope (rope.h:58-76): split interleaved (re, im) pairs into two [D/2, T, H, B] halves
void rope_split(Builder & c, T q, int64_t D_half, int64_t Tn, int64_t H, int64_t B, T & qr, T & qi) {
    const int64_t BH = B * H;
    q = ggml_cont(c, q);
    q = ggml_reshape_4d(c, q, 2, D_half, Tn, BH);
    q = ggml_cont(c, ggml_permute(c, q, 3, 0, 1, 2));
    qr = ggml_view_3d(c, q, D_half, Tn, BH, q->nb[1], q->nb[2], 0);
    qi = ggml_view_3d(c, q, D_half, Tn, BH, q->nb[1], q->nb[2], q->nb[2] * (size_t) BH);
    qr = ggml_reshape_4d(c, qr, D_half, Tn, H, B);
    qi = ggml_reshape_4d(c, qi, D_half, Tn, H, B);
}

// moshi_apply_rope (rope.h:33-128), time_before_heads = false
void apply_rope(Builder & c, T & q, T & k, const Rot & rot) {
    const int64_t B = q->ne[3], H = q->ne[2], Tn = q->ne[1], D = q->ne[0];
    T qr, qi, kr, ki;
    rope_split(c, q, D / 2, Tn, H, B, qr, qi);
    rope_split(c, k, D / 2, Tn, H, B, kr, ki);
    T qor = ggml_sub(c, ggml_mul(c, qr, rot.rotr), ggml_mul(c, qi, rot.roti));
    T qoi = ggml_add(c, ggml_mul(c, qr, rot.roti), ggml_mul(c, qi, rot.rotr));
    T kor = ggml_sub(c, ggml_mul(c, kr, rot.rotr), ggml_mul(c, ki, rot.roti));
    T koi = ggml_add(c, ggml_mul(c, kr, rot.roti), ggml_mul(c, ki, rot.rotr));
    q = ggml_concat(c, qor, qoi, 0);
    k = ggml_concat(c, kor, koi, 0);
}

struct Layer {
    Norm norm1, norm2;
    std::vector<T> in_proj, out_proj;          // one per step weight set (Depth) or a single entry
    std::vector<T> gate_in, gate_out;          // gated FFN (silu) weight sets
    T linear1 = nullptr, linear2 = nullptr;    // plain FFN (gelu)
    T layer_scale_1 = nullptr, layer_scale_2 = nullptr;
    T kcache = nullptr, vcache = nullptr;      // state: BF16 [D, C, H]
    // cross-attention (tts): LayerNorm eps 0, in_proj [dim, 3 dim] used through row views, cached F32 K/V [D, Tc, H]
    Norm norm_cross; T cross_in = nullptr, cross_out = nullptr, k_cross = nullptr, v_cross = nullptr;
};

struct Transformer {
    int dim = 0, heads = 0, capacity = 0, max_period = 0;
    std::vector<Layer> layers;
    std::vector<int> schedule;                 // weights_per_step_schedule (transformer.h:80-83); empty = step k uses weight set k
    // bias-mask lookup table (torch.h:162-223)
    struct Pat { struct ggml_context * ctx; ggml_backend_buffer_t buf; T pattern; };
    std::map<int, Pat> pats;                   // one table per block length T (frame steps use one T; batched prefill adds its own)
    // cached-graph inputs (transformer.h:1101-1109)
    T g_bias = nullptr, g_offset = nullptr, g_indices = nullptr;
    int offset = 0;
    ~Transformer() { for (auto & p : pats) { ggml_backend_buffer_free(p.second.buf); ggml_free(p.second.ctx); } }
};

// create_bias_pattern(capacity, t, hi = 0, lo = -inf) (torch.h:170-203)
T create_bias_pattern(ggml_backend_t be, Transformer & tr, int Tn) {
    auto it = tr.pats.find(Tn);
    if (it != tr.pats.end()) return it->second.pattern;
    const int C = tr.capacity, start = C * 2 - Tn, width = start + C;
    Transformer::Pat pt;
    pt.ctx = ggml_init({ ggml_tensor_overhead(), NULL, true });
    pt.pattern = ggml_new_tensor_2d(pt.ctx, GGML_TYPE_F32, width, Tn);
    pt.buf = ggml_backend_alloc_ctx_tensors(pt.ctx, be);
    tr.pats[Tn] = pt;
    std::vector<float> v((size_t) width * (size_t) Tn);
    for (int j = 0; j < Tn; j++) {
        float * row = v.data() + (size_t) j * (size_t) width;
        const int right = start + 1 + j;
        for (int i = 0; i < width; i++) row[i] = i < right ? 0.0f : -INFINITY;
        for (int i = 0; i < Tn - j - 1; i++) row[C - 1 - i] = -INFINITY;
    }
    ggml_backend_tensor_set(pt.pattern, v.data(), 0, v.size() * 4);
    return pt.pattern;
}

// bias_pattern_index (torch.h:205-223)
T bias_pattern_index(Builder & c, Transformer & tr, int Tn, int offset) {
    T pattern = create_bias_pattern(c.be, tr, Tn);
    const int C = tr.capacity, start = C * 2 - Tn;
    const int col = offset <= C ? start - offset : C - (offset % C);
    T view = ggml_view_2d(c, pattern, C, Tn, pattern->nb[1], (size_t) col * pattern->nb[0]);
    return ggml_cont(c, view);
}

// moshi_streaming_multihead_attention (transformer.h:449-583): self-attention, ring cache via set_rows
T attention(Builder & c, const Transformer & tr, Layer & L, int wi, T indices, T x, T attn_bias, const Rot * rot, bool per_step_views) {
    const int H = tr.heads;
    T xin = x;
    if (per_step_views)   // moshi_apply_weights_per_step_linear takes a per-timestep view first (transformer.h:85-91)
        xin = ggml_view_3d(c, x, x->ne[0], 1, x->ne[2], x->nb[1], x->nb[2], 0);
    T projected = linear(c, L.in_proj[(size_t) wi], xin);
    T q = ggml_view_3d(c, projected, projected->ne[0] / 3, projected->ne[1], projected->ne[2], projected->nb[1], projected->nb[2], 0);
    q = ggml_cont(c, q);
    T k = ggml_view_3d(c, projected, projected->ne[0] / 3, projected->ne[1], projected->ne[2], projected->nb[1], projected->nb[2], projected->nb[1] / 3);
    k = ggml_cont(c, k);
    k = ggml_reshape_4d(c, k, k->ne[0] / H, H, k->ne[1], k->ne[2]);
    k = ggml_permute(c, k, 0, 2, 1, 3);
    T v = ggml_view_3d(c, projected, projected->ne[0] / 3, projected->ne[1], projected->ne[2], projected->nb[1], projected->nb[2], projected->nb[1] * 2 / 3);
    v = ggml_cont(c, v);
    v = ggml_reshape_4d(c, v, v->ne[0] / H, H, v->ne[1], v->ne[2]);
    v = ggml_permute(c, v, 0, 2, 1, 3);
    q = ggml_reshape_4d(c, q, q->ne[0] / H, H, q->ne[1], q->ne[2]);
    q = ggml_permute(c, q, 0, 2, 1, 3);
    if (rot) apply_rope(c, q, k, *rot);
    // moshi_kv_cache_insert_kv, tensor-index overload (transformer.h:238-249)
    k = ggml_set_rows(c, L.kcache, k, indices);
    v = ggml_set_rows(c, L.vcache, v, indices);
    // torch_nn_functional_scaled_dot_product_attention_custom (torch.h:225-237)
    const float scale = 1.f / sqrtf((float) q->ne[0]);
    T w = ggml_mul_mat(c, k, q);
    w = ggml_soft_max_ext(c, w, attn_bias, scale, 0.0f);
    v = ggml_cont(c, ggml_transpose(c, v));
    T o = ggml_mul_mat(c, v, w);
    T o2 = ggml_cont(c, ggml_permute(c, o, 0, 2, 1, 3));
    o = ggml_reshape_3d(c, o2, o2->ne[0] * o2->ne[1], o2->ne[2], o2->ne[3]);
    if (per_step_views) o = ggml_view_3d(c, o, o->ne[0], 1, o->ne[2], o->nb[1], o->nb[2], 0);
    return linear(c, L.out_proj[(size_t) wi], o);
}

// moshi_activation_gating (gating.h:11-37)
T gating(Builder & c, T w_in, T w_out, T x) {
    x = linear(c, w_in, x);
    T left = ggml_view_4d(c, x, x->ne[0] / 2, 1, x->ne[1], x->ne[2], x->nb[1] / 2, x->nb[1], x->nb[2], 0);
    T right = ggml_view_4d(c, x, x->ne[0] / 2, 1, x->ne[1], x->ne[2], x->nb[1] / 2, x->nb[1], x->nb[2], x->nb[1] / 2);
    left = ggml_silu(c, left);
    x = ggml_mul(c, left, right);
    return linear(c, w_out, x);
}

// torch_nn_linear_view (torch.h:103-118): rows [offset, offset + width) of a (quantised) weight
T linear_view(Builder & c, T w, int offset, int width, T x) {
    return ggml_mul_mat(c, ggml_view_2d(c, w, w->ne[0], width, w->nb[1], w->nb[1] * (size_t) offset), x);
}

// moshi_streaming_multihead_cross_attention (transformer.h:714-762): q = first third of in_proj, K/V cached, no mask
T cross_attention(Builder & c, const Transformer & tr, Layer & L, T query) {
    const int H = tr.heads, dim = (int) L.cross_in->ne[1] / 3;
    T q = linear_view(c, L.cross_in, 0, dim, query);
    q = ggml_reshape_4d(c, q, q->ne[0] / H, H, q->ne[1], q->ne[2]);
    q = ggml_permute(c, q, 0, 2, 1, 3);
    const float scale = 1.f / sqrtf((float) q->ne[0]);
    T w = ggml_mul_mat(c, L.k_cross, q);
    w = ggml_soft_max_ext(c, w, nullptr, scale, 0.0f);
    T v = ggml_cont(c, ggml_transpose(c, L.v_cross));
    T o = ggml_mul_mat(c, v, w);
    T o2 = ggml_cont(c, ggml_permute(c, o, 0, 2, 1, 3));
    o = ggml_reshape_3d(c, o2, o2->ne[0] * o2->ne[1], o2->ne[2], o2->ne[3]);
    return linear(c, L.cross_out, o);
}

// init() of the cross-attention state (transformer.h:343-396): K/V = last two thirds of in_proj applied to condition_cross
void init_cross(Builder & s, const Transformer & tr, Layer & L, T condition_cross) {
    const int H = tr.heads, dim = (int) L.cross_in->ne[1] / 3;
    T kv = linear_view(s, L.cross_in, dim, 2 * dim, condition_cross);
    auto half = [&](size_t off) {
        T t = ggml_view_3d(s, kv, kv->ne[0] / 2, kv->ne[1], kv->ne[2], kv->nb[1], kv->nb[2], off);
        t = ggml_cont(s, t);
        t = ggml_reshape_4d(s, t, t->ne[0] / H, H, t->ne[1], t->ne[2]);
        return ggml_permute(s, t, 0, 2, 1, 3);
    };
    s.expand(ggml_cpy(s, half(0), L.k_cross));
    s.expand(ggml_cpy(s, half(kv->nb[1] / 2), L.v_cross));
    s.compute_scratch();
}

// moshi_streaming_transformer_layer (transformer.h:910-1039)
T transformer_layer(Builder & c, const Transformer & tr, Layer & L, int wi, T indices, T x, T attn_bias, const Rot * rot, bool per_step_views) {
    T nx = apply_norm(c, L.norm1, x);
    T update = attention(c, tr, L, wi, indices, nx, attn_bias, rot, per_step_views);
    if (L.layer_scale_1) update = ggml_mul(c, update, L.layer_scale_1);
    x = ggml_add(c, x, update);
    if (L.cross_in) {   // transformer.h:936-944
        nx = apply_norm(c, L.norm_cross, x);
        x = ggml_add(c, x, cross_attention(c, tr, L, nx));
    }
    nx = apply_norm(c, L.norm2, x);
    if (L.gate_in.empty()) {
        T h = linear(c, L.linear1, nx);
        h = ggml_gelu(c, h);
        update = linear(c, L.linear2, h);
    } else {
        T gx = nx;
        if (per_step_views) gx = ggml_view_3d(c, nx, nx->ne[0], 1, nx->ne[2], nx->nb[1], nx->nb[2], 0);   // transformer.h:132-138
        update = gating(c, L.gate_in[(size_t) wi], L.gate_out[(size_t) wi], gx);
        // moshi_activation_gating returns [dim, 1, T]: addable to x only for T = 1, the reference's only LM shape; the batched prefill folds it back
        if (x->ne[1] > 1) update = ggml_reshape_3d(c, update, update->ne[0], x->ne[1], x->ne[2]);
    }
    if (L.layer_scale_2) update = ggml_mul(c, update, L.layer_scale_2);
    return ggml_add(c, x, update);
}

// moshi_streaming_transformer_graph_build (transformer.h:1217-1257): inputs live in the cached graph
T transformer_graph_build(Builder & g, Transformer & tr, T x) {
    const int Tn = (int) x->ne[1];
    create_bias_pattern(g.be, tr, Tn);
    tr.g_bias = g.tensor(GGML_TYPE_F32, tr.capacity, Tn);
    Rot rot;
    if (tr.max_period) {
        tr.g_offset = g.tensor(GGML_TYPE_F32, 1);
        rot = timestep_embedding(g, Tn, tr.dim / tr.heads, tr.g_offset, tr.max_period);
    }
    tr.g_indices = g.tensor(GGML_TYPE_I32, Tn);
    for (auto & L : tr.layers) x = transformer_layer(g, tr, L, 0, tr.g_indices, x, tr.g_bias, tr.max_period ? &rot : nullptr, false);
    return x;
}

// moshi_streaming_transformer_graph_step (transformer.h:1259-1289): refresh mask, rope offset, ring slots
void transformer_graph_step(Builder & scratch, Transformer & tr, int Tn) {
    const int offset = tr.offset;
    tr.offset += Tn;
    T bias = bias_pattern_index(scratch, tr, Tn, offset);
    scratch.expand(ggml_cpy(scratch, bias, tr.g_bias));
    if (tr.g_offset) { const float f = (float) offset; ggml_backend_tensor_set(tr.g_offset, &f, 0, 4); }
    std::vector<int32_t> idx((size_t) Tn);
    for (int i = 0; i < Tn; i++) idx[(size_t) i] = (offset + i) % tr.capacity;
    ggml_backend_tensor_set(tr.g_indices, idx.data(), 0, idx.size() * 4);
}

// moshi_streaming_transformer, non-graph overload with offsets baked at build time (transformer.h:1182-1215);
// this is how the Depth steps sit inside one cached graph (lm.h:469-470)
T transformer_inline(Builder & g, Transformer & tr, T x, T mask_override = nullptr) {
    const int Tn = (int) x->ne[1];
    const int offset = tr.offset;
    T attn_bias = mask_override ? mask_override : bias_pattern_index(g, tr, Tn, offset);
    Rot rot;
    if (tr.max_period) rot = timestep_embedding(g, Tn, tr.dim / tr.heads, g.f32((float) offset), tr.max_period);
    std::vector<int32_t> idx((size_t) Tn);
    for (int i = 0; i < Tn; i++) idx[(size_t) i] = (offset + i) % tr.capacity;
    T indices = g.i32s(idx);
    const bool multi = tr.layers[0].in_proj.size() > 1;
    const int wi = !multi ? 0 : tr.schedule.empty() ? offset : tr.schedule[(size_t) offset];
    for (auto & L : tr.layers) x = transformer_layer(g, tr, L, wi, indices, x, attn_bias, tr.max_period ? &rot : nullptr, multi);
    tr.offset += Tn;
    return x;
}

// moshi_sample_token (moshi/utils/sampling.h:4-64)
T sample_token(Builder & g, T logits, float temp, int top_k) {
    if (!(temp > 0.f)) return ggml_argmax(g, logits);
    T probs = ggml_soft_max(g, ggml_scale(g, logits, 1.f / temp));
    const int k = (int) probs->ne[0] < top_k ? (int) probs->ne[0] : top_k;
    T indices = ggml_argsort_top_k(g, probs, k);
    T rows = ggml_permute(g, probs, 1, 0, 2, 3);
    rows = ggml_get_rows(g, ggml_cont(g, rows), indices);
    probs = ggml_permute(g, rows, 1, 0, 2, 3);
    T in2 = ggml_reshape_2d(g, probs, probs->ne[0], probs->ne[1] * probs->ne[2] * probs->ne[3]);
    T q = ggml_div(g, in2, g.exponential(in2->ne[0], in2->ne[1], 1.f));
    T next = ggml_argmax(g, q);
    next = ggml_reshape_4d(g, next, next->ne[0], probs->ne[1], probs->ne[2], probs->ne[3]);
    T irows = ggml_permute(g, indices, 1, 0, 2, 3);
    return ggml_get_rows(g, ggml_cont(g, irows), next);
}

// ---- Mimi ---------------------------------------------------------------------------------------------------
struct Conv { int cin, cout, k, stride; T w = nullptr, b = nullptr; T prev = nullptr; };          // streaming / stateless conv1d
struct ConvTr { int cin, cout, k, stride, groups; T w = nullptr, b = nullptr; T prev = nullptr; };
struct ResBlock { Conv block1, block3; };

// moshi_streaming_conv_1d (moshi/modules/conv.h:50-96)
T streaming_conv(Builder & g, Conv & cv, T x) {
    const int TP = cv.k - cv.stride;
    x = ggml_concat(g, cv.prev, x, 0);
    T tail = ggml_view_3d(g, x, TP, x->ne[1], x->ne[2], x->nb[1], x->nb[2], x->nb[0] * (size_t) (x->ne[0] - TP));
    g.expand(ggml_cpy(g, tail, cv.prev));
    T y = ggml_conv_1d(g, cv.w, x, cv.stride, 0, 1);
    if (cv.b) y = ggml_add(g, y, cv.b);
    return y;
}
// moshi_stateless_conv_1d (conv.h:137-161)
T stateless_conv(Builder & g, Conv & cv, T x) {
    T y = ggml_conv_1d(g, cv.w, x, 1, 0, 1);
    if (cv.b) y = ggml_add(g, y, cv.b);
    return y;
}
// moshi_streaming_conv_transpose_1d (conv.h:240-310)
T streaming_conv_transpose(Builder & g, ConvTr & ct, T x) {
    const int PT = ct.k - ct.stride;
    T y = nullptr;
    if (ct.groups == 1) y = ggml_conv_transpose_1d(g, ct.w, x, ct.stride, 0, 1);
    else {
        for (int i = 0; i < ct.w->ne[0]; i++) {   // depthwise: one multiply per kernel tap (conv.h:262-278)
            T sub = ggml_view_3d(g, ct.w, 1, ct.w->ne[2], ct.w->ne[1], ct.w->nb[2], ct.w->nb[2], ct.w->nb[0] * (size_t) i);
            T piece = ggml_mul(g, x, sub);
            y = y ? ggml_concat(g, y, piece, 0) : piece;
        }
    }
    T prev = ct.prev;
    T partial = ggml_view_3d(g, prev, PT, prev->ne[1], prev->ne[2], prev->nb[1], prev->nb[2], prev->nb[0] * (size_t) (prev->ne[0] - PT));
    T lower = ggml_view_3d(g, y, PT, y->ne[1], y->ne[2], y->nb[1], y->nb[2], 0);
    lower = ggml_add_inplace(g, lower, partial);
    y = ggml_view_3d(g, lower, y->ne[0], y->ne[1], y->ne[2], y->nb[1], y->nb[2], 0);
    y = ggml_cpy(g, y, prev);
    if (ct.b) y = ggml_add(g, y, ct.b);
    y = ggml_view_3d(g, y, y->ne[0] - PT, y->ne[1], y->ne[2], y->nb[1], y->nb[2], 0);
    return ggml_cont(g, y);
}
// moshi_seanet_resnet_block (moshi/modules/seanet.h:14-27)
T resnet_block(Builder & g, ResBlock & rb, T x) {
    T v = ggml_elu(g, x);
    v = streaming_conv(g, rb.block1, v);
    v = ggml_elu(g, v);
    v = stateless_conv(g, rb.block3, v);
    return ggml_add(g, x, v);
}

struct Codebook { T embedding = nullptr; };
struct Rvq { std::vector<Codebook> layers; T input_proj = nullptr, output_proj = nullptr; int n_q = 0; };

// moshi_vq_decode (moshi/quantization/core_vq.h:100-109)
T vq_decode(Builder & g, Codebook & cb, T codes) {
    T q = ggml_get_rows(g, cb.embedding, ggml_cont(g, codes));
    q = ggml_permute(g, q, 1, 0, 2, 3);
    return ggml_cont(g, q);
}
// moshi_rvq_decode + moshi_residual_vq_decode (vq.h:18-30, core_vq.h:139-169)
T rvq_decode(Builder & g, Rvq & rvq, T codes) {
    codes = ggml_permute(g, codes, 0, 2, 1, 3);
    const int64_t Tn = codes->ne[0], B = codes->ne[1], K = codes->ne[2];
    T quantized = nullptr;
    for (size_t i = 0; i < rvq.layers.size() && (int64_t) i < K; i++) {
        T lc = ggml_view_3d(g, codes, Tn, B, 1, codes->nb[1], codes->nb[2], codes->nb[2] * i);
        T dec = vq_decode(g, rvq.layers[i], lc);
        quantized = quantized ? ggml_add(g, quantized, dec) : dec;
    }
    if (rvq.output_proj) quantized = ggml_conv_1d(g, rvq.output_proj, quantized, 1, 0, 1);
    return quantized;
}
// moshi_EuclideanCodebook_encode (core_vq.h:27-56): nearest centroid via the materialised difference
T codebook_encode(Builder & g, Codebook & cb, T x) {
    T a = ggml_cont(g, x);
    T b = cb.embedding;
    const int64_t ane1 = a->ne[1], bne1 = b->ne[1];
    a = ggml_reshape_3d(g, a, a->ne[0], 1, a->ne[1]);
    a = ggml_repeat_4d(g, a, a->ne[0], bne1, a->ne[2], 1);
    a = ggml_reshape_3d(g, a, a->ne[0], a->ne[1] * a->ne[2], a->ne[3]);
    b = ggml_repeat_4d(g, b, b->ne[0], b->ne[1] * ane1, b->ne[2], b->ne[3]);
    T d = ggml_sub(g, b, a);
    d = ggml_mul(g, d, d);
    d = ggml_sum_rows(g, d);
    d = ggml_reshape_3d(g, d, bne1, ane1, 1);
    d = ggml_add(g, d, g.f32(1.f));
    d = ggml_div(g, g.fill(d->ne, 1.f), d);
    return ggml_argmax(g, d);
}
// moshi_rvq_encode + moshi_residual_vq_encode (vq.h:32-45, core_vq.h:171-194)
T rvq_encode(Builder & g, Rvq & rvq, T x, T * latent = nullptr) {
    x = ggml_conv_1d(g, rvq.input_proj, x, 1, 0, 1);
    if (latent) *latent = x;
    T residual = x, out = nullptr;
    for (int i = 0; i < rvq.n_q; i++) {
        T xp = ggml_permute(g, residual, 1, 0, 2, 3);
        T idx = codebook_encode(g, rvq.layers[(size_t) i], xp);
        T quant = vq_decode(g, rvq.layers[(size_t) i], idx);
        idx = ggml_cast(g, idx, GGML_TYPE_F32);
        residual = ggml_sub(g, residual, quant);
        out = out ? ggml_concat(g, out, idx, 2) : idx;
    }
    return ggml_permute(g, out, 0, 2, 1, 3);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------
struct moshi_hot_model {
    moshi_hot_config cfg;
    ggml_backend_t be;
    ggml_backend_t be_codec = nullptr; bool own_codec_be = false;   // codec_stream: the Mimi graphs run on a second command stream of the same GPU
    Weights * W = nullptr;
    // persistent state (StateContext, src/context.h:656-780)
    struct ggml_context * st_ctx = nullptr; ggml_backend_buffer_t st_buf = nullptr;
    std::vector<std::pair<T, std::vector<uint8_t>>> st_init;
    Builder * scratch = nullptr, * scratch_codec = nullptr;

    // LM
    Transformer temporal, depth;
    Norm out_norm;
    T text_emb = nullptr, text_linear = nullptr;
    std::vector<T> emb, depformer_in, depformer_emb, linears;
    T depformer_text_emb = nullptr;
    std::vector<T> extra_heads;
    // tts variants
    T text_out1 = nullptr, text_out2 = nullptr, dep_text_out1 = nullptr, dep_text_out2 = nullptr;   // demux (lm_utils.h:48-85)
    T dep_text_low_rank = nullptr; std::vector<T> depformer_emb_low_rank;                          // low-rank embeddings (lm_utils.h:157-217)
    T cond_sum = nullptr, cond_cross = nullptr;
    T emb_right_idx = nullptr, emb_right_scale = nullptr, dep_right_idx = nullptr, dep_right_scale = nullptr;
    moshi_hot_text_hook_t text_hook = nullptr; void * text_hook_user = nullptr;
    T transformer_out = nullptr;   // state F32[dim] (lm.h:434)
    // chain_depth: state I32[1 + dep_q] = the tokens sampled by the last step (text, then the Depth chain's), handed between the graphs on the device the way
    // transformer_out is: the Depth graph takes its text index from [0] and samples into [1..]; the Temporal graph's embedding indices of those codebooks
    // are views of it, so the next step can be queued before the host has seen the tokens (run-ahead)
    T tok_state = nullptr;
    // run-ahead (chain_depth = 2): steps queued but not yet completed on the host, oldest first
    struct InFlight { bool done = false; bool steady = false; int ok = 0; std::vector<int32_t> raw; int32_t out_text = 0; std::vector<int32_t> out_audio; ggml_backend_event_t ev = nullptr; };
    std::deque<InFlight> inflight;
    std::vector<ggml_backend_event_t> ev_pool;
    int tok_state_for = -1;        // the step whose model-side inputs tok_state holds (the samples of step tok_state_for - 1), -1: unknown (provided / forced frames)
    Builder * g_temporal = nullptr; std::vector<T> emb_idx, emb_scale; T sampler_out = nullptr, text_logits = nullptr, g_transformer_out = nullptr, g_transformer_in = nullptr, g_stack_out = nullptr;
    Builder * g_depth = nullptr; T dep_text_idx = nullptr, dep_text_scale = nullptr, dep_tokens = nullptr; std::vector<T> dep_logits;
    // Depth codebook shard (moshi_hot.h): per-step graphs, import graphs, the two messages, the device-side token vector
    std::vector<Builder *> g_shard_step, g_shard_import; Builder * g_shard_begin = nullptr;
    T shard_msg = nullptr, shard_tout = nullptr, shard_tokens = nullptr;
    std::vector<T> shard_text_idx, shard_text_scale;
    moshi_hot_depth_hook_t depth_hook = nullptr; void * depth_hook_user = nullptr;
    // Depth codebook shard: transport of the per-step / per-frame messages (RCCL opened at run time, or a caller's function)
    moshi_hot_bcast_t shard_bcast = nullptr; void * shard_bcast_user = nullptr;
    void * rccl_lib = nullptr, * rccl_comm = nullptr;
    int64_t shard_hops = 0;   // broadcasts issued so far
    int (*rccl_broadcast)(const void *, void *, size_t, int, int, void *, void *) = nullptr;
    int (*rccl_comm_destroy)(void *) = nullptr;
    int (*rccl_all_reduce)(const void *, void *, size_t, int, int, void *, void *) = nullptr;
    moshi_hot_allreduce_t tp_allreduce = nullptr; void * tp_allreduce_user = nullptr;
    int64_t tp_reductions = 0;
    // tensor-parallel Temporal stack (moshi_hot.h): this rank's sliced layers, the replicated stream x, the partial message, one graph per segment
    Transformer temporal_tp; T tp_x = nullptr, tp_msg = nullptr; std::vector<Builder *> g_tp;
    // tensor-parallel FRAME mode (moshi_hot_tp_install): the Temporal half of an LM step = embedding-sum graph -> broadcast of x -> the stack above -> head graph
    bool tp_frame = false; T tp_in = nullptr; Builder * g_tp_pre = nullptr, * g_tp_import = nullptr, * g_tp_post = nullptr; int64_t tp_frames = 0;
    // delay ring (lm.h:715-743)
    int offset = 0; std::vector<std::vector<int>> cache; std::vector<int> initial; int max_delay = 0;

    // Mimi
    Rvq rvq_first, rvq_rest;
    ConvTr upsample; Conv downsample;
    Transformer dec_tr, enc_tr;
    std::vector<Conv> dec_convs; std::vector<ConvTr> dec_convtrs; std::vector<ResBlock> dec_res;
    std::vector<Conv> enc_convs; std::vector<ResBlock> enc_res;
    Builder * g_dec = nullptr; T dec_codes = nullptr, dec_frame = nullptr; int dec_T = 0;
    Builder * g_enc = nullptr; T enc_frame = nullptr, enc_codes = nullptr; int enc_T = 0;
    T enc_latent[2] = { nullptr, nullptr };   // the projected latents the two RVQ stacks quantise (inputs of their first levels), for tie analysis in tests

    std::vector<int32_t> tokens_tmp;
    // software-pipelined frame loop (moshi_hot_sts_pipeline_*): codes of the frame the next LM step consumes, tokens of the frame still to be decoded
    std::vector<int32_t> pipe_codes, pipe_tokens; bool pipe_have_codes = false, pipe_have_tokens = false; float pipe_vad = 0.f;
    bool temporal_staged = false;                  // chain_depth: the next frame's Temporal step inputs (mask row, RoPE phase, ring slot) are already queued
    std::function<void()> after_temporal_launch;   // runs once the Temporal graph is queued, before its text token is waited for
    int32_t last_text = 0; std::vector<int32_t> last_audio;   // raw (un-delayed) tokens of the last step
    // optional per-phase wall-clock (moshi_hot_set_timing): 0 mimi encode, 1 temporal, 2 depth, 3 mimi decode
    bool timing = false; double phase_us[4] = { 0, 0, 0, 0 }; int64_t phase_n[4] = { 0, 0, 0, 0 };
};

namespace {

T state(moshi_hot_model * m, enum ggml_type type, int64_t n0, int64_t n1 = 1, int64_t n2 = 1) {
    T t = ggml_new_tensor_3d(m->st_ctx, type, n0, n1, n2);
    m->st_init.push_back({ t, std::vector<uint8_t>(ggml_nbytes(t), 0) });   // zero-filled (transformer.h:164-166, conv.h:112)
    return t;
}

// Depth codebook shard: does this model hold the weight set of Depth step k?
bool owns_step(const moshi_hot_config & c, int k) { return c.dep_shard_world <= 1 || k % c.dep_shard_world == c.dep_shard_rank; }

void make_transformer(moshi_hot_model * m, Transformer & tr, const std::string & name, int dim, int heads, int n_layers, int ffn_hidden,
                      int capacity, int max_period, int n_weight_sets, bool mimi_style, enum ggml_type wtype, int cross_len = 0, bool norms_only = false) {
    Weights & W = *m->W;
    tr.dim = dim; tr.heads = heads; tr.capacity = capacity; tr.max_period = max_period;
    tr.layers.resize((size_t) n_layers);
    const float s_in = 1.f / sqrtf((float) dim);
    const float upd = m->cfg.update_scale > 0.f ? m->cfg.update_scale : 1.f;   // contractive variant for tight full-width parity runs (moshi_hot.h)
    auto qgen = [](float sd) { return [sd](T t, Rng & r, std::vector<uint8_t> & o) { gen_quant(t, r, o, sd); }; };
    auto ones = [](T t, Rng &, std::vector<uint8_t> & o) { gen_const(t, 1.f, o); };
    auto zeros = [](T t, Rng &, std::vector<uint8_t> & o) { gen_const(t, 0.f, o); };
    for (int l = 0; l < n_layers; l++) {
        Layer & L = tr.layers[(size_t) l];
        const std::string p = name + ".layers." + std::to_string(l) + ".";
        if (mimi_style) {
            L.norm1 = { false, 1e-5f, W.add(p + "norm1.weight", GGML_TYPE_F32, dim, 1, 1, ones), W.add(p + "norm1.bias", GGML_TYPE_F32, dim, 1, 1, zeros) };
            L.norm2 = { false, 1e-5f, W.add(p + "norm2.weight", GGML_TYPE_F32, dim, 1, 1, ones), W.add(p + "norm2.bias", GGML_TYPE_F32, dim, 1, 1, zeros) };
            L.layer_scale_1 = W.add(p + "layer_scale_1.scale", GGML_TYPE_F32, dim, 1, 1, [](T t, Rng &, std::vector<uint8_t> & o) { gen_const(t, 0.5f, o); });
            L.layer_scale_2 = W.add(p + "layer_scale_2.scale", GGML_TYPE_F32, dim, 1, 1, [](T t, Rng &, std::vector<uint8_t> & o) { gen_const(t, 0.5f, o); });
            L.linear1 = W.add(p + "linear1.weight", wtype, dim, ffn_hidden, 1, qgen(s_in));
            L.linear2 = W.add(p + "linear2.weight", wtype, ffn_hidden, dim, 1, qgen(1.f / sqrtf((float) ffn_hidden)));
        } else {
            L.norm1 = { true, 1e-8f, W.add(p + "norm1.alpha", GGML_TYPE_F32, dim, 1, 1, ones), nullptr };
            L.norm2 = { true, 1e-8f, W.add(p + "norm2.alpha", GGML_TYPE_F32, dim, 1, 1, ones), nullptr };
        }
        if (norms_only) continue;   // a tensor-parallel rank (tp_world > 1) holds slices of the matrices and its own ring shard instead (moshi_hot_create)
        for (int w = 0; w < n_weight_sets; w++) {
            const std::string ws = n_weight_sets > 1 ? "." + std::to_string(w) : "";
            if (n_weight_sets > 1 && !owns_step(m->cfg, w)) {   // another rank's step: no weights here
                L.in_proj.push_back(nullptr); L.out_proj.push_back(nullptr);
                if (!mimi_style) { L.gate_in.push_back(nullptr); L.gate_out.push_back(nullptr); }
                continue;
            }
            L.in_proj.push_back(W.add(p + "self_attn.in_projs" + ws + ".weight", wtype, dim, 3 * dim, 1, qgen(s_in)));
            L.out_proj.push_back(W.add(p + "self_attn.out_projs" + ws + ".weight", wtype, dim, dim, 1, qgen(s_in * upd)));
            if (!mimi_style) {
                L.gate_in.push_back(W.add(p + "gating" + ws + ".linear_in.weight", wtype, dim, 2 * ffn_hidden, 1, qgen(s_in)));
                L.gate_out.push_back(W.add(p + "gating" + ws + ".linear_out.weight", wtype, ffn_hidden, dim, 1, qgen(upd / sqrtf((float) ffn_hidden))));
            }
        }
        L.kcache = state(m, GGML_TYPE_BF16, dim / heads, capacity, heads);
        L.vcache = state(m, GGML_TYPE_BF16, dim / heads, capacity, heads);
        if (cross_len > 0) {   // lm_default.h:20-35; states filled by init() (transformer.h:335-339)
            L.norm_cross = { false, 0.0f, W.add(p + "norm_cross.weight", GGML_TYPE_F32, dim, 1, 1, ones),
                             W.add(p + "norm_cross.bias", GGML_TYPE_F32, dim, 1, 1, [](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, 0.02f); }) };
            L.cross_in = W.add(p + "cross_attention.in_projs.0.weight", wtype, dim, 3 * dim, 1, qgen(s_in));
            L.cross_out = W.add(p + "cross_attention.out_projs.0.weight", wtype, dim, dim, 1, qgen(s_in * upd));   // a residual update like out_proj / linear_out
            L.k_cross = state(m, GGML_TYPE_F32, dim / heads, cross_len, heads);
            L.v_cross = state(m, GGML_TYPE_F32, dim / heads, cross_len, heads);
        }
    }
}

Conv make_conv(moshi_hot_model * m, const std::string & name, int cin, int cout, int k, int stride, bool bias, bool stateful) {
    Weights & W = *m->W;
    Conv c = { cin, cout, k, stride };
    const float sd = 1.f / sqrtf((float) (cin * k));
    c.w = W.add(name + ".weight", GGML_TYPE_F16, k, cin, cout, [sd](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, sd); });   // loader.h:209: conv weights are F16
    if (bias) c.b = W.add(name + ".bias", GGML_TYPE_F32, 1, cout, 1, [](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, 0.02f); });
    if (stateful && k - stride > 0) c.prev = state(m, GGML_TYPE_F32, k - stride, cin);
    return c;
}
ConvTr make_convtr(moshi_hot_model * m, const std::string & name, int cin, int cout, int k, int stride, int groups, bool bias, int64_t in_len) {
    Weights & W = *m->W;
    ConvTr c = { cin, cout, k, stride, groups };
    const float sd = 1.f / sqrtf((float) (cin / groups));
    c.w = W.add(name + ".weight", GGML_TYPE_F32, k, cout / groups, cin, [sd](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, sd); });
    if (bias) c.b = W.add(name + ".bias", GGML_TYPE_F32, 1, cout, 1, [](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, 0.02f); });
    c.prev = state(m, GGML_TYPE_F32, (in_len - 1) * stride + k, cout);   // conv.h:205-217
    return c;
}
ResBlock make_res(moshi_hot_model * m, const std::string & name, int dim) {
    ResBlock rb;
    rb.block1 = make_conv(m, name + ".block.1.conv", dim, dim / 2, 3, 1, true, true);
    rb.block3 = make_conv(m, name + ".block.3.conv", dim / 2, dim, 1, 1, true, false);
    return rb;
}
void make_rvq(moshi_hot_model * m, Rvq & rvq, const std::string & name, int n_layers, int card) {
    Weights & W = *m->W;
    rvq.n_q = n_layers;
    for (int i = 0; i < n_layers; i++) {
        Codebook cb;
        cb.embedding = W.add(name + ".vq.layers." + std::to_string(i) + "._codebook.embedding", GGML_TYPE_F32, 256, card, 1,
                             [](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, 1.f); });
        rvq.layers.push_back(cb);
    }
    rvq.input_proj = W.add(name + ".input_proj.weight", GGML_TYPE_F16, 1, 512, 256, [](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, 1.f / sqrtf(512.f)); });
    rvq.output_proj = W.add(name + ".output_proj.weight", GGML_TYPE_F16, 1, 256, 512, [](T t, Rng & r, std::vector<uint8_t> & o) { gen_normal(t, r, o, 1.f / 16.f); });
}

// moshi_lmmodel_forward_text_build + sampler (lm.h:555-584, 659-677, 853-869)
T build_input_embedding(moshi_hot_model * m, Builder & g) {
    const moshi_hot_config & c = m->cfg;
    auto embed = [&](T table) {   // moshi_scaled_embedding_build (lm_utils.h:157-170)
        const int i = (int) m->emb_idx.size();
        T idx = m->tok_state && i <= c.dep_q ? ggml_view_1d(g, m->tok_state, 1, (size_t) i * 4) : g.tensor(GGML_TYPE_I32, 1), scale = g.tensor(GGML_TYPE_F32, 1);
        m->emb_idx.push_back(idx); m->emb_scale.push_back(scale);
        return ggml_mul(g, ggml_get_rows(g, table, idx), scale);
    };
    T input;
    if (c.demux_second_stream) {   // moshi_scaled_embedding_demux_build (lm_utils.h:48-66)
        T left = g.tensor(GGML_TYPE_I32, 1);
        m->emb_idx.push_back(left); m->emb_scale.push_back(nullptr);
        m->emb_right_idx = g.tensor(GGML_TYPE_I32, 1); m->emb_right_scale = g.tensor(GGML_TYPE_F32, 1);
        T l = ggml_get_rows(g, m->text_emb, left), r = ggml_get_rows(g, m->text_emb, m->emb_right_idx);
        T ry = linear(g, m->text_out2, r), ly = linear(g, m->text_out1, l);
        input = ggml_add(g, ly, ggml_mul(g, ry, m->emb_right_scale));
    } else input = embed(m->text_emb);
    for (int k = 0; k < c.n_q; k++) input = ggml_add(g, input, embed(m->emb[(size_t) k]));
    if (c.condition_sum) input = ggml_add(g, m->cond_sum, input);   // lm.h:579-581
    m->g_transformer_in = input;
    return input;
}
void build_temporal_graph(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    m->g_temporal = new Builder(m->be, 256);
    Builder & g = *m->g_temporal;
    T input = build_input_embedding(m, g);
    T x = transformer_graph_build(g, m->temporal, input);
    m->g_stack_out = x;
    x = apply_norm(g, m->out_norm, x);
    m->g_transformer_out = x;
    m->text_logits = linear(g, m->text_linear, x);
    g.expand(ggml_cpy(g, x, m->transformer_out));
    m->sampler_out = sample_token(g, m->text_logits, c.temp_text, c.top_k_text);
    g.expand(m->sampler_out);
    if (m->tok_state) g.expand(ggml_cpy(g, ggml_reshape_1d(g, m->sampler_out, 1), ggml_view_1d(g, m->tok_state, 1, 0)));
    g.alloc();
}

// moshi_lmmodel_depformer_step, graph part (lm.h:489-531)
void build_depth_graph(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    m->g_depth = new Builder(m->be, 256);
    Builder & g = *m->g_depth;
    m->dep_text_idx = g.tensor(GGML_TYPE_I32, 1);
    if (c.chain_depth) {
        // the text token is taken from the state tensor the Temporal graph copies its sample into: no host round trip between the graphs
        GGML_ASSERT(!c.demux_second_stream && !c.delay_steps && m->tok_state && m->sampler_out->type == GGML_TYPE_I32 && ggml_nelements(m->sampler_out) == 1);
        m->dep_text_idx = ggml_view_1d(g, m->tok_state, 1, 0);
    }
    T last;
    if (c.demux_second_stream) {
        m->dep_right_idx = g.tensor(GGML_TYPE_I32, 1); m->dep_right_scale = g.tensor(GGML_TYPE_F32, 1);
        T l = ggml_get_rows(g, m->depformer_text_emb, m->dep_text_idx), r = ggml_get_rows(g, m->depformer_text_emb, m->dep_right_idx);
        T ry = linear(g, m->dep_text_out2, r), ly = linear(g, m->dep_text_out1, l);
        last = ggml_add(g, ly, ggml_mul(g, ry, m->dep_right_scale));
    } else {
        m->dep_text_scale = g.tensor(GGML_TYPE_F32, 1);
        last = ggml_mul(g, ggml_get_rows(g, m->depformer_text_emb, m->dep_text_idx), m->dep_text_scale);
        if (m->dep_text_low_rank) last = linear(g, m->dep_text_low_rank, last);
    }
    T tokens = m->tok_state ? ggml_view_1d(g, m->tok_state, c.dep_q, 4) : g.tensor(GGML_TYPE_I32, c.dep_q);
    T view = nullptr, next = nullptr;
    for (int k = 0; k < c.dep_q; k++) {
        if (k > 0) {   // moshi_scaled_embedding_chained (lm_utils.h:208-217)
            last = ggml_get_rows(g, m->depformer_emb[(size_t) (k - 1)], next);
            if (!m->depformer_emb_low_rank.empty()) last = linear(g, m->depformer_emb_low_rank[(size_t) (k - 1)], last);
        }
        // moshi_lmmodel_forward_depformer_transform (lm.h:446-475)
        const int in_index = c.dep_schedule_len ? c.dep_schedule[k] : k;
        T din = linear(g, m->depformer_in[(size_t) in_index], m->transformer_out);
        last = ggml_cast(g, last, GGML_TYPE_F32);
        din = ggml_add(g, din, last);
        T dout = transformer_inline(g, m->depth, din);
        T logits = linear(g, m->linears[(size_t) k], dout);
        m->dep_logits.push_back(logits);
        next = sample_token(g, logits, c.temp, c.top_k);
        view = k == 0 ? ggml_view_1d(g, tokens, 1, 0) : ggml_view_1d(g, view, 1, 4);
        g.expand(ggml_cpy(g, next, view));
    }
    m->dep_tokens = ggml_view_1d(g, view, tokens->ne[0], (size_t) (-(int64_t) (c.dep_q - 1) * 4));
    g.expand(m->dep_tokens);
    g.alloc();
}

// mimi_decode, graph part (compression.h:156-187)
void build_decode_graph(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    m->g_dec = new Builder(m->be_codec, 256);
    Builder & g = *m->g_dec;
    T codes = ggml_new_tensor_2d(g, GGML_TYPE_I32, 1, c.mimi_n_q);
    m->dec_codes = codes;
    // moshi_split_rvq_decode (vq.h:62-95)
    const int64_t B = codes->ne[2], K = codes->ne[1], Tn = codes->ne[0];
    T first = ggml_view_3d(g, codes, Tn, 1, B, codes->nb[1], codes->nb[2], 0);
    T emb = rvq_decode(g, m->rvq_first, first);
    if (K > 1) {
        T rest = ggml_view_3d(g, codes, Tn, K - 1, B, codes->nb[1], codes->nb[2], codes->nb[1] * 1);
        emb = ggml_add(g, emb, rvq_decode(g, m->rvq_rest, rest));
    }
    emb = streaming_conv_transpose(g, m->upsample, emb);
    m->dec_T = (int) emb->ne[0];
    // moshi_projected_transformer_graph_build (transformer.h:1356-1365)
    T x = ggml_cont(g, ggml_transpose(g, emb));
    x = transformer_graph_build(g, m->dec_tr, x);
    x = ggml_transpose(g, x);
    // moshi_seanet_decoder (seanet.h:179-211)
    x = streaming_conv(g, m->dec_convs[0], x);
    x = ggml_elu(g, x);
    for (int i = 0; i < 4; i++) {
        x = streaming_conv_transpose(g, m->dec_convtrs[(size_t) i], x);
        x = resnet_block(g, m->dec_res[(size_t) i], x);
        x = ggml_elu(g, x);
    }
    x = streaming_conv(g, m->dec_convs[1], x);
    m->dec_frame = x;
    g.expand(x);
    g.alloc();
}

// mimi_encode, graph part (compression.h:284-308)
void build_encode_graph(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    m->g_enc = new Builder(m->be_codec, 256);
    Builder & g = *m->g_enc;
    T x = ggml_new_tensor_1d(g, GGML_TYPE_F32, 1920);
    m->enc_frame = x;
    // moshi_seanet_encoder (seanet.h:88-113)
    x = streaming_conv(g, m->enc_convs[0], x);
    for (int i = 0; i < 4; i++) {
        x = resnet_block(g, m->enc_res[(size_t) i], x);
        x = ggml_elu(g, x);
        x = streaming_conv(g, m->enc_convs[(size_t) (1 + i)], x);
    }
    x = ggml_elu(g, x);
    x = streaming_conv(g, m->enc_convs[5], x);
    m->enc_T = (int) x->ne[0];
    x = ggml_cont(g, ggml_transpose(g, x));
    x = transformer_graph_build(g, m->enc_tr, x);
    x = ggml_transpose(g, x);
    x = streaming_conv(g, m->downsample, x);
    // moshi_split_rvq_encode (vq.h:97-114)
    T codes = rvq_encode(g, m->rvq_first, x, &m->enc_latent[0]);
    if (c.mimi_n_q > 1) codes = ggml_concat(g, codes, rvq_encode(g, m->rvq_rest, x, &m->enc_latent[1]), 1);
    codes = ggml_cast(g, codes, GGML_TYPE_I32);
    m->enc_codes = codes;
    g.expand(codes);
    g.alloc();
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------
extern "C" void moshi_hot_config_moshika(struct moshi_hot_config * c) {
    memset(c, 0, sizeof(*c));
    c->dim = 4096; c->num_heads = 32; c->num_layers = 32; c->ffn_hidden = 11264; c->context = 3000; c->max_period = 10000;
    c->text_card = 32000; c->card = 2048; c->n_q = 16; c->dep_q = 8;
    const int d[17] = { 0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1 };
    for (int i = 0; i < 17; i++) c->delays[i] = d[i];
    c->dep_dim = 1024; c->dep_heads = 16; c->dep_layers = 6; c->dep_ffn_hidden = 2816; c->dep_context = 8;
    c->linear_type = GGML_TYPE_Q4_K; c->embed_type = GGML_TYPE_Q4_0;
    c->mimi_n_q = 8; c->mimi_codebook_size = 2048;
    c->enable_lm = 1; c->enable_mimi_encoder = 1; c->enable_mimi_decoder = 1;
    c->temp = 0.f; c->temp_text = 0.f; c->top_k = 250; c->top_k_text = 25;
}

extern "C" void moshi_hot_config_personaplex(struct moshi_hot_config * c) {
    moshi_hot_config_moshika(c);     // same widths; dep_q 16 chained Depth steps over a ring of 8 (tools/personaplex-config.json)
    c->dep_q = 16;
    c->personaplex = 1;
}

static moshi_hot_model_t * create_model(ggml_backend_t backend, const struct moshi_hot_config * cfg, uint64_t seed, const char * gguf_path);
extern "C" moshi_hot_model_t * moshi_hot_create(ggml_backend_t backend, const struct moshi_hot_config * cfg, uint64_t seed) { return create_model(backend, cfg, seed, nullptr); }
// the model's weights come from a GGUF file written by moshi_hot_save_gguf (the reference's `*.gguf` checkpoints: WeightLoader::from_gguf + load_gguf)
extern "C" moshi_hot_model_t * moshi_hot_create_from_gguf(ggml_backend_t backend, const struct moshi_hot_config * cfg, const char * path) { return create_model(backend, cfg, 0, path); }
// WeightLoader::save_gguf (loader.h:227-233): every tensor of the weight context, in context order
extern "C" int moshi_hot_save_gguf(moshi_hot_model_t * m, const char * path) {
    struct gguf_context * gg = gguf_init_empty();
    for (T t = ggml_get_first_tensor(m->W->ctx); t; t = ggml_get_next_tensor(m->W->ctx, t)) gguf_add_tensor(gg, t);
    const bool ok = gguf_write_to_file(gg, path, false);
    gguf_free(gg);
    return ok ? 1 : 0;
}
extern "C" int moshi_hot_tensor_file_name(const char * checkpoint_name, char * out, int n) {
    const std::string f = Weights::file_name(checkpoint_name);
    if (out && n > 0) { strncpy(out, f.c_str(), (size_t) n - 1); out[n - 1] = 0; }
    return (int) f.size();
}
static moshi_hot_model_t * create_model(ggml_backend_t backend, const struct moshi_hot_config * cfg, uint64_t seed, const char * gguf_path) {
    moshi_hot_model * m = new moshi_hot_model;
    m->cfg = *cfg;
    m->be = backend;
    const moshi_hot_config & c = m->cfg;
    m->W = new Weights(backend, seed, 4096);
    if (gguf_path) m->W->gguf_path = gguf_path;
    m->st_ctx = ggml_init({ ggml_tensor_overhead() * 1024, NULL, true });
    m->scratch = new Builder(backend, 16);
    m->be_codec = backend;
    if (c.codec_stream) {
        ggml_backend_t b2 = ggml_backend_mi355x_init_stream(backend);   // NULL on any other backend: one stream, same results
        if (b2) { m->be_codec = b2; m->own_codec_be = true; }
    }
    m->scratch_codec = m->own_codec_be ? new Builder(m->be_codec, 16) : m->scratch;
    Weights & W = *m->W;
    const enum ggml_type lt = (enum ggml_type) c.linear_type, et = (enum ggml_type) c.embed_type;
    auto qgen = [](float sd) { return [sd](T t, Rng & r, std::vector<uint8_t> & o) { gen_quant(t, r, o, sd); }; };
    auto ones = [](T t, Rng &, std::vector<uint8_t> & o) { gen_const(t, 1.f, o); };

    if (c.enable_lm) {
        W.part = 4;
        // WeightLoader type fall-back (loader.h:162-173): Q4_K needs ne0 % 256 == 0, else Q4_0 (ne0 % 32), else the source type
        auto fit = [](enum ggml_type t, int64_t ne0) {
            if (t == GGML_TYPE_Q4_K && ne0 % 256 != 0) t = GGML_TYPE_Q4_0;
            if ((t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0) && ne0 % 32 != 0) t = GGML_TYPE_F32;
            return t;
        };
        if (!c.depth_only) {   // (a Depth-only shard rank holds neither the embeddings nor the Temporal stack)
        m->text_emb = W.add("lm.text_emb.weight", et, c.dim, c.text_card + 1, 1, qgen(1.f));
        if (c.demux_second_stream) {
            m->text_out1 = W.add("lm.text_emb.out1.weight", lt, c.dim, c.dim, 1, qgen(1.f / sqrtf((float) c.dim)));
            m->text_out2 = W.add("lm.text_emb.out2.weight", lt, c.dim, c.dim, 1, qgen(1.f / sqrtf((float) c.dim)));
        }
        for (int k = 0; k < c.n_q; k++) m->emb.push_back(W.add("lm.emb." + std::to_string(k) + ".weight", et, c.dim, c.card + 1, 1, qgen(1.f)));
        W.part = 0;
        make_transformer(m, m->temporal, "lm.transformer", c.dim, c.num_heads, c.num_layers, c.ffn_hidden, c.context, c.max_period, 1, false, lt,
                         c.cross_attention ? c.cross_len : 0, c.tp_world > 1);
        if (c.tp_world > 1 || c.tp_world == 1) {
            // this rank's slices of every Temporal layer, cut from the very matrices the unsplit stack above is made of
            const int N = c.tp_world, r = c.tp_rank, H = c.num_heads, D = c.dim / H, Hr = H / N, F = c.ffn_hidden, Fr = F / N;
            GGML_ASSERT(H % N == 0 && F % N == 0 && (c.dim / N) % 256 == 0 && Fr % 256 == 0 && !c.cross_attention);
            Transformer & tp = m->temporal_tp;
            tp.dim = c.dim / N; tp.heads = Hr; tp.capacity = c.context; tp.max_period = c.max_period;
            tp.layers.resize((size_t) c.num_layers);
            const float s_in = 1.f / sqrtf((float) c.dim);
            const float upd = c.update_scale > 0.f ? c.update_scale : 1.f;
            for (int l = 0; l < c.num_layers; l++) {
                Layer & L = tp.layers[(size_t) l];
                const Layer & full = m->temporal.layers[(size_t) l];
                const std::string p = "lm.transformer.layers." + std::to_string(l) + ".";
                L.norm1 = full.norm1; L.norm2 = full.norm2;
                std::vector<int64_t> rows;
                for (int part = 0; part < 3; part++) for (int j = 0; j < Hr * D; j++) rows.push_back((int64_t) part * c.dim + (int64_t) r * Hr * D + j);
                L.in_proj.push_back(W.add_slice(p + "self_attn.in_projs.weight", lt, c.dim, 3 * c.dim, qgen(s_in), rows, 0, c.dim));
                rows.clear(); for (int j = 0; j < c.dim; j++) rows.push_back(j);
                L.out_proj.push_back(W.add_slice(p + "self_attn.out_projs.weight", lt, c.dim, c.dim, qgen(s_in * upd), rows, (int64_t) r * Hr * D, (int64_t) Hr * D));
                rows.clear(); for (int half = 0; half < 2; half++) for (int j = 0; j < Fr; j++) rows.push_back((int64_t) half * F + (int64_t) r * Fr + j);
                L.gate_in.push_back(W.add_slice(p + "gating.linear_in.weight", lt, c.dim, 2 * F, qgen(s_in), rows, 0, c.dim));
                rows.clear(); for (int j = 0; j < c.dim; j++) rows.push_back(j);
                L.gate_out.push_back(W.add_slice(p + "gating.linear_out.weight", lt, F, c.dim, qgen(upd / sqrtf((float) F)), rows, (int64_t) r * Fr, Fr));
                L.kcache = state(m, GGML_TYPE_BF16, D, c.context, Hr);
                L.vcache = state(m, GGML_TYPE_BF16, D, c.context, Hr);
            }
            m->tp_x = state(m, GGML_TYPE_F32, c.dim);
            m->tp_msg = state(m, GGML_TYPE_F32, c.dim);
            m->tp_in = state(m, GGML_TYPE_F32, (int64_t) c.dim + 8);   // frame mode: the stack input as rank 0 broadcasts it, then a "more frames" flag
            // the position inputs of transformer_graph_step are shared by all segment graphs: persistent tensors instead of per-graph inputs
            tp.g_bias = state(m, GGML_TYPE_F32, c.context);
            if (c.max_period) tp.g_offset = state(m, GGML_TYPE_F32, 1);
            tp.g_indices = state(m, GGML_TYPE_I32, 1);
        }
        m->out_norm = { true, 1e-8f, W.add("lm.out_norm.alpha", GGML_TYPE_F32, c.dim, 1, 1, ones), nullptr };
        m->text_linear = W.add("lm.text_linear.weight", lt, c.dim, c.text_card, 1, qgen(1.f / sqrtf((float) c.dim)));
        }
        m->transformer_out = state(m, GGML_TYPE_F32, c.dim);
        if (c.chain_depth) m->tok_state = state(m, GGML_TYPE_I32, 1 + c.dep_q);
        if (c.condition_sum) m->cond_sum = state(m, GGML_TYPE_F32, c.dim);
        if (c.cross_attention) m->cond_cross = state(m, GGML_TYPE_F32, c.dim, c.cross_len);
        for (int k = 0; k < c.extra_heads; k++)
            m->extra_heads.push_back(W.add("lm.extra_heads." + std::to_string(k) + ".weight", lt, c.dim, c.extra_heads_dim, 1, qgen(1.f / sqrtf((float) c.dim))));
        if (c.dep_q > 0) {
            // depformer_num_weights (lm_default.h:71-81): one set per step, or max(schedule) + 1
            int n_sets = c.dep_q;
            if (c.dep_schedule_len) { n_sets = 0; for (int i = 0; i < c.dep_schedule_len; i++) if (c.dep_schedule[i] + 1 > n_sets) n_sets = c.dep_schedule[i] + 1; }
            const int E = c.depformer_low_rank ? c.depformer_low_rank : c.dep_dim;   // embedding table width
            W.part = 1;
            GGML_ASSERT(c.dep_shard_world <= 1 || c.dep_schedule_len == 0);   // the shard is by step = by weight set
            for (int k = 0; k < n_sets; k++)
                m->depformer_in.push_back(!owns_step(c, k) ? nullptr : W.add("lm.depformer_in." + std::to_string(k) + ".weight", lt, c.dim, c.dep_dim, 1, qgen(1.f / sqrtf((float) c.dim))));
            for (int k = 0; k < c.dep_q; k++) {
                m->linears.push_back(!owns_step(c, k) ? nullptr : W.add("lm.linears." + std::to_string(k) + ".weight", lt, c.dep_dim, c.card, 1, qgen(1.f / sqrtf((float) c.dep_dim))));
                if (k > 0) {
                    W.part = 4;
                    m->depformer_emb.push_back(!owns_step(c, k) ? nullptr : W.add("lm.depformer_emb." + std::to_string(k - 1) + ".weight", fit(et, E), E, c.card + 1, 1, qgen(1.f)));
                    W.part = 1;
                    if (c.depformer_low_rank)
                        m->depformer_emb_low_rank.push_back(W.add("lm.depformer_emb." + std::to_string(k - 1) + ".low_rank.weight", fit(lt, E), E, c.dep_dim, 1, qgen(1.f / sqrtf((float) E))));
                }
            }
            W.part = 4;
            if (owns_step(c, 0)) m->depformer_text_emb = W.add("lm.depformer_text_emb.weight", fit(et, E), E, c.text_card + 1, 1, qgen(1.f));
            W.part = 1;
            if (c.demux_second_stream) {
                m->dep_text_out1 = W.add("lm.depformer_text_emb.out1.weight", fit(lt, E), E, c.dep_dim, 1, qgen(1.f / sqrtf((float) E)));
                m->dep_text_out2 = W.add("lm.depformer_text_emb.out2.weight", fit(lt, E), E, c.dep_dim, 1, qgen(1.f / sqrtf((float) E)));
            } else if (c.depformer_low_rank)
                m->dep_text_low_rank = W.add("lm.depformer_text_emb.low_rank.weight", fit(lt, E), E, c.dep_dim, 1, qgen(1.f / sqrtf((float) E)));
            // capacity = context ? context : weights_per_step (lm_default.h:86-90)
            make_transformer(m, m->depth, "lm.depformer", c.dep_dim, c.dep_heads, c.dep_layers, c.dep_ffn_hidden, c.dep_context ? c.dep_context : c.dep_schedule_len, 0, n_sets, false, lt);
            for (int i = 0; i < c.dep_schedule_len; i++) m->depth.schedule.push_back(c.dep_schedule[i]);
            if (c.dep_shard_world >= 1) {   // shard messages + the device-side token vector (zero-filled states)
                m->shard_msg = state(m, GGML_TYPE_F32, (int64_t) c.dep_layers * c.dep_dim + 8);   // 2 x layers x dim BF16 ring values, then the token (F32)
                m->shard_tout = state(m, GGML_TYPE_F32, (int64_t) c.dim + 8);
                m->shard_tokens = state(m, GGML_TYPE_I32, c.dep_q);
            }
        }
        // moshi_lmgen_state (lm.h:722-743)
        const int ncb = c.n_q + 1;
        for (int i = 0; i < ncb; i++) if (c.delays[i] > m->max_delay) m->max_delay = c.delays[i];
        m->cache.assign((size_t) (m->max_delay + 2 + (c.personaplex ? 1 : 0)), std::vector<int>((size_t) ncb, -2));
        m->initial.assign((size_t) ncb, c.card);
        m->initial[0] = c.text_card;
    }
    if (c.enable_mimi_decoder || c.enable_mimi_encoder) {
        W.part = c.enable_mimi_decoder ? 3 : 2;
        make_rvq(m, m->rvq_first, "mimi.quantizer.rvq_first", 1, c.mimi_codebook_size);
        make_rvq(m, m->rvq_rest, "mimi.quantizer.rvq_rest", c.mimi_n_q > 1 ? c.mimi_n_q - 1 : 0, c.mimi_codebook_size);
    }
    if (c.enable_mimi_decoder) {   // moshi_mimi_alloc_default, decoder half (lm_default.h:229-415); states: src/moshi.cpp:244-262
        W.part = 3;
        m->upsample = make_convtr(m, "mimi.upsample.convtr", 512, 512, 4, 2, 512, false, 1);
        make_transformer(m, m->dec_tr, "mimi.decoder_transformer.transformer", 512, 8, 8, 2048, 250, 10000, 1, true, GGML_TYPE_F32);
        m->dec_convs.push_back(make_conv(m, "mimi.decoder.model.0.conv", 512, 1024, 7, 1, true, true));
        const int ch[5] = { 1024, 512, 256, 128, 64 }, ks[4] = { 16, 12, 10, 8 }, st[4] = { 8, 6, 5, 4 };
        int64_t len = 2;
        for (int i = 0; i < 4; i++) {
            m->dec_convtrs.push_back(make_convtr(m, "mimi.decoder.model." + std::to_string(2 + 3 * i) + ".convtr", ch[i], ch[i + 1], ks[i], st[i], 1, true, len));
            len = len * st[i];
            m->dec_res.push_back(make_res(m, "mimi.decoder.model." + std::to_string(3 + 3 * i), ch[i + 1]));
        }
        m->dec_convs.push_back(make_conv(m, "mimi.decoder.model.14.conv", 64, 1, 3, 1, true, true));
    }
    if (c.enable_mimi_encoder) {   // encoder half (lm_default.h:416-560)
        W.part = 2;
        m->enc_convs.push_back(make_conv(m, "mimi.encoder.model.0.conv", 1, 64, 7, 1, true, true));
        const int ch[5] = { 64, 128, 256, 512, 1024 }, ks[4] = { 8, 10, 12, 16 }, st[4] = { 4, 5, 6, 8 };
        for (int i = 0; i < 4; i++) {
            m->enc_res.push_back(make_res(m, "mimi.encoder.model." + std::to_string(1 + 3 * i), ch[i]));
            m->enc_convs.push_back(make_conv(m, "mimi.encoder.model." + std::to_string(3 + 3 * i) + ".conv", ch[i], ch[i + 1], ks[i], st[i], true, true));
        }
        m->enc_convs.push_back(make_conv(m, "mimi.encoder.model.14.conv", 1024, 512, 3, 1, true, true));
        make_transformer(m, m->enc_tr, "mimi.encoder_transformer.transformer", 512, 8, 8, 2048, 250, 10000, 1, true, GGML_TYPE_F32);
        m->downsample = make_conv(m, "mimi.downsample.conv", 512, 512, 4, 2, false, true);
    }
    W.load();
    m->st_buf = ggml_backend_alloc_ctx_tensors(m->st_ctx, backend);
    GGML_ASSERT(m->st_buf);
    for (auto & s : m->st_init) ggml_backend_tensor_set(s.first, s.second.data(), 0, s.second.size());
    m->st_init.clear();
    m->tokens_tmp.resize((size_t) (c.n_q + 1 + 32));
    m->pipe_codes.resize((size_t) (c.n_q + 1 + 32)); m->pipe_tokens.resize((size_t) (c.n_q + 1 + 32));
    ggml_backend_synchronize(backend);   // weights and zeroed states are in place before any other stream touches them
    return m;
}

extern "C" void moshi_hot_free(moshi_hot_model_t * m) {
    if (!m) return;
    ggml_backend_synchronize(m->be);
    moshi_hot_depth_shard_rccl_free(m);
    for (auto & f : m->inflight) if (f.ev) ggml_backend_event_free(f.ev);
    for (auto e : m->ev_pool) ggml_backend_event_free(e);
    if (m->own_codec_be) ggml_backend_synchronize(m->be_codec);
    delete m->g_temporal; delete m->g_depth; delete m->g_dec; delete m->g_enc; if (m->scratch_codec != m->scratch) delete m->scratch_codec; delete m->scratch; delete m->g_shard_begin;
    for (auto * b : m->g_shard_step) delete b;
    for (auto * b : m->g_shard_import) delete b;
    for (auto * b : m->g_tp) delete b;
    delete m->g_tp_pre; delete m->g_tp_import; delete m->g_tp_post;
    if (m->st_buf) ggml_backend_buffer_free(m->st_buf);
    ggml_free(m->st_ctx);
    delete m->W;
    if (m->own_codec_be) ggml_backend_free(m->be_codec);
    delete m;
}

namespace {
struct PhaseTimer {
    moshi_hot_model * m; int phase; int64_t t0;
    ggml_backend_t be() const { return phase == 0 || phase == 3 ? m->be_codec : m->be; }
    PhaseTimer(moshi_hot_model * m_, int p) : m(m_), phase(p), t0(0) { if (m->timing) { ggml_backend_synchronize(be()); t0 = ggml_time_us(); } }
    ~PhaseTimer() { if (m->timing) { ggml_backend_synchronize(be()); m->phase_us[phase] += (double) (ggml_time_us() - t0); m->phase_n[phase]++; } }
};
}

// mimi_decode, per-frame part (compression.h:189-203), as launch (uploads + graphs queued on the codec stream, nothing waited for) and finish (read-back)
namespace {
void mimi_decode_launch(moshi_hot_model * m, const int32_t * codes) {
    if (!m->g_dec) build_decode_graph(m);
    ggml_backend_tensor_set(m->dec_codes, codes, 0, ggml_nbytes(m->dec_codes));
    transformer_graph_step(*m->scratch_codec, m->dec_tr, m->dec_T);
    m->scratch_codec->compute_scratch();
    m->g_dec->compute();
}
void mimi_decode_finish(moshi_hot_model * m, float * pcm) {
    if (pcm) ggml_backend_tensor_get(m->dec_frame, pcm, 0, ggml_nbytes(m->dec_frame));
    else { float tmp; ggml_backend_tensor_get(m->dec_frame, &tmp, 0, 4); }
}
// mimi_encode, per-frame part (compression.h:310-324)
void mimi_encode_launch(moshi_hot_model * m, const float * pcm) {
    if (!m->g_enc) build_encode_graph(m);
    ggml_backend_tensor_set(m->enc_frame, pcm, 0, ggml_nbytes(m->enc_frame));
    transformer_graph_step(*m->scratch_codec, m->enc_tr, m->enc_T);
    m->scratch_codec->compute_scratch();
    m->g_enc->compute();
}
void mimi_encode_finish(moshi_hot_model * m, int32_t * codes) { ggml_backend_tensor_get(m->enc_codes, codes, 0, ggml_nbytes(m->enc_codes)); }
}  // namespace

extern "C" void moshi_hot_mimi_decode(moshi_hot_model_t * m, const int32_t * codes, float * pcm) {
    PhaseTimer pt(m, 3);
    mimi_decode_launch(m, codes);
    mimi_decode_finish(m, pcm);
}
extern "C" void moshi_hot_mimi_encode(moshi_hot_model_t * m, const float * pcm, int32_t * codes) {
    PhaseTimer pt(m, 0);
    mimi_encode_launch(m, pcm);
    mimi_encode_finish(m, codes);
}

// moshi_lmgen_step (lm.h:778-979) for the plain moshi model: no state machine, no prefixes
namespace {
// moshi_scaled_embedding_demux_step (lm_utils.h:68-85)
void demux_set(const moshi_hot_config & c, int32_t input, T left, T right, T right_scale) {
    if (input < 0) input = 0;
    const int32_t n = c.text_card + 1;
    int32_t l = input % n, r = input / n - 1;
    const float sc = r < 0 ? 0.f : 1.f;
    if (r < 0) r = 0;
    ggml_backend_tensor_set(left, &l, 0, 4);
    ggml_backend_tensor_set(right, &r, 0, 4);
    ggml_backend_tensor_set(right_scale, &sc, 0, 4);
}
void depth_step(moshi_hot_model * m, int32_t text_token, std::vector<int32_t> & audio) {   // moshi_lmmodel_depformer_step (lm.h:532-552)
    PhaseTimer pt(m, 2);
    if (!m->g_depth) build_depth_graph(m);
    if (m->cfg.demux_second_stream) demux_set(m->cfg, text_token, m->dep_text_idx, m->dep_right_idx, m->dep_right_scale);
    else {
        int32_t id = text_token; const float sc = id == -1 ? 0.f : 1.f;
        if (id < 0) id = 0;
        if (!m->cfg.chain_depth) ggml_backend_tensor_set(m->dep_text_idx, &id, 0, 4);
        ggml_backend_tensor_set(m->dep_text_scale, &sc, 0, 4);
    }
    m->g_depth->compute();
    if (m->cfg.chain_depth && !m->temporal_staged) {
        // the next frame's Temporal step inputs do not depend on this frame's samples: queue them behind the Depth graph, off the host's critical path
        transformer_graph_step(*m->scratch, m->temporal, 1);
        m->scratch->compute_scratch();
        m->temporal_staged = true;
    }
    ggml_backend_tensor_get(m->dep_tokens, audio.data(), 0, audio.size() * 4);
}
// any other user of the Temporal stream position first takes back a step staged by depth_step
void unstage_temporal(moshi_hot_model * m) { if (m->temporal_staged) { m->temporal.offset -= 1; m->temporal_staged = false; } }
}

// ---- tensor-parallel Temporal stack ---------------------------------------------------------------------------
namespace {
void build_tp_segment(moshi_hot_model * m, int i) {
    const moshi_hot_config & c = m->cfg;
    Transformer & tp = m->temporal_tp;
    const int Lc = c.num_layers;
    Builder * b = new Builder(m->be, 32);
    Builder & g = *b;
    T x = m->tp_x;
    if (i > 0) { x = ggml_add(g, m->tp_x, m->tp_msg); g.expand(ggml_cpy(g, x, m->tp_x)); }   // the residual add of the previous half layer, on the summed partial
    if (i < 2 * Lc) {
        Layer & L = tp.layers[(size_t) (i / 2)];
        T part;
        if ((i & 1) == 0) {
            T nx = apply_norm(g, L.norm1, x);
            Rot rot;
            if (tp.max_period) rot = timestep_embedding(g, 1, tp.dim / tp.heads, tp.g_offset, tp.max_period);
            part = attention(g, tp, L, 0, tp.g_indices, nx, tp.g_bias, tp.max_period ? &rot : nullptr, false);
        } else {
            T nx = apply_norm(g, L.norm2, x);
            part = gating(g, L.gate_in[0], L.gate_out[0], nx);
        }
        g.expand(ggml_cpy(g, part, m->tp_msg));
    }
    g.alloc();
    if ((int) m->g_tp.size() <= i) m->g_tp.resize((size_t) i + 1, nullptr);
    m->g_tp[(size_t) i] = b;
}
}  // namespace

extern "C" void * moshi_hot_tp_msg(moshi_hot_model_t * m, int64_t * n) { GGML_ASSERT(m->tp_msg); if (n) *n = ggml_nelements(m->tp_msg); return m->tp_msg->data; }
extern "C" void moshi_hot_tp_begin(moshi_hot_model_t * m, const float * x) {
    GGML_ASSERT(m->tp_x);
    ggml_backend_tensor_set(m->tp_x, x, 0, (size_t) m->cfg.dim * 4);
    transformer_graph_step(*m->scratch, m->temporal_tp, 1);   // mask row -> g_bias (scratch cpy), RoPE phase, ring slot (transformer.h:1259-1289)
    m->scratch->compute_scratch();
}
extern "C" void moshi_hot_tp_segment(moshi_hot_model_t * m, int i) {
    GGML_ASSERT(i >= 0 && i <= 2 * m->cfg.num_layers);
    if ((int) m->g_tp.size() <= i || !m->g_tp[(size_t) i]) build_tp_segment(m, i);
    m->g_tp[(size_t) i]->compute();
}
// the partial-sum message through the backend's own transfer calls (a host-side stand-in for the all-reduce: tests that drive several ranks' models in one process)
extern "C" void moshi_hot_tp_msg_read(moshi_hot_model_t * m, float * out) { GGML_ASSERT(m->tp_msg); ggml_backend_tensor_get(m->tp_msg, out, 0, ggml_nbytes(m->tp_msg)); }
extern "C" void moshi_hot_tp_msg_write(moshi_hot_model_t * m, const float * in) { GGML_ASSERT(m->tp_msg); ggml_backend_tensor_set(m->tp_msg, in, 0, ggml_nbytes(m->tp_msg)); }
extern "C" void moshi_hot_tp_end(moshi_hot_model_t * m, float * out) { ggml_backend_tensor_get(m->tp_x, out, 0, (size_t) m->cfg.dim * 4); }
// The whole tensor-parallel stack pass behind the C-ABI: 2 L + 1 segment graphs with an in-place sum of the F32[dim] partial over the ranks between them -
// ncclAllReduce called from here on the backend's own stream (the communicator of moshi_hot_depth_shard_rccl_init: one per model, both sharded modes use it),
// or the caller's function (host memory on the CPU device: gloo in tests/test_temporal_tp_cpu.py). No interpreter between the segments.
extern "C" void moshi_hot_tp_set_transport(moshi_hot_model_t * m, moshi_hot_allreduce_t fn, void * user) { m->tp_allreduce = fn; m->tp_allreduce_user = user; }
extern "C" int64_t moshi_hot_tp_reductions(moshi_hot_model_t * m) { return m->tp_reductions; }
namespace {
// the 2 L + 1 segment graphs with the in-place sum of the F32[dim] partial over the ranks between them
void tp_run_segments(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    const int last = 2 * c.num_layers;
    for (int i = 0; i <= last; i++) {
        moshi_hot_tp_segment(m, i);
        if (i == last || (c.tp_world <= 1 && !m->rccl_comm && !m->tp_allreduce)) continue;
        m->tp_reductions++;
        const int64_t n = ggml_nelements(m->tp_msg);
        if (m->tp_allreduce) { m->tp_allreduce(m->tp_allreduce_user, (float *) m->tp_msg->data, n); continue; }
        GGML_ASSERT(m->rccl_comm && "moshi_hot tensor-parallel stack: no transport (moshi_hot_depth_shard_rccl_init or moshi_hot_tp_set_transport first)");
        const int rc = m->rccl_all_reduce(m->tp_msg->data, m->tp_msg->data, (size_t) n, /* ncclFloat32 */ 7, /* ncclSum */ 0, m->rccl_comm, ggml_backend_mi355x_get_stream(m->be));
        GGML_ASSERT(rc == 0 && "ncclAllReduce failed");
    }
}
// ---- frame mode (SURVEY.md section 8f.2 as a whole LM step; lm.h:555-607, 659-677 around transformer.h:910-971) ----------------------------------------
// rank 0: embedding sum -> tp_in ; broadcast(tp_in) ; every rank: tp_in -> tp_x, the stream position advances, segments + all-reduces ; rank 0: out_norm ->
// transformer_out state, text_linear, sample. The Depth graph (rank 0) follows as in any other LM step.
void tp_broadcast_in(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    if (c.tp_world <= 1 && !m->rccl_comm && !m->shard_bcast) return;
    const size_t bytes = ggml_nbytes(m->tp_in);
    if (m->shard_bcast) { m->shard_bcast(m->shard_bcast_user, m->tp_in->data, (int64_t) bytes, 0); return; }
    GGML_ASSERT(m->rccl_comm && "moshi_hot tensor-parallel frame: no transport");
    const int rc = m->rccl_broadcast(m->tp_in->data, m->tp_in->data, bytes, /* ncclUint8 */ 1, 0, m->rccl_comm, ggml_backend_mi355x_get_stream(m->be));
    GGML_ASSERT(rc == 0 && "ncclBroadcast failed");
}
void tp_stack_from_in(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    if (!m->g_tp_import) {
        m->g_tp_import = new Builder(m->be, 8);
        Builder & g = *m->g_tp_import;
        g.expand(ggml_cpy(g, ggml_view_1d(g, m->tp_in, c.dim, 0), m->tp_x));
        g.alloc();
    }
    m->g_tp_import->compute();
    transformer_graph_step(*m->scratch, m->temporal_tp, 1);
    m->scratch->compute_scratch();
    tp_run_segments(m);
    m->tp_frames++;
}
void tp_build_frame_graphs(moshi_hot_model * m) {   // (before the step's token uploads: the embedding sum's index / scale inputs are made here)
    const moshi_hot_config & c = m->cfg;
    if (!m->g_tp_pre) {
        m->g_tp_pre = new Builder(m->be, 256);
        Builder & g = *m->g_tp_pre;
        T input = build_input_embedding(m, g);
        g.expand(ggml_cpy(g, input, ggml_view_1d(g, m->tp_in, c.dim, 0)));
        g.alloc();
        m->g_tp_post = new Builder(m->be, 64);
        Builder & h = *m->g_tp_post;
        T x = apply_norm(h, m->out_norm, m->tp_x);
        m->g_transformer_out = x;
        m->text_logits = linear(h, m->text_linear, x);
        h.expand(ggml_cpy(h, x, m->transformer_out));
        m->sampler_out = sample_token(h, m->text_logits, c.temp_text, c.top_k_text);
        h.expand(m->sampler_out);
        if (m->tok_state) h.expand(ggml_cpy(h, ggml_reshape_1d(h, m->sampler_out, 1), ggml_view_1d(h, m->tok_state, 1, 0)));
        h.alloc();
    }
}
void tp_temporal_frame(moshi_hot_model * m) {   // rank 0, inside moshi_hot_lm_step_n: the token inputs of the embedding sum are already uploaded
    const moshi_hot_config & c = m->cfg;
    // The "more frames" flag rides behind the stack input. A small tensor_set is only QUEUED on the MI355X backend and reaches the stream with the next
    // compute / get / synchronize, so it is set BEFORE the embedding-sum graph: that compute's upload flush puts it on the stream ahead of the broadcast
    // (the graph itself writes elements [0, dim) only).
    const float more = 1.f;
    ggml_backend_tensor_set(m->tp_in, &more, (size_t) c.dim * 4, 4);
    m->g_tp_pre->compute();
    tp_broadcast_in(m);
    tp_stack_from_in(m);
    m->g_tp_post->compute();
}
}  // namespace
extern "C" void moshi_hot_tp_stack(moshi_hot_model_t * m, const float * x, float * out) {
    moshi_hot_tp_begin(m, x);
    tp_run_segments(m);
    moshi_hot_tp_end(m, out);
}
extern "C" void moshi_hot_tp_install(moshi_hot_model_t * m) {
    GGML_ASSERT(m->tp_x && m->tp_in && !m->cfg.chain_depth && "moshi_hot_tp_install: a model created with tp_world >= 1 and chain_depth = 0");
    // build_input_embedding appends to the model's embedding index / scale input lists: a Temporal graph built earlier would make them double
    GGML_ASSERT(!m->g_temporal && "moshi_hot_tp_install: call before the first LM step (the ordinary Temporal graph already owns the embedding inputs)");
    m->tp_frame = true;
}
extern "C" void moshi_hot_tp_stop(moshi_hot_model_t * m) {
    const float more = 0.f;
    ggml_backend_tensor_set(m->tp_in, &more, (size_t) m->cfg.dim * 4, 4);
    ggml_backend_synchronize(m->be);   // flushes the queued upload: the broadcast below must carry THIS flag, not the previous frame's
    tp_broadcast_in(m);
    ggml_backend_synchronize(m->be);
}
extern "C" int64_t moshi_hot_tp_serve(moshi_hot_model_t * m) {
    GGML_ASSERT(m->tp_x && m->tp_in);
    int64_t frames = 0;
    for (;;) {
        tp_broadcast_in(m);
        float more = 0.f;
        ggml_backend_tensor_get(m->tp_in, &more, (size_t) m->cfg.dim * 4, 4);
        if (more == 0.f) return frames;
        tp_stack_from_in(m);
        frames++;
    }
}
extern "C" int64_t moshi_hot_tp_frames(moshi_hot_model_t * m) { return m->tp_frames; }

// ---- Depth codebook shard ---------------------------------------------------------------------------------------
namespace {
// ring rows of slot `slot` of one cache tensor [D, C, H] <-> a dense [D, 1, H] F32 window of the step message
T ring_rows(Builder & g, T cache, int slot) { return ggml_view_3d(g, cache, cache->ne[0], 1, cache->ne[2], cache->nb[1], cache->nb[2], (size_t) slot * cache->nb[1]); }
T msg_rows(Builder & g, T msg, T like, int64_t index) {
    // A BF16 window into the (F32-typed) message: the ring rows travel as they are stored - 2 x layers x dim BF16 values = 24 KB per hop for the 1024-wide
    // Depth transformer, half of what the F32 form moved, and bit for bit what the importing rank's ring then holds.
    const int64_t n = like->ne[0] * like->ne[2];
    GGML_ASSERT(like->type == GGML_TYPE_BF16 && like->ne[0] % 2 == 0);
    T v = ggml_view_3d(g, msg, like->ne[0] / 2, 1, like->ne[2], (size_t) like->ne[0] * 2, (size_t) like->ne[0] * 2, (size_t) (index * n) * 2);
    v->type = GGML_TYPE_BF16; v->ne[0] = like->ne[0]; v->nb[0] = 2;
    return v;
}
// one Depth step as its own cached graph: the body of the chained loop (lm.h:505-527) for step k, the previous token read from the token
// vector, plus the packing of this step's message
void build_shard_step(moshi_hot_model * m, int k) {
    const moshi_hot_config & c = m->cfg;
    GGML_ASSERT(owns_step(c, k) && !c.demux_second_stream && !c.depformer_low_rank && c.dep_schedule_len == 0);
    Builder * b = new Builder(m->be, 64);
    Builder & g = *b;
    T last;
    if (k == 0) {
        T idx = g.tensor(GGML_TYPE_I32, 1), sc = g.tensor(GGML_TYPE_F32, 1);
        m->shard_text_idx.push_back(idx); m->shard_text_scale.push_back(sc);
        last = ggml_mul(g, ggml_get_rows(g, m->depformer_text_emb, idx), sc);
    } else last = ggml_get_rows(g, m->depformer_emb[(size_t) (k - 1)], ggml_view_1d(g, m->shard_tokens, 1, (size_t) (k - 1) * 4));
    T din = linear(g, m->depformer_in[(size_t) k], m->transformer_out);
    last = ggml_cast(g, last, GGML_TYPE_F32);
    din = ggml_add(g, din, last);
    m->depth.offset = k;
    T dout = transformer_inline(g, m->depth, din);
    T logits = linear(g, m->linears[(size_t) k], dout);
    T next = sample_token(g, logits, c.temp, c.top_k);
    g.expand(ggml_cpy(g, next, ggml_view_1d(g, m->shard_tokens, 1, (size_t) k * 4)));
    // the message: this step's new K / V rows of every layer, then the token
    const int slot = k % m->depth.capacity;
    int64_t w = 0;
    for (auto & L : m->depth.layers) for (T cache : { L.kcache, L.vcache }) {
        T rows = ring_rows(g, cache, slot);
        g.expand(ggml_cpy(g, rows, msg_rows(g, m->shard_msg, rows, w++)));
    }
    g.expand(ggml_cpy(g, next, ggml_view_1d(g, m->shard_msg, 1, (size_t) (c.dep_layers * c.dep_dim) * 4)));
    g.alloc();
    if ((int) m->g_shard_step.size() <= k) m->g_shard_step.resize((size_t) k + 1, nullptr);
    m->g_shard_step[(size_t) k] = b;
}
void build_shard_import(moshi_hot_model * m, int k) {
    const moshi_hot_config & c = m->cfg;
    Builder * b = new Builder(m->be, 16);
    Builder & g = *b;
    const int slot = k % m->depth.capacity;
    int64_t w = 0;
    for (auto & L : m->depth.layers) for (T cache : { L.kcache, L.vcache }) {
        T rows = ring_rows(g, cache, slot);
        g.expand(ggml_cpy(g, msg_rows(g, m->shard_msg, rows, w++), rows));   // BF16 -> BF16
    }
    g.expand(ggml_cpy(g, ggml_view_1d(g, m->shard_msg, 1, (size_t) (c.dep_layers * c.dep_dim) * 4), ggml_view_1d(g, m->shard_tokens, 1, (size_t) k * 4)));
    g.alloc();
    if ((int) m->g_shard_import.size() <= k) m->g_shard_import.resize((size_t) k + 1, nullptr);
    m->g_shard_import[(size_t) k] = b;
}
}  // namespace

extern "C" void * moshi_hot_depth_shard_msg(moshi_hot_model_t * m, int64_t * n) { GGML_ASSERT(m->shard_msg); if (n) *n = ggml_nelements(m->shard_msg); return m->shard_msg->data; }
extern "C" void * moshi_hot_depth_shard_tout(moshi_hot_model_t * m, int64_t * n) { GGML_ASSERT(m->shard_tout); if (n) *n = ggml_nelements(m->shard_tout); return m->shard_tout->data; }
extern "C" void moshi_hot_depth_shard_begin_export(moshi_hot_model_t * m, int32_t text_token, int more) {
    const moshi_hot_config & c = m->cfg;
    if (!m->g_shard_begin) {
        m->g_shard_begin = new Builder(m->be, 4);
        Builder & g = *m->g_shard_begin;
        g.expand(ggml_cpy(g, m->transformer_out, ggml_view_1d(g, m->shard_tout, c.dim, 0)));
        g.alloc();
    }
    const float flag = more ? 1.f : 0.f;
    ggml_backend_tensor_set(m->shard_tout, &flag, (size_t) c.dim * 4, 4);
    m->g_shard_begin->compute();
    if (owns_step(c, 0)) {
        if (m->g_shard_step.empty() || !m->g_shard_step[0]) build_shard_step(m, 0);
        int32_t id = text_token; const float sc = id == -1 ? 0.f : 1.f;   // moshi_lmmodel_depformer_step's text embedding (lm.h:532-552)
        if (id < 0) id = 0;
        ggml_backend_tensor_set(m->shard_text_idx[0], &id, 0, 4);
        ggml_backend_tensor_set(m->shard_text_scale[0], &sc, 0, 4);
    }
}
extern "C" int moshi_hot_depth_shard_begin_import(moshi_hot_model_t * m) {
    const moshi_hot_config & c = m->cfg;
    if (!m->g_shard_begin) {
        m->g_shard_begin = new Builder(m->be, 4);
        Builder & g = *m->g_shard_begin;
        g.expand(ggml_cpy(g, ggml_view_1d(g, m->shard_tout, c.dim, 0), m->transformer_out));
        g.alloc();
    }
    float flag = 0.f;
    ggml_backend_tensor_get(m->shard_tout, &flag, (size_t) c.dim * 4, 4);
    if (flag != 0.f) m->g_shard_begin->compute();
    return flag != 0.f;
}
extern "C" void moshi_hot_depth_shard_step(moshi_hot_model_t * m, int k) {
    PhaseTimer pt(m, 2);
    if ((int) m->g_shard_step.size() <= k || !m->g_shard_step[(size_t) k]) build_shard_step(m, k);
    m->g_shard_step[(size_t) k]->compute();
}
extern "C" void moshi_hot_depth_shard_import(moshi_hot_model_t * m, int k) {
    if ((int) m->g_shard_import.size() <= k || !m->g_shard_import[(size_t) k]) build_shard_import(m, k);
    m->g_shard_import[(size_t) k]->compute();
}
extern "C" void moshi_hot_depth_shard_tokens(moshi_hot_model_t * m, int32_t * out, int n) { ggml_backend_tensor_get(m->shard_tokens, out, 0, (size_t) n * 4); }

// ---- the sharded frame behind the C-ABI (the loop of lm.h:505-527 spread over ranks) ------------------------------------------
namespace {
struct nccl_id { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
void * rccl_open() {
    static void * lib = nullptr;
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return lib;
}
void shard_broadcast(moshi_hot_model * m, T msg, int root) {
    const moshi_hot_config & c = m->cfg;
    if (c.dep_shard_world <= 1 && !m->rccl_comm && !m->shard_bcast) return;   // (one rank and no transport set up: nothing to move)
    m->shard_hops++;
    const size_t bytes = ggml_nbytes(msg);
    if (m->shard_bcast) { m->shard_bcast(m->shard_bcast_user, msg->data, (int64_t) bytes, root); return; }
    GGML_ASSERT(m->rccl_comm && "moshi_hot depth shard: no transport (moshi_hot_depth_shard_rccl_init or _set_transport first)");
    // in place on the message tensor's own device storage, on the backend's stream: ordered behind the kernels that packed it and in front of the ones that unpack it
    const int rc = m->rccl_broadcast(msg->data, msg->data, bytes, /* ncclUint8 */ 1, root, m->rccl_comm, ggml_backend_mi355x_get_stream(m->be));
    GGML_ASSERT(rc == 0 && "ncclBroadcast failed");
}
void shard_chain(moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    for (int k = 0; k < c.dep_q; k++) {
        const int owner = k % c.dep_shard_world;
        if (owner == c.dep_shard_rank) moshi_hot_depth_shard_step(m, k);
        shard_broadcast(m, m->shard_msg, owner);
        if (owner != c.dep_shard_rank) moshi_hot_depth_shard_import(m, k);
    }
}
void shard_frame_hook(void * user, int32_t text_token, int32_t * audio) {
    moshi_hot_model * m = (moshi_hot_model *) user;
    moshi_hot_depth_shard_begin_export(m, text_token, 1);
    shard_broadcast(m, m->shard_tout, 0);
    shard_chain(m);
    moshi_hot_depth_shard_tokens(m, audio, m->cfg.dep_q);
}
}  // namespace
extern "C" void moshi_hot_depth_shard_set_transport(moshi_hot_model_t * m, moshi_hot_bcast_t fn, void * user) { m->shard_bcast = fn; m->shard_bcast_user = user; }
extern "C" int moshi_hot_depth_shard_rccl_unique_id(char * id128) {
    void * lib = rccl_open();
    if (!lib) return -1;
    auto get_id = (int (*)(nccl_id *)) dlsym(lib, "ncclGetUniqueId");
    if (!get_id) return -2;
    nccl_id id;
    const int rc = get_id(&id);
    if (rc == 0) memcpy(id128, id.internal, 128);
    return rc;
}
extern "C" int moshi_hot_depth_shard_rccl_init(moshi_hot_model_t * m, int rank, int world, const char * id128) {
    void * lib = rccl_open();
    if (!lib) return -1;
    auto init = (int (*)(void **, int, nccl_id, int)) dlsym(lib, "ncclCommInitRank");
    m->rccl_broadcast = (int (*)(const void *, void *, size_t, int, int, void *, void *)) dlsym(lib, "ncclBroadcast");
    m->rccl_comm_destroy = (int (*)(void *)) dlsym(lib, "ncclCommDestroy");
    m->rccl_all_reduce = (int (*)(const void *, void *, size_t, int, int, void *, void *)) dlsym(lib, "ncclAllReduce");
    if (!init || !m->rccl_broadcast || !m->rccl_comm_destroy || !m->rccl_all_reduce) return -2;
    if (ggml_backend_dev_type(ggml_backend_get_device(m->be)) != GGML_BACKEND_DEVICE_TYPE_GPU) return -3;   // (RCCL needs the MI355X backend's stream; the CPU device takes _set_transport)
    nccl_id id;
    memcpy(id.internal, id128, 128);
    m->rccl_lib = lib;
    // ncclCommInitRank binds the communicator to the calling thread's CURRENT HIP device: that must be this model's backend's, whatever the thread (or a
    // torch in the same process) touched last. A second call replaces the communicator instead of leaking it.
    ggml_backend_mi355x_make_current(m->be);
    if (m->rccl_comm) { ggml_backend_synchronize(m->be); m->rccl_comm_destroy(m->rccl_comm); m->rccl_comm = nullptr; }
    return init(&m->rccl_comm, world, id, rank);
}
extern "C" void moshi_hot_depth_shard_rccl_free(moshi_hot_model_t * m) {
    if (m->rccl_comm && m->rccl_comm_destroy) { ggml_backend_mi355x_make_current(m->be); ggml_backend_synchronize(m->be); m->rccl_comm_destroy(m->rccl_comm); }
    m->rccl_comm = nullptr;
}
extern "C" void moshi_hot_depth_shard_broadcast(moshi_hot_model_t * m, int which, int root) { shard_broadcast(m, which ? m->shard_tout : m->shard_msg, root); }
extern "C" int64_t moshi_hot_depth_shard_hops(moshi_hot_model_t * m) { return m->shard_hops; }
extern "C" void moshi_hot_depth_shard_install(moshi_hot_model_t * m) { moshi_hot_set_depth_hook(m, shard_frame_hook, m); }
extern "C" void moshi_hot_depth_shard_stop(moshi_hot_model_t * m) {
    moshi_hot_depth_shard_begin_export(m, 0, 0);
    shard_broadcast(m, m->shard_tout, 0);
}
extern "C" int64_t moshi_hot_depth_shard_serve(moshi_hot_model_t * m) {
    int64_t frames = 0;
    for (;;) {
        shard_broadcast(m, m->shard_tout, 0);
        if (!moshi_hot_depth_shard_begin_import(m)) return frames;
        shard_chain(m);
        frames++;
    }
}
extern "C" int moshi_hot_host_ring(moshi_hot_model_t * m, int32_t * dst, int max_values) {
    // the host-side delay ring of moshi_lmgen (lm.h:819-824, 935-943), row-major [rows][n_q + 1]; returns the number of values (0 when dst is too small)
    const int rows = (int) m->cache.size(), cols = rows ? (int) m->cache[0].size() : 0;
    if (!dst || rows * cols > max_values) return dst ? 0 : rows * cols;
    for (int r = 0; r < rows; r++) for (int q = 0; q < cols; q++) dst[r * cols + q] = m->cache[(size_t) r][(size_t) q];
    return rows * cols;
}
extern "C" void moshi_hot_set_depth_hook(moshi_hot_model_t * m, moshi_hot_depth_hook_t fn, void * user) { m->depth_hook = fn; m->depth_hook_user = user; }

namespace {
// the half of moshi_lmgen_step that follows the sampling (lm.h:930-979): ring write, stream position, delayed read-out
int lm_finish(moshi_hot_model * m, int32_t text_token, std::vector<int32_t> audio, bool provided, bool replace, int32_t * text_token_out, int32_t * out_audio, float * vad, bool newer_step_queued = false) {
    const moshi_hot_config & c = m->cfg;
    const int CT = (int) m->cache.size();
    const int dep_q = c.personaplex ? 8 : c.dep_q, dep_q_1 = dep_q + 1;
    m->last_text = text_token; m->last_audio = audio;
    m->offset++;
    if (!provided) {                      // lm.h:935-943
        const int wpos = m->offset % CT;
        m->cache[(size_t) wpos][0] = text_token;
        for (int q = 0; q < c.dep_q; q++) {
            // Run-ahead: the NEXT step has already been queued and has put the other speaker's delay-0 codes into this very row (lm.h:819-824); in the
            // serial order this write comes first and those codes land on top of it. Leave them: the ring then holds what the serial loop's holds.
            if (newer_step_queued && q + 1 >= dep_q_1 && c.delays[q + 1] == 0) continue;
            m->cache[(size_t) wpos][(size_t) (q + 1)] = audio[(size_t) q];
        }
    }
    if (m->offset <= m->max_delay || replace) return 0;       // lm.h:950
    int idx = (m->offset - m->max_delay + c.delays[0]) % CT;   // lm.h:954-959
    *text_token_out = m->cache[(size_t) idx][0];
    for (int i = 1; i < dep_q_1; i++) {
        idx = (m->offset - m->max_delay + c.delays[i]) % CT;
        audio[(size_t) (i - 1)] = m->cache[(size_t) idx][(size_t) i];
    }
    for (int i = 0; i < dep_q; i++) out_audio[i] = audio[(size_t) i];
    for (int32_t x : audio) if (x == -1) return 0;             // all lm->dep_q entries, the tail holding this frame's raw samples (lm.h:961-964)
    if (vad) {                                                 // lm.h:966-976
        if (m->extra_heads.size() > 2) {
            Builder & s = *m->scratch;
            T sm = ggml_soft_max(s, linear(s, m->extra_heads[2], m->transformer_out));
            s.expand_read(ggml_view_1d(s, sm, 1, 0), vad);
            s.compute_scratch();
        } else *vad = 0.f;
    }
    return 1;
}
}  // namespace

namespace { void lm_finish_entry(moshi_hot_model * m, moshi_hot_model::InFlight & f); }
extern "C" int moshi_hot_lm_step_n(moshi_hot_model_t * m, const int32_t * tokens, int n_tokens, int32_t * text_token_out, int32_t * out_audio, float * vad) {
    for (auto & o : m->inflight) lm_finish_entry(m, o);   // a blocking step behind run-ahead ones: those finish first (their results stay parked for lm_complete)
    const moshi_hot_config & c = m->cfg;
    const int ncb = c.n_q + 1, CT = (int) m->cache.size();
    const int dep_q = c.personaplex ? 8 : c.dep_q, dep_q_1 = dep_q + 1;   // lm.h:802-805
    const int needed = ncb - dep_q - 1;
    bool provided = false;
    if (needed > 0) {
        if (n_tokens == ncb) {           // every codebook given: prompt frames (lm.h:812-817)
            for (int i = 0; i < ncb; i++) m->cache[(size_t) ((m->offset + c.delays[i]) % CT)][(size_t) i] = tokens[i];
            provided = true;
        } else {                         // other speaker's codes enter the delay ring (lm.h:819-824)
            GGML_ASSERT(n_tokens >= needed);
            for (int i = 0; i < needed; i++) m->cache[(size_t) ((m->offset + c.delays[dep_q_1 + i]) % CT)][(size_t) (dep_q_1 + i)] = tokens[i];
        }
    }
    const int pos = m->offset % CT;
    std::vector<int> input((size_t) ncb);
    for (int i = 0; i < ncb; i++) input[(size_t) i] = m->offset <= c.delays[i] ? m->initial[(size_t) i] : m->cache[(size_t) pos][(size_t) i];

    if (m->tp_frame) tp_build_frame_graphs(m);
    else if (!m->g_temporal) build_temporal_graph(m);
    int32_t text_token = 0;
    // chain_depth: the Depth graph reads the sampled text token on the device; it is queued right behind the Temporal graph and both results are read afterwards
    const bool chain = c.chain_depth && c.dep_q > 0 && !m->depth_hook;
    GGML_ASSERT(!(c.chain_depth && (m->text_hook || m->depth_hook)) && "chain_depth: the text token never visits the host between the two graphs");
    {
    PhaseTimer pt(m, 1);
    // moshi_lmmodel_text_token_embed_step (lm.h:586-607): -1 -> scale 0, negative ids -> row 0
    for (int i = 0; i < ncb; i++) {
        int32_t id = input[(size_t) i];
        if (i == 0 && c.demux_second_stream) { demux_set(c, id, m->emb_idx[0], m->emb_right_idx, m->emb_right_scale); continue; }
        const float sc = id == -1 ? 0.f : 1.f;
        if (id < 0) id = 0;
        ggml_backend_tensor_set(m->emb_idx[(size_t) i], &id, 0, 4);
        ggml_backend_tensor_set(m->emb_scale[(size_t) i], &sc, 0, 4);
    }
    if (m->tp_frame) tp_temporal_frame(m);
    else {
    if (m->temporal_staged) m->temporal_staged = false;
    else {
        transformer_graph_step(*m->scratch, m->temporal, 1);
        m->scratch->compute_scratch();
    }
    m->g_temporal->compute();
    }
    if (m->after_temporal_launch) { m->after_temporal_launch(); m->after_temporal_launch = nullptr; }
    if (!chain) ggml_backend_tensor_get(m->sampler_out, &text_token, 0, 4);
    }
    if (m->text_hook) text_token = m->text_hook(m->text_hook_user, m->offset, text_token);   // on_text_hook (lm.h:880-900)

    std::vector<int32_t> audio((size_t) c.dep_q, 0);   // int_audio_tokens.resize(lm->dep_q) (lm.h:902)
    const bool replace = m->offset < c.delay_steps;    // depformer_replace_tokens (src/moshi.cpp:905)
    if (c.dep_q > 0) {
        if (!replace) {
            if (m->depth_hook) m->depth_hook(m->depth_hook_user, text_token, audio.data()); else depth_step(m, text_token, audio);
            if (chain) ggml_backend_tensor_get(m->sampler_out, &text_token, 0, 4);
        }
        else for (auto & a : audio) a = -1;            // lm.h:910-913
        if (c.delay_steps)                             // on_audio_hook (lm.h:915-921)
            for (int q = 0; q < c.dep_q; q++) if (m->offset < c.delays[q + 1] + c.delay_steps) audio[(size_t) q] = -1;
    }
    m->tok_state_for = provided ? -1 : m->offset + 1;
    return lm_finish(m, text_token, audio, provided, replace, text_token_out, out_audio, vad);
}

// ---- run-ahead (chain_depth = 2) ---------------------------------------------------------------------------------------------------------------
// A step's model-side inputs (the text and Depth tokens of the previous step) reach the Temporal graph through tok_state in device memory, so step q can
// be QUEUED - other speaker's codes uploaded, Temporal graph, Depth graph, the next step's mask row / RoPE phase / ring slot, a stream-ordered read of the
// token state, an event - while step q - 1 is still running; lm_complete then waits for the oldest event and does the host half (delay ring, read-out)
// exactly as moshi_lmgen_step orders it. The stream never waits for the host. Steps that cannot run ahead (the first max-delay frames whose inputs are
// the initial tokens, anything after provided / forced tokens, timing mode, hooks, PersonaPlex whose user columns are rewritten from the samples) are
// stepped synchronously through moshi_hot_lm_step_n and parked in the same queue, so callers see one protocol.
namespace {
bool run_ahead_config(const moshi_hot_model * m) {
    const moshi_hot_config & c = m->cfg;
    return c.chain_depth == 2 && m->tok_state && c.dep_q > 0 && !c.demux_second_stream && !c.delay_steps && !c.extra_heads &&
           !m->text_hook && !m->depth_hook && c.dep_shard_world <= 1;
}
ggml_backend_event_t take_event(moshi_hot_model * m) {
    if (!m->ev_pool.empty()) { ggml_backend_event_t e = m->ev_pool.back(); m->ev_pool.pop_back(); return e; }
    return ggml_backend_event_new(ggml_backend_get_device(m->be));
}
void lm_finish_entry(moshi_hot_model * m, moshi_hot_model::InFlight & f) {
    if (f.done || !f.ev) return;   // (no event: the entry of a blocking step that is being taken right now)
    ggml_backend_event_synchronize(f.ev);
    m->ev_pool.push_back(f.ev); f.ev = nullptr;
    std::vector<int32_t> audio(f.raw.begin() + 1, f.raw.end());
    bool newer = false, seen = false;
    for (auto & o : m->inflight) { if (&o == &f) seen = true; else if (seen && o.steady) newer = true; }
    f.ok = lm_finish(m, f.raw[0], audio, false, false, &f.out_text, f.out_audio.data(), nullptr, newer);
    f.done = true;
}
void lm_queue(moshi_hot_model * m, const int32_t * user_codes) {
    const moshi_hot_config & c = m->cfg;
    // (PersonaPlex: the Depth chain samples all 16 codebooks, the protocol takes the other speaker's 8 and hands back the model's 8, lm.h:802-805)
    const int ncb = c.n_q + 1, CT = (int) m->cache.size(), dep_q_1 = (c.personaplex ? 8 : c.dep_q) + 1, needed = ncb - dep_q_1;
    // column i of the input row comes from the device-side token state when the Depth chain samples it and the host would read that sample back: the
    // model's own columns always; one of the other speaker's only where the sample of step q - 1 lands on the ring row AFTER the code received at step
    // q - 1 (delay >= 1: the code went to that row first). With delay 0 the code received at step q overwrites the sample: host value.
    auto from_device = [&](int i) { return i < dep_q_1 || (i <= c.dep_q && c.delays[i] > 0); };
    int q = m->offset;                                   // this step's stream position: completed steps + those still running
    for (auto & o : m->inflight) if (!o.done) q++;
    bool steady = run_ahead_config(m) && !m->timing && m->g_temporal && m->g_depth && m->tok_state_for == q;
    for (int i = 0; steady && i < ncb; i++) if (from_device(i) && q <= c.delays[i]) steady = false;   // an initial token is still due on such a column
    m->inflight.emplace_back();
    moshi_hot_model::InFlight & f = m->inflight.back();
    f.out_audio.assign((size_t) c.dep_q + 8, 0);
    if (!steady) {
        for (auto & o : m->inflight) if (&o != &f) lm_finish_entry(m, o);   // older steps complete first: the blocking step needs the stream position they leave
        f.ok = moshi_hot_lm_step_n(m, user_codes, needed, &f.out_text, f.out_audio.data(), nullptr);
        f.done = true;
        return;
    }
    // other speaker's codes enter the delay ring at step q (lm.h:819-824); their columns of the input row go up from the host as ever
    f.steady = true;
    for (int i = 0; i < needed; i++) m->cache[(size_t) ((q + c.delays[dep_q_1 + i]) % CT)][(size_t) (dep_q_1 + i)] = user_codes[i];
    const int pos = q % CT;
    for (int i = 0; i < ncb; i++) {
        int32_t id = q <= c.delays[i] ? m->initial[(size_t) i] : m->cache[(size_t) pos][(size_t) i];
        float sc = id == -1 ? 0.f : 1.f;
        if (id < 0) id = 0;
        if (from_device(i)) sc = 1.f;                                   // sampled ids are never -1
        else ggml_backend_tensor_set(m->emb_idx[(size_t) i], &id, 0, 4);
        ggml_backend_tensor_set(m->emb_scale[(size_t) i], &sc, 0, 4);
    }
    if (m->temporal_staged) m->temporal_staged = false;
    else { transformer_graph_step(*m->scratch, m->temporal, 1); m->scratch->compute_scratch(); }
    m->g_temporal->compute();
    if (m->after_temporal_launch) { m->after_temporal_launch(); m->after_temporal_launch = nullptr; }
    { const float one = 1.f; ggml_backend_tensor_set(m->dep_text_scale, &one, 0, 4); }
    m->g_depth->compute();                      // text index: tok_state[0] on the device
    transformer_graph_step(*m->scratch, m->temporal, 1);
    m->scratch->compute_scratch();
    m->temporal_staged = true;
    f.raw.assign((size_t) c.dep_q + 1, 0);
    f.ev = take_event(m);
    ggml_backend_tensor_get_async(m->be, m->tok_state, f.raw.data(), 0, f.raw.size() * 4);
    ggml_backend_event_record(f.ev, m->be);
    m->tok_state_for = q + 1;
}
int lm_complete(moshi_hot_model * m, int32_t * text_token, int32_t * out_audio) {
    GGML_ASSERT(!m->inflight.empty());
    moshi_hot_model::InFlight & f = m->inflight.front();
    lm_finish_entry(m, f);
    const int ok = f.ok;
    if (ok) { *text_token = f.out_text; memcpy(out_audio, f.out_audio.data(), (size_t) (m->cfg.personaplex ? 8 : m->cfg.dep_q) * sizeof(int32_t)); }
    m->inflight.pop_front();
    return ok;
}
}  // namespace

extern "C" int moshi_hot_lm_step_run_ahead(moshi_hot_model_t * m, const int32_t * in_audio, int32_t * text_token_out, int32_t * out_audio) {
    if (in_audio) lm_queue(m, in_audio);
    const size_t keep = in_audio ? 1 : 0;
    if (m->inflight.size() <= keep) return -1;          // nothing older to hand back yet
    return lm_complete(m, text_token_out, out_audio);
}

extern "C" int moshi_hot_lm_step(moshi_hot_model_t * m, const int32_t * in_audio, int32_t * text_token_out, int32_t * out_audio) {
    const int io_dep_q = m->cfg.personaplex ? 8 : m->cfg.dep_q;
    return moshi_hot_lm_step_n(m, in_audio, m->cfg.n_q - io_dep_q, text_token_out, out_audio, nullptr);
}

extern "C" void moshi_hot_lm_step_embedding(moshi_hot_model_t * m, const float * embedding) {
    unstage_temporal(m);
    m->tok_state_for = -1;
    const moshi_hot_config & c = m->cfg;
    Builder & s = *m->scratch;
    int32_t sampled = 0;
    {
    PhaseTimer pt(m, 1);
    T input = s.constant(s.tensor(GGML_TYPE_F32, c.dim, 1, 1), embedding);
    input = ggml_cast(s, input, GGML_TYPE_F32);                       // lm.h:1022
    // the mask row as an uploaded constant instead of a view into the bias table (same values: slot cc is open iff the ring has wrapped or
    // cc <= offset, torch.h:205-223 for T = 1): consecutive voice-prompt frames then rebuild a structurally identical graph in place and the
    // backend reuses (and from the second frame on replays) one plan instead of planning ~4 000 nodes per frame
    T mask = s.tensor(GGML_TYPE_F32, m->temporal.capacity, 1);
    {
        std::vector<float> mv((size_t) m->temporal.capacity);
        for (int cc = 0; cc < m->temporal.capacity; cc++) mv[(size_t) cc] = (m->temporal.offset >= m->temporal.capacity || cc <= m->temporal.offset) ? 0.0f : -INFINITY;
        s.constant(mask, mv.data());
    }
    T x = transformer_inline(s, m->temporal, input, mask);            // moshi_lmmodel_forward_embedding (lm.h:694-709)
    x = apply_norm(s, m->out_norm, x);
    T logits = linear(s, m->text_linear, x);
    s.expand(ggml_cpy(s, x, m->transformer_out));
    s.expand_read(sample_token(s, logits, c.temp_text, c.top_k_text), &sampled);   // moshi_sample_token_int: computed, then discarded
    s.compute_scratch();
    }
    std::vector<int32_t> audio((size_t) c.dep_q, 0);
    if (c.dep_q > 0) depth_step(m, 3, audio);                         // text_token = 3 (lm.h:1035)
    m->last_text = sampled; m->last_audio = audio;
    m->offset++;
}

extern "C" void moshi_hot_set_conditions(moshi_hot_model_t * m, const float * sum, const float * cross) {
    const moshi_hot_config & c = m->cfg;
    if (sum) { GGML_ASSERT(m->cond_sum); ggml_backend_tensor_set(m->cond_sum, sum, 0, (size_t) c.dim * 4); }
    if (cross) {
        GGML_ASSERT(m->cond_cross);
        ggml_backend_tensor_set(m->cond_cross, cross, 0, (size_t) c.dim * (size_t) c.cross_len * 4);
        for (auto & L : m->temporal.layers) init_cross(*m->scratch, m->temporal, L, m->cond_cross);
    }
}
extern "C" void moshi_hot_set_text_hook(moshi_hot_model_t * m, moshi_hot_text_hook_t hook, void * user) { m->text_hook = hook; m->text_hook_user = user; }

// Batched prompt prefill: n_frames "provided" frames (every codebook given, lm.h:812-817) pushed through the Temporal stack as ONE
// [dim, T] pass per chunk instead of T single-frame steps. The reference steps prompts frame by frame (lm.h:1063-1134) and discards
// what the model samples on them; what survives a provided frame is the delay ring, the offsets and the Temporal KV rows — exactly
// what this leaves behind. The Depth graph is not run: every slot of its ring is rewritten by the next
// frame before it is read. Falls back to single provided frames where a chunk would wrap the ring (the T > 1 mask table is only
// causal before the wrap, torch.h:170-223).
extern "C" void moshi_hot_prefill(moshi_hot_model_t * m, const int32_t * tokens, int n_frames, int chunk) {
    unstage_temporal(m);
    m->tok_state_for = -1;
    const moshi_hot_config & c = m->cfg;
    const int ncb = c.n_q + 1, CT = (int) m->cache.size();
    GGML_ASSERT(ncb - (c.personaplex ? 8 : c.dep_q) - 1 > 0 && !c.demux_second_stream && !c.cross_attention);
    if (chunk < 1) chunk = 64;
    if (chunk > 64) chunk = 64;   // the device's batched kernels take up to 64 rows per pass
    ggml_backend_mi355x_set_capture(m->be, 0);   // same-shaped chunk graphs: reuse the plan, do not pay a hipGraph capture for a handful of replays
    int done = 0;
    while (done < n_frames) {
        int Tn = n_frames - done < chunk ? n_frames - done : chunk;
        if (Tn == 1 || m->temporal.offset + Tn > m->temporal.capacity) {
            int32_t text, audio[MOSHI_HOT_MAX_CODEBOOKS];
            moshi_hot_lm_step_n(m, tokens + (size_t) done * ncb, ncb, &text, audio, nullptr);
            done++;
            continue;
        }
        // host side of Tn provided frames (lm.h:812-817, 826-834, 933): ring writes, model inputs, offset
        std::vector<std::vector<int32_t>> ids((size_t) ncb, std::vector<int32_t>((size_t) Tn));
        std::vector<std::vector<float>> scales((size_t) ncb, std::vector<float>((size_t) Tn));
        for (int t = 0; t < Tn; t++) {
            const int32_t * tk = tokens + (size_t) (done + t) * ncb;
            for (int i = 0; i < ncb; i++) m->cache[(size_t) ((m->offset + c.delays[i]) % CT)][(size_t) i] = tk[i];
            const int pos = m->offset % CT;
            for (int i = 0; i < ncb; i++) {
                int32_t id = m->offset <= c.delays[i] ? m->initial[(size_t) i] : m->cache[(size_t) pos][(size_t) i];
                scales[(size_t) i][(size_t) t] = id == -1 ? 0.f : 1.f;
                ids[(size_t) i][(size_t) t] = id < 0 ? 0 : id;
            }
            m->offset++;
        }
        PhaseTimer pt(m, 1);
        Builder & s = *m->scratch;
        T input = nullptr;
        for (int i = 0; i < ncb; i++) {   // moshi_lmmodel_text_token_embed over Tn columns
            T idx = s.i32s(ids[(size_t) i]);
            T sc = s.constant(s.tensor(GGML_TYPE_F32, 1, Tn), scales[(size_t) i].data());
            T e = ggml_mul(s, ggml_get_rows(s, i == 0 ? m->text_emb : m->emb[(size_t) (i - 1)], idx), sc);
            input = input ? ggml_add(s, input, e) : e;
        }
        if (c.condition_sum) input = ggml_add(s, m->cond_sum, input);
        // the [C, Tn] block of the bias table as an uploaded constant (before the wrap it is plainly causal: slot cc is open to row t iff
        // cc <= offset + t, torch.h:170-223): the chunk graphs of one prefill are then structurally identical and the backend reuses one plan
        T mask = s.tensor(GGML_TYPE_F32, m->temporal.capacity, Tn);
        {
            std::vector<float> mv((size_t) m->temporal.capacity * (size_t) Tn);
            for (int t = 0; t < Tn; t++)
                for (int cc = 0; cc < m->temporal.capacity; cc++) mv[(size_t) t * (size_t) m->temporal.capacity + (size_t) cc] = cc <= m->temporal.offset + t ? 0.0f : -INFINITY;
            s.constant(mask, mv.data());
        }
        T x = transformer_inline(s, m->temporal, input, mask);
        // transformer_out as the last of these frames would have left it (lm.h:434, 847-849)
        T last = ggml_view_2d(s, x, x->ne[0], 1, x->nb[1], (size_t) (Tn - 1) * x->nb[1]);
        s.expand(ggml_cpy(s, apply_norm(s, m->out_norm, last), m->transformer_out));
        s.compute_scratch();
        done += Tn;
    }
    ggml_backend_mi355x_set_capture(m->be, 1);
}

static const int32_t PERSONAPLEX_PROMPT_TOKENS[17] = { 3, 948, 243, 1178, 546, 1736, 1030, 1978, 2008, 430, 1268, 381, 1611, 1095, 1495, 56, 472 };
extern "C" const int32_t * moshi_hot_personaplex_prompt_tokens(void) { return PERSONAPLEX_PROMPT_TOKENS; }

// the same prompt frames through moshi_hot_prefill: identical state afterwards, a fraction of the time
extern "C" void moshi_hot_personaplex_system_prompts_batched(moshi_hot_model_t * m, const int32_t * text_prompt, int n_text, int chunk) {
    GGML_ASSERT(m->cfg.n_q + 1 == 17);
    std::vector<int32_t> frames((size_t) (12 + n_text) * 17);
    for (int f = 0; f < 12 + n_text; f++) {
        memcpy(&frames[(size_t) f * 17], PERSONAPLEX_PROMPT_TOKENS, sizeof(PERSONAPLEX_PROMPT_TOKENS));
        if (f >= 6 && f < 6 + n_text) frames[(size_t) f * 17] = text_prompt[f - 6];
    }
    moshi_hot_prefill(m, frames.data(), 12 + n_text, chunk);
}

extern "C" void moshi_hot_personaplex_system_prompts(moshi_hot_model_t * m, const int32_t * text_prompt, int n_text) {
    GGML_ASSERT(m->cfg.n_q + 1 == 17);
    int32_t tokens[17], text, audio[MOSHI_HOT_MAX_CODEBOOKS];
    auto frame = [&](int32_t text_id) {
        memcpy(tokens, PERSONAPLEX_PROMPT_TOKENS, sizeof(tokens));
        tokens[0] = text_id;
        moshi_hot_lm_step_n(m, tokens, 17, &text, audio, nullptr);
    };
    for (int i = 0; i < 6; i++) frame(PERSONAPLEX_PROMPT_TOKENS[0]);      // moshi_lmgen_step_audio_silence, audio_silence_frame_cnt = 6
    for (int i = 0; i < n_text; i++) frame(text_prompt[i]);               // moshi_lmgen_step_text_prompt
    for (int i = 0; i < 6; i++) frame(PERSONAPLEX_PROMPT_TOKENS[0]);
}

// one iteration of the moshi-sts --bench loop (tools/moshi-sts.cpp:770-808)
extern "C" int moshi_hot_sts_frame(moshi_hot_model_t * m, const float * pcm_in, int32_t * text_token, int32_t * audio_tokens, float * pcm_out) {
    int32_t * codes = m->tokens_tmp.data();
    moshi_hot_mimi_encode(m, pcm_in, codes);
    if (!moshi_hot_lm_step(m, codes, text_token, audio_tokens)) return 0;
    moshi_hot_mimi_decode(m, audio_tokens, pcm_out);
    return 1;
}

// The same loop software-pipelined over the two command streams (codec_stream): call k runs the LM step of frame k while the codec stream decodes
// frame k - 1 and encodes frame k + 1. Each graph sees exactly the inputs and states it sees in moshi_hot_sts_frame, in the same per-graph order
// (encode 0, 1, 2 ...; decode 0, 1, 2 ...; LM 0, 1, 2 ...), so tokens and PCM are bit-identical; the hand-offs are host round trips (codes and
// tokens are a few integers). Without a second stream the calls degenerate to the serial order.
extern "C" void moshi_hot_sts_pipeline_begin(moshi_hot_model_t * m, const float * pcm0) {
    if (m->cfg.enable_mimi_encoder) moshi_hot_mimi_encode(m, pcm0, m->pipe_codes.data());     // (a model without encoder - tts - steps on no input codes)
    m->pipe_have_codes = true;
    m->pipe_have_tokens = false;
}
extern "C" int moshi_hot_sts_pipeline_frame(moshi_hot_model_t * m, const float * pcm_next, int32_t * text_token, int32_t * audio_tokens, float * pcm_prev) {
    GGML_ASSERT(m->pipe_have_codes && "moshi_hot_sts_pipeline_begin first");
    if (m->cfg.chain_depth == 2 && m->tok_state) {
        // run-ahead: the LM step of frame k is queued behind the one of frame k - 1 before that one's tokens are looked at; the tokens (and, as
        // ever, the PCM) this call hands back are frame k - 1's
        lm_queue(m, m->pipe_codes.data());                                              // frame k
        const bool have_prev = m->inflight.size() > 1;
        const int ok = have_prev ? lm_complete(m, text_token, audio_tokens) : 0;       // frame k - 1
        // both codec halves start now, i.e. beside the Temporal graph of frame k (its bandwidth-bound mat-vecs lose ~6 % to them); holding the decode
        // half back until that graph has finished, so that it runs beside the latency-bound Depth chain instead, measured 340 vs 350 frames/s
        if (ok) mimi_decode_launch(m, audio_tokens);
        if (pcm_next) mimi_encode_launch(m, pcm_next);                                  // frame k + 1
        if (ok) mimi_decode_finish(m, pcm_prev);
        if (pcm_next) mimi_encode_finish(m, m->pipe_codes.data());
        m->pipe_have_codes = pcm_next != nullptr;
        return (ok ? 3 : 0) | 4;
    }
    // (stt has no decoder, tts no encoder: the halves that exist are overlapped)
    const bool dec = m->pipe_have_tokens && m->cfg.enable_mimi_decoder;
    if (!m->cfg.enable_mimi_encoder) pcm_next = nullptr;
    // the LM stream is the critical path: its Temporal graph is queued first, the codec launches follow while it runs
    m->after_temporal_launch = [m, dec, pcm_next]() {
        if (dec) mimi_decode_launch(m, m->pipe_tokens.data());          // frame k - 1
        if (pcm_next) mimi_encode_launch(m, pcm_next);                  // frame k + 1
    };
    const int io = m->cfg.personaplex ? 8 : m->cfg.dep_q;
    const int ok = moshi_hot_lm_step_n(m, m->pipe_codes.data(), m->cfg.n_q - io, text_token, audio_tokens,
                                       m->extra_heads.size() > 2 ? &m->pipe_vad : nullptr);   // frame k (blocks on the LM stream only)
    if (dec) mimi_decode_finish(m, pcm_prev);
    if (pcm_next) mimi_encode_finish(m, m->pipe_codes.data());
    m->pipe_have_codes = pcm_next != nullptr || !m->cfg.enable_mimi_encoder;
    m->pipe_have_tokens = ok != 0;
    if (ok) memcpy(m->pipe_tokens.data(), audio_tokens, (size_t) (m->cfg.personaplex ? 8 : m->cfg.dep_q) * sizeof(int32_t));
    return (ok ? 1 : 0) | (dec ? 2 : 0);
}
extern "C" int moshi_hot_sts_pipeline_end(moshi_hot_model_t * m, int32_t * text_token, int32_t * audio_tokens, float * pcm_last) {
    if (m->cfg.chain_depth == 2 && m->tok_state) {
        int r = 0;
        while (!m->inflight.empty()) {           // at most one step is outstanding
            const int ok = lm_complete(m, text_token, audio_tokens);
            if (ok) { moshi_hot_mimi_decode(m, audio_tokens, pcm_last); r = 3; }
        }
        return r;
    }
    if (!m->pipe_have_tokens || !m->cfg.enable_mimi_decoder) return 0;
    moshi_hot_mimi_decode(m, m->pipe_tokens.data(), pcm_last);
    m->pipe_have_tokens = false;
    return 2;
}

extern "C" float moshi_hot_sts_pipeline_vad(moshi_hot_model_t * m) { return m->pipe_vad; }
extern "C" int64_t moshi_hot_offset(moshi_hot_model_t * m) { return m->offset; }
extern "C" void moshi_hot_last_raw_tokens(moshi_hot_model_t * m, int32_t * text_token, int32_t * audio_tokens) {
    *text_token = m->last_text;
    for (size_t i = 0; i < m->last_audio.size(); i++) audio_tokens[i] = m->last_audio[i];
}
extern "C" size_t moshi_hot_weight_bytes(moshi_hot_model_t * m, int part) { return part >= 0 && part < 6 ? m->W->bytes[part] : 0; }
extern "C" int moshi_hot_read_last(moshi_hot_model_t * m, const char * what, float * out, int64_t n) {
    T t = nullptr;
    if (!strcmp(what, "text_logits")) t = m->text_logits;
    else if (!strcmp(what, "transformer_out")) t = m->transformer_out;
    else if (!strcmp(what, "transformer_in")) t = m->g_transformer_in;     // sum of the 17 embeddings (input of the Temporal stack)
    else if (!strcmp(what, "stack_out")) t = m->g_stack_out;               // output of the Temporal stack, before out_norm
    else if (!strcmp(what, "enc_latent_first")) t = m->enc_latent[0];
    else if (!strcmp(what, "enc_latent_rest")) t = m->enc_latent[1];
    else if (!strncmp(what, "dep_logits", 10)) { const int k = atoi(what + 10); if (k >= 0 && k < (int) m->dep_logits.size()) t = m->dep_logits[(size_t) k]; }
    if (!t || ggml_nelements(t) < n) return -1;
    ggml_backend_tensor_get(t, out, 0, (size_t) n * 4);
    return 0;
}
extern "C" struct ggml_cgraph * moshi_hot_graph(moshi_hot_model_t * m, int which) {
    Builder * b = which == 0 ? m->g_temporal : which == 1 ? m->g_depth : which == 2 ? m->g_enc : which == 3 ? m->g_dec : nullptr;
    return b ? b->gf : nullptr;
}
extern "C" struct ggml_tensor * moshi_hot_weight(moshi_hot_model_t * m, const char * name) {
    auto it = m->W->by_name.find(name);
    return it == m->W->by_name.end() ? nullptr : it->second;
}
extern "C" void moshi_hot_set_timing(moshi_hot_model_t * m, int on) { m->timing = on != 0; for (int i = 0; i < 4; i++) { m->phase_us[i] = 0; m->phase_n[i] = 0; } }
extern "C" void moshi_hot_get_timing(moshi_hot_model_t * m, double * us_per_call) { for (int i = 0; i < 4; i++) us_per_call[i] = m->phase_n[i] ? m->phase_us[i] / (double) m->phase_n[i] : 0.0; }
extern "C" void moshi_hot_force_last(moshi_hot_model_t * m, int32_t text_token, const int32_t * audio_tokens) {
    const int wpos = m->offset % (int) m->cache.size();
    m->cache[(size_t) wpos][0] = text_token;
    for (int q = 0; q < m->cfg.dep_q; q++) m->cache[(size_t) wpos][(size_t) (q + 1)] = audio_tokens[q];
    m->tok_state_for = -1;   // the device-side token state still holds the model's own samples
}
extern "C" void moshi_hot_set_context_fill(moshi_hot_model_t * m, int64_t offset) {
    unstage_temporal(m);
    m->temporal.offset = (int) offset;
    if (m->tp_x) m->temporal_tp.offset = (int) offset;   // tensor-parallel frame mode steps its own stack (head-sliced rings): same stream position
}

// The K (kv = 0) / V (kv = 1) ring of one layer as the bytes it holds (BF16 [D, C, H]), out of / into the model: parity runs that restart every frame from
// another executor's state (tests/test_full_width_parity.py: teacher forcing per frame on the benchmark's own weights). Returns the ring's size in bytes;
// copies min(nbytes, size) when buf is not NULL. which: 0 Temporal, 1 Depth.
extern "C" int64_t moshi_hot_ring_bytes(moshi_hot_model_t * m, int which, int layer, int kv, void * buf, int64_t nbytes, int write) {
    Transformer & tr = which == 0 ? m->temporal : m->depth;
    if (layer < 0 || layer >= (int) tr.layers.size()) return -1;
    T t = kv == 0 ? tr.layers[(size_t) layer].kcache : tr.layers[(size_t) layer].vcache;
    if (!t) return -1;
    const int64_t size = (int64_t) ggml_nbytes(t);
    if (buf) {
        const size_t n = (size_t) (nbytes < size ? nbytes : size);
        if (write) ggml_backend_tensor_set(t, buf, 0, n); else ggml_backend_tensor_get(t, buf, 0, n);
    }
    return size;
}

// identical pseudo-random BF16 rows in every slot of the K / V rings (tests: long-context attention against the oracle over a known cache)
extern "C" void moshi_hot_fill_ring(moshi_hot_model_t * m, int which, int layer, uint64_t seed, float scale) {
    Transformer & tr = which == 0 ? m->temporal : m->depth;
    for (int l = 0; l < (int) tr.layers.size(); l++) {
        if (layer >= 0 && l != layer) continue;
        Layer & L = tr.layers[(size_t) l];
        for (int kv = 0; kv < 2; kv++) {
            T t = kv == 0 ? L.kcache : L.vcache;
            if (!t) continue;
            GGML_ASSERT(t->type == GGML_TYPE_BF16);
            const size_t n = (size_t) ggml_nelements(t);
            std::vector<uint16_t> bits(n);
            uint64_t s = seed * 0x9E3779B97F4A7C15ull + (uint64_t) which * 1000003ull + (uint64_t) l * 7919ull + (uint64_t) kv;
            for (size_t i = 0; i < n; i++) {
                // splitmix64 -> sum of four uniforms (variance 1/3, close enough to a bell) -> BF16 by truncation of an exactly representable value
                s += 0x9E3779B97F4A7C15ull;
                uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
                const float u = ((float) (z & 0xffff) + (float) ((z >> 16) & 0xffff) + (float) ((z >> 32) & 0xffff) + (float) (z >> 48)) / 65536.0f - 2.0f;
                const float v = u * 1.7320508f * scale;
                uint32_t w; memcpy(&w, &v, 4);
                bits[i] = (uint16_t) (w >> 16);
            }
            ggml_backend_tensor_set(t, bits.data(), 0, n * 2);
        }
    }
}

extern "C" int moshi_hot_layer_probe(moshi_hot_model_t * m, int which, int layer, int weight_set, const float * x_in, int offset, float * x_out,
                                     moshi_hot_node_visitor_t visit, void * user) {
    Transformer & tr = which == 0 ? m->temporal : m->depth;
    GGML_ASSERT(layer >= 0 && layer < (int) tr.layers.size());
    Layer & L = tr.layers[(size_t) layer];
    const bool multi = L.in_proj.size() > 1;
    GGML_ASSERT(weight_set >= 0 && weight_set < (int) L.in_proj.size());
    Builder & s = *m->scratch;
    T x = s.constant(s.tensor(GGML_TYPE_F32, tr.dim, 1, 1), x_in);
    // the same three inputs transformer_inline derives from the offset (transformer.h:1182-1215), mask row read from the bias table
    T attn_bias = bias_pattern_index(s, tr, 1, offset);
    Rot rot;
    if (tr.max_period) rot = timestep_embedding(s, 1, tr.dim / tr.heads, s.f32((float) offset), tr.max_period);
    T indices = s.i32s({ offset % tr.capacity });
    T y = transformer_layer(s, tr, L, weight_set, indices, x, attn_bias, tr.max_period ? &rot : nullptr, multi);
    s.expand(y);
    s.alloc_poisoned();
    s.compute();
    if (x_out) ggml_backend_tensor_get(y, x_out, 0, (size_t) tr.dim * 4);
    const int n = ggml_graph_n_nodes(s.gf);
    if (visit) for (int i = 0; i < n; i++) visit(user, i, ggml_graph_node(s.gf, i));
    s.release_scratch();
    return n;
}
