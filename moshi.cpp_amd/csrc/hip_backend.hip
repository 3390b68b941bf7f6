// hip_backend.hip — the MI355X ("ROCm<N>") ggml backend: device registry, pooled device buffers,
// batched small uploads, and the graph planner / executor.
//
// Execution model (SURVEY.md §7 "hard parts": ~4.5 k ggml nodes per frame vs. a sub-millisecond roofline
// budget, so per-node dispatch is impossible):
//   * ggml_backend_graph_compute() turns a cgraph into a *plan*: an ordered list of kernel launches in
//     which the hot node sequences (norm -> mat-vec, gated FFN, residual add, the whole single-token
//     attention block, the 17-stream embedding sum) are matched and replaced by the fused kernels of
//     hip_kernels_fused.hip; everything else falls back to one generic kernel per node.
//   * plans of cached graphs (Temporal / Depth / Mimi graphs are built once by the caller and replayed
//     every frame, src/context.h:538-544) are captured into a hipGraph and replayed with one
//     hipGraphLaunch; throw-away scratch graphs (src/context.h:628-653) are launched directly.
//   * the ~40 scalar tensor_set calls per frame (lm_utils.h:172-182) are queued in pinned host memory
//     and scattered by one kernel at the next compute / read-back.
// ggml semantics preserved: every non-view tensor has its own storage, execution order = node order,
// results are visible to tensor_get when graph_compute returns (stream-ordered, tensor_get synchronises).
#include "hip_common.h"
#include "ggml-cpu.h"

#include <string.h>

#include <algorithm>
#include <functional>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------------------------------
// per-device context
// ---------------------------------------------------------------------------------------------------
#define UPLOAD_SLOTS       8
#define UPLOAD_BLOB_BYTES  (64 * 1024)
#define UPLOAD_MAX_DESCS   1024
#define READBACK_PINNED_BYTES (64 * 1024)
#define AGET_SLOTS 64
#define AGET_SLOT_BYTES 256
#define UPLOAD_SMALL_MAX   16384   // the 7.7 KB PCM frame rides the batched path too (a pageable hipMemcpy + sync otherwise)

struct plan_t;

struct upload_slot {
    char * blob = nullptr;            // pinned, device-mapped
    upload_desc * descs = nullptr;    // pinned, device-mapped
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

struct hip_ctx {
    int device = 0;
    std::string name, description;
    hipStream_t stream = nullptr;
    int flags = 0;
    bool no_capture = false;     // ggml_backend_mi355x_set_capture(.., 0): plans of repeated graphs are reused but not captured into hipGraphs
    ggml_mi355x_stats stats = {};
    // pooled allocations: size class -> free pointers
    std::map<size_t, std::vector<void *>> pool;
    // batched uploads
    upload_slot slots[UPLOAD_SLOTS];
    int cur_slot = 0;
    int n_pending = 0;
    size_t pending_bytes = 0;
    // flag 8: per-launch timing of the dominant kernel
    mv_profile prof = { nullptr, 0, 0 };
    double prof_seconds = 0; int64_t prof_launches = 0, prof_bytes = 0;
    double prof_seconds_v[3] = { 0, 0, 0 }; int64_t prof_launches_v[3] = { 0, 0, 0 }, prof_bytes_v[3] = { 0, 0, 0 };   // the same, by kernel variant
    double prof_chain_seconds = 0; int64_t prof_chain_launches = 0, prof_chain_bytes = 0, prof_chain_phases = 0;   // persistent chain launches (matvec_chain_kernel), stream events
    hipEvent_t chain_ev[2] = { nullptr, nullptr };
    // cached plans keyed by cgraph pointer
    std::unordered_map<const ggml_cgraph *, plan_t *> plans;
    uint64_t orphan_clock = 0;
    // device-visible error word in pinned host memory: kernels with bounded waits raise it instead of hanging
    volatile unsigned * err_host = nullptr;
    unsigned * err_dev = nullptr;
    char * readback = nullptr;   // pinned staging for small device -> host reads
    // stream-ordered small read-backs (ggml_backend_tensor_get_async): staged in pinned slots, handed to the caller's memory by the synchronize / event
    // wait that covers them
    struct async_get { void * dst; size_t size; int slot; uint64_t seq; };
    char * aget_pinned = nullptr;
    std::deque<async_get> aget_pending;
    uint64_t aget_seq = 0;
    bool in_use = false;         // stream contexts only: handed out by ggml_backend_mi355x_init_stream, returned by ggml_backend_free
    int usable_cus = 0;          // compute units this context's stream may use: the device's count, or MI355X_STREAM_CUS of them (CU-masked stream)
    ggml_backend_device dev_obj;
};

static std::vector<hip_ctx *> & contexts() { static std::vector<hip_ctx *> v; return v; }
// additional command streams on a device (ggml_backend_mi355x_init_stream): a full context of its own - stream, upload slots, pool, plan cache -
// on the same GPU, so graphs submitted through it overlap with the device's first stream. Not listed by the registry.
static std::vector<hip_ctx *> & stream_contexts() { static std::vector<hip_ctx *> v; return v; }

static void set_device(hip_ctx * c) { HIP_CHECK(hipSetDevice(c->device)); }

static void ctx_init_lazy(hip_ctx * c) {
    if (c->stream) return;
    set_device(c);
    {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, c->device));
        c->usable_cus = prop.multiProcessorCount;
        // MI355X_STREAM_CUS=n: confine this backend's stream to n compute units (hipExtStreamCreateWithCUMask with bits 0 .. n - 1 of the mask set; which
        // physical CUs / XCDs those bits name is the runtime's mapping and is not assumed here) - the shape of a CU-masked or partitioned deployment. Kernels
        // whose workgroups wait for each other (persistent chains, the fused attention + out_proj launch, split attention) are planned against this NUMBER
        // and fall back to plain launches when their grid cannot be resident (build_plan). Unlike the default stream below, a CU-masked stream is created
        // WITHOUT hipStreamNonBlocking (the call takes no flags): it synchronises implicitly with the NULL stream - a diagnostic / test shape, not a fast path.
        const char * e = getenv("MI355X_STREAM_CUS");
        const int lim = e ? atoi(e) : 0;
        if (lim > 0 && lim < prop.multiProcessorCount) {
            std::vector<uint32_t> mask((size_t) (prop.multiProcessorCount + 31) / 32, 0u);
            for (int i = 0; i < lim; i++) mask[(size_t) i / 32] |= 1u << (i % 32);
            HIP_CHECK(hipExtStreamCreateWithCUMask(&c->stream, (uint32_t) mask.size(), mask.data()));
            c->usable_cus = lim;
        }
    }
    if (!c->stream) HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    for (auto & s : c->slots) {
        HIP_CHECK(hipHostMalloc((void **) &s.blob, UPLOAD_BLOB_BYTES, hipHostMallocMapped));
        HIP_CHECK(hipHostMalloc((void **) &s.descs, UPLOAD_MAX_DESCS * sizeof(upload_desc), hipHostMallocMapped));
        HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    HIP_CHECK(hipHostMalloc((void **) &c->readback, READBACK_PINNED_BYTES, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **) &c->aget_pinned, AGET_SLOTS * AGET_SLOT_BYTES, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void **) &c->err_host, 64, hipHostMallocMapped));
    c->err_host[0] = 0u;
    HIP_CHECK(hipHostGetDevicePointer((void **) &c->err_dev, (void *) c->err_host, 0));
}

static void check_device_error(hip_ctx * c) {
    if (c->err_host && c->err_host[0] != 0u)
        GGML_ABORT("mi355x backend: a kernel gave up a bounded wait (code %u: 1 = split attention head barrier, 2 = chain engine hand-off, 3 = stream engine hand-off); results are invalid", c->err_host[0]);
}

// ---- pool -----------------------------------------------------------------------------------------
static size_t size_class(size_t n) {
    if (n > ((size_t) 64 << 20)) return 0;   // large: not pooled
    size_t c = 256;
    while (c < n) c <<= 1;
    return c;
}
// MI355X_POISON=1 (diagnosis): every buffer and workspace handed out is first filled with 0xFF bytes (NaN as F32 / BF16 / F16, -1 as I32), so
// that a kernel reading memory nobody wrote shows up as NaNs or wild indices in the parity tests instead of passing on leftover zeros
static void * pool_poison(hip_ctx * c, void * p, size_t n) {
    static const bool poison = getenv("MI355X_POISON") != nullptr;
    if (poison && p) { HIP_CHECK(hipMemsetAsync(p, 0xFF, n, c->stream)); HIP_CHECK(hipStreamSynchronize(c->stream)); }
    return p;
}
// MI355X_GUARD=1 (diagnosis): every allocation sits between two 4 KB zones filled with 0xA5; they are checked when the allocation is
// returned - a kernel writing just outside its buffer aborts the process with the offending size instead of corrupting a neighbour
#define GUARD_BYTES 4096
static bool guard_mode() { static const bool g = getenv("MI355X_GUARD") != nullptr; return g; }
static void guard_fill(hip_ctx * c, char * user, size_t actual) {
    HIP_CHECK(hipMemsetAsync(user - GUARD_BYTES, 0xA5, GUARD_BYTES, c->stream));
    HIP_CHECK(hipMemsetAsync(user + actual, 0xA5, GUARD_BYTES, c->stream));
}
static void guard_check(hip_ctx * c, char * user, size_t actual) {
    static std::vector<unsigned char> h(2 * GUARD_BYTES);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipMemcpy(h.data(), user - GUARD_BYTES, GUARD_BYTES, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(h.data() + GUARD_BYTES, user + actual, GUARD_BYTES, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < 2 * GUARD_BYTES; i++)
        if (h[i] != 0xA5) GGML_ABORT("mi355x: guard zone %s a %zu-byte allocation was overwritten (offset %zu)", i < GUARD_BYTES ? "before" : "after", actual, i % GUARD_BYTES);
}
static void * pool_alloc(hip_ctx * c, size_t n, size_t * actual) {
    set_device(c);
    const size_t cls = size_class(n ? n : 1);
    if (cls) {
        auto & fl = c->pool[cls];
        *actual = cls;
        if (!fl.empty()) { void * p = fl.back(); fl.pop_back(); if (guard_mode()) guard_fill(c, (char *) p, *actual); return pool_poison(c, p, *actual); }
    } else *actual = n;
    void * p = nullptr;
    hipError_t e = hipMalloc(&p, *actual + (guard_mode() ? 2 * GUARD_BYTES : 0));
    if (e != hipSuccess) { fprintf(stderr, "mi355x: hipMalloc(%zu) failed: %s\n", *actual, hipGetErrorString(e)); return nullptr; }
    if (guard_mode()) { p = (char *) p + GUARD_BYTES; guard_fill(c, (char *) p, *actual); }
    return pool_poison(c, p, *actual);
}
static void pool_free(hip_ctx * c, void * p, size_t actual) {
    if (!p) return;
    if (guard_mode()) guard_check(c, (char *) p, actual);
    const size_t cls = size_class(actual);
    if (cls && cls == actual) { c->pool[cls].push_back(p); return; }
    set_device(c);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    HIP_CHECK(hipFree(guard_mode() ? (char *) p - GUARD_BYTES : (char *) p));
}

// ---- batched uploads -------------------------------------------------------------------------------
static void flush_uploads(hip_ctx * c) {
    if (c->n_pending == 0) return;
    upload_slot & s = c->slots[c->cur_slot];
    k_scatter_uploads(c->stream, s.descs, s.blob, c->n_pending);
    HIP_CHECK(hipEventRecord(s.done, c->stream));
    s.in_flight = true;
    c->stats.uploads_batched += c->n_pending;
    c->n_pending = 0;
    c->pending_bytes = 0;
    c->cur_slot = (c->cur_slot + 1) % UPLOAD_SLOTS;
    upload_slot & n = c->slots[c->cur_slot];
    if (n.in_flight) { HIP_CHECK(hipEventSynchronize(n.done)); n.in_flight = false; }
}

static void queue_upload(hip_ctx * c, void * dst, const void * src, size_t size) {
    const size_t padded = GGML_PAD(size, 16);
    if (c->n_pending >= UPLOAD_MAX_DESCS || c->pending_bytes + padded > UPLOAD_BLOB_BYTES) flush_uploads(c);
    // the scatter kernel copies all descriptors of a batch concurrently: a second write to bytes already pending (a state's zero fill
    // followed by its first value, say) must not share a batch with the first, or the older data may land last
    for (int i = 0; i < c->n_pending; i++) {
        const upload_desc & d = c->slots[c->cur_slot].descs[i];
        if ((char *) dst < d.dst + d.size && d.dst < (char *) dst + size) { flush_uploads(c); break; }
    }
    upload_slot & s = c->slots[c->cur_slot];
    memcpy(s.blob + c->pending_bytes, src, size);
    s.descs[c->n_pending] = { (char *) dst, (uint32_t) c->pending_bytes, (uint32_t) size };
    c->n_pending++;
    c->pending_bytes += padded;
}

// ---------------------------------------------------------------------------------------------------
// buffers
// ---------------------------------------------------------------------------------------------------
struct hip_buffer_ctx { hip_ctx * c; size_t actual; };

// the stream has reached (at least) read-back `upto`: copy the staged bytes to where the callers asked for them
static void deliver_async_gets(hip_ctx * c, uint64_t upto) {
    while (!c->aget_pending.empty() && c->aget_pending.front().seq <= upto) {
        const hip_ctx::async_get & g = c->aget_pending.front();
        memcpy(g.dst, c->aget_pinned + (size_t) g.slot * AGET_SLOT_BYTES, g.size);
        c->aget_pending.pop_front();
    }
}

static void evict_plans_of_buffer(hip_ctx * c, const ggml_backend_buffer * b);

static void hip_buf_free(ggml_backend_buffer_t b) {
    hip_buffer_ctx * bc = (hip_buffer_ctx *) b->context;
    flush_uploads(bc->c);
    // plans (captured hipGraphs with baked pointers, pool workspaces) must not silently outlive the memory they address - on any stream of the device
    evict_plans_of_buffer(bc->c, b);
    for (hip_ctx * o : contexts()) if (o != bc->c && o->device == bc->c->device) evict_plans_of_buffer(o, b);
    for (hip_ctx * o : stream_contexts()) if (o != bc->c && o->device == bc->c->device) evict_plans_of_buffer(o, b);
    pool_free(bc->c, b->base, bc->actual);
    delete bc;
    delete b;
}
static void hip_buf_set(ggml_backend_buffer_t b, struct ggml_tensor * t, const void * data, size_t offset, size_t size) {
    hip_ctx * c = ((hip_buffer_ctx *) b->context)->c;
    set_device(c);
    if (size <= UPLOAD_SMALL_MAX && !(c->flags & 4)) { queue_upload(c, (char *) t->data + offset, data, size); return; }
    flush_uploads(c);
    HIP_CHECK(hipMemcpyAsync((char *) t->data + offset, data, size, hipMemcpyHostToDevice, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
static void hip_buf_get(ggml_backend_buffer_t b, const struct ggml_tensor * t, void * data, size_t offset, size_t size) {
    hip_ctx * c = ((hip_buffer_ctx *) b->context)->c;
    set_device(c);
    flush_uploads(c);
    if (size <= READBACK_PINNED_BYTES) {
        // small read-backs (token ids, one PCM frame) land in a pinned staging buffer: a pageable destination makes the runtime
        // pin / stage it on every call
        HIP_CHECK(hipMemcpyAsync(c->readback, (const char *) t->data + offset, size, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
        memcpy(data, c->readback, size);
    } else {
        HIP_CHECK(hipMemcpyAsync(data, (const char *) t->data + offset, size, hipMemcpyDeviceToHost, c->stream));
        HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    deliver_async_gets(c, c->aget_seq);
    check_device_error(c);
}
static void hip_buf_memset(ggml_backend_buffer_t b, struct ggml_tensor * t, uint8_t v, size_t offset, size_t size) {
    hip_ctx * c = ((hip_buffer_ctx *) b->context)->c;
    set_device(c);
    flush_uploads(c);
    HIP_CHECK(hipMemsetAsync((char *) t->data + offset, v, size, c->stream));
}
static void hip_buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    hip_ctx * c = ((hip_buffer_ctx *) b->context)->c;
    set_device(c);
    flush_uploads(c);
    HIP_CHECK(hipMemsetAsync(b->base, v, b->size, c->stream));
}

static ggml_backend_buffer_t hip_alloc_buffer(ggml_backend_t backend, size_t size) {
    hip_ctx * c = (hip_ctx *) backend->context;
    ctx_init_lazy(c);
    size_t actual = 0;
    void * p = pool_alloc(c, size, &actual);
    if (!p) return NULL;
    auto * b = new ggml_backend_buffer;
    b->iface = { hip_buf_free, hip_buf_memset, hip_buf_set, hip_buf_get, hip_buf_clear };
    b->device = &c->dev_obj;
    b->base = p;
    b->size = size;
    b->context = new hip_buffer_ctx{ c, actual };
    b->is_host = false;
    return b;
}

// ---------------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------------
typedef std::function<void(hipStream_t)> step_fn;
// one launch of a plan. Block mat-vecs keep their arguments visible: runs of consecutive small ones are merged into persistent chain launches
// (hip_chain.hip) once the whole plan is laid out
struct pstep {
    step_fn fn; bool is_mv = false; mv_args mv;
    chain_plan * chain = nullptr;   // a persistent chain launch (timed on its own in profile mode)
    const vq_level_args * vq = nullptr;   // one RVQ encode level (k_vq_level): consecutive levels of a stack may become one launch (k_vq_chain_*)
    // a step that reads nothing a launch of this graph writes (the RoPE table of add(positions, offset)) and writes [hoist_dst, + hoist_bytes): it may move in
    // front of a neighbouring mat-vec whose operands it does not overlap, so that the mat-vec stays next to the steps it chains with
    const char * hoist_dst = nullptr; size_t hoist_bytes = 0;
    template <typename F> pstep(F f) : fn(std::move(f)) { memset(&mv, 0, sizeof(mv)); }
    pstep(const mv_args & a) : fn([a](hipStream_t s) { k_matvec(s, a); }), is_mv(true), mv(a) {}
    // a step that is not a mat-vec but may be taken into a persistent step program next to its neighbours (mv.special): it keeps its own launch otherwise
    template <typename F> static pstep special_step(F f, int special, const attn_args * at, const lowrank_embed_args * lr, const sample_args * smp = nullptr) {
        pstep st(std::move(f));
        st.is_mv = true; st.mv.special = special; st.mv.attn = at; st.mv.lr = lr; st.mv.smp = smp;
        return st;
    }
};

struct plan_t {
    std::vector<pstep> steps;
    std::vector<chain_plan *> chains;
    std::vector<std::pair<void *, size_t>> workspaces;
    std::vector<std::unique_ptr<attn_args>> attn_copies;
    std::vector<std::unique_ptr<lowrank_embed_args>> lowrank_copies;
    std::vector<std::unique_ptr<sample_args>> sampler_copies;
    std::vector<std::unique_ptr<vq_level_args>> vq_copies;
    std::vector<vq_chain_plan *> vq_chains;
    std::vector<std::unique_ptr<conv_scatter>> scatter_slots;   // see emitter::scatter_of
    std::vector<const ggml_backend_buffer *> buffers;   // every buffer a node / leaf of the planned graph lives in: freeing one orphans the plan
    uint64_t orphan_seq = 0;                            // != 0: a buffer of the planned graph has been freed since (see evict_plans_of_buffer)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t hash = 0;
    int n_nodes = 0, n_fused = 0, n_chained = 0, n_attn_folded = 0, n_step_programs = 0, n_vq_chained = 0;
};

static void plan_free(hip_ctx * c, plan_t * p) {
    if (p->exec) (void) hipGraphExecDestroy(p->exec);
    if (p->graph) (void) hipGraphDestroy(p->graph);
    for (auto & w : p->workspaces) pool_free(c, w.first, w.second);
    for (chain_plan * ch : p->chains) k_chain_free(ch);
    for (vq_chain_plan * vc : p->vq_chains) k_vq_chain_free(vc);
    delete p;
}

// A plan bakes device addresses into its launch arguments. When a buffer it touches is freed (moshi_hot_free, a scratch context going away) the
// plan becomes an ORPHAN: it is only ever used again if a graph with the same full hash - same tensor descriptors at the same host addresses with
// the same device addresses - is submitted under the same cgraph pointer (the scratch protocol of src/context.h:628-653 rebuilds structurally
// identical graphs in place: one plan then serves every chunk of a prompt prefill). Anything else at that address replaces it, and at most
// MAX_ORPHANS are kept (oldest first out), so plans of freed models do not pile up with their pool workspaces.
#define MAX_ORPHANS 8
static void collect_plan_buffers(plan_t * p, const ggml_cgraph * g) {
    p->buffers.clear();
    auto note = [&](const ggml_tensor * t) {
        const ggml_backend_buffer * b = t->buffer ? t->buffer : (t->view_src ? t->view_src->buffer : nullptr);
        if (b && std::find(p->buffers.begin(), p->buffers.end(), b) == p->buffers.end()) p->buffers.push_back(b);
    };
    for (int i = 0; i < g->n_nodes; i++) note(g->nodes[i]);
    for (int i = 0; i < g->n_leafs; i++) note(g->leafs[i]);
}
static void evict_plans_of_buffer(hip_ctx * c, const ggml_backend_buffer * b) {
    for (auto & kv : c->plans) {
        plan_t * p = kv.second;
        if (p->orphan_seq || std::find(p->buffers.begin(), p->buffers.end(), b) == p->buffers.end()) continue;
        p->orphan_seq = ++c->orphan_clock;
        p->buffers.clear();
    }
    for (;;) {   // bound the orphans
        int n = 0; auto oldest = c->plans.end();
        for (auto it = c->plans.begin(); it != c->plans.end(); ++it)
            if (it->second->orphan_seq) { n++; if (oldest == c->plans.end() || it->second->orphan_seq < oldest->second->orphan_seq) oldest = it; }
        if (n <= MAX_ORPHANS) break;
        if (c->stream) { set_device(c); HIP_CHECK(hipStreamSynchronize(c->stream)); }
        plan_free(c, oldest->second);
        c->plans.erase(oldest);
    }
}

// Everything the planner reads goes into the hash: for every node AND leaf the tensor's address and its whole descriptor up to (not including)
// the name - type, ne, nb, op, op_params, flags, the src pointers, view_src / view_offs, data (tensors are zero-filled on creation, so the
// padding bytes are defined). A source is either a node or a leaf of the same graph, so its shape / type / strides / data are covered too.
// Four interleaved multiply-xor lanes keep this at ~15 us for the 1.3 k-node Temporal graph.
static uint64_t graph_hash(const ggml_cgraph * g) {
    constexpr size_t NW = offsetof(ggml_tensor, name) / 8;
    static_assert(offsetof(ggml_tensor, name) % 32 == 0, "tensor descriptor is hashed in 4 lanes of 8-byte words");
    uint64_t h[4] = { 0xcbf29ce484222325ull ^ (uint64_t) g->n_nodes, 0x9e3779b97f4a7c15ull ^ (uint64_t) g->n_leafs, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull };
    auto mix = [&](const ggml_tensor * t) {
        uint64_t w[NW];
        memcpy(w, t, NW * 8);
        w[offsetof(ggml_tensor, buffer) / 8] = 0;   // the buffer OBJECT is not read by the planner (data addresses are): a re-allocated scratch buffer keeps the hash
        h[0] = (h[0] ^ (uint64_t) (uintptr_t) t) * 0x100000001b3ull;
        for (size_t i = 0; i < NW; i += 4) {
            h[0] = (h[0] ^ w[i]) * 0x100000001b3ull; h[1] = (h[1] ^ w[i + 1]) * 0x100000001b3ull;
            h[2] = (h[2] ^ w[i + 2]) * 0x100000001b3ull; h[3] = (h[3] ^ w[i + 3]) * 0x100000001b3ull;
        }
    };
    for (int i = 0; i < g->n_nodes; i++) mix(g->nodes[i]);
    for (int i = 0; i < g->n_leafs; i++) mix(g->leafs[i]);
    uint64_t r = h[0];
    for (int k = 1; k < 4; k++) r = (r ^ (h[k] >> 29) ^ h[k]) * 0x100000001b3ull;
    return r;
}

// ---- graph analysis helpers --------------------------------------------------------------------------
struct analysis {
    const ggml_cgraph * g;
    std::unordered_map<const ggml_tensor *, int> index;          // node -> position
    std::unordered_map<const ggml_tensor *, int> uses;           // tensor -> number of consuming nodes
    std::unordered_map<const ggml_tensor *, int> last_consumer;  // tensor -> position of its (last) consumer
    std::vector<char> skip;
};

static void analyse(analysis & an, const ggml_cgraph * g) {
    an.g = g;
    an.skip.assign((size_t) g->n_nodes, 0);
    for (int i = 0; i < g->n_nodes; i++) {
        const ggml_tensor * n = g->nodes[i];
        an.index[n] = i;
        for (int s = 0; s < GGML_MAX_SRC; s++) {
            const ggml_tensor * src = n->src[s];
            if (!src || src == n) continue;
            bool dup = false;
            for (int s2 = 0; s2 < s; s2++) if (n->src[s2] == src) dup = true;
            if (dup) continue;
            an.uses[src]++;
            an.last_consumer[src] = i;
        }
    }
}
static int uses_of(const analysis & an, const ggml_tensor * t) { auto it = an.uses.find(t); return it == an.uses.end() ? 0 : it->second; }
static int pos_of(const analysis & an, const ggml_tensor * t) { auto it = an.index.find(t); return it == an.index.end() ? -1 : it->second; }
static const ggml_tensor * sole_consumer(const analysis & an, const ggml_tensor * t) {
    if (uses_of(an, t) != 1) return nullptr;
    return an.g->nodes[an.last_consumer.at(t)];
}

static bool is_view_op(enum ggml_op op) { return op == GGML_OP_VIEW || op == GGML_OP_RESHAPE || op == GGML_OP_PERMUTE || op == GGML_OP_TRANSPOSE || op == GGML_OP_NONE; }
static bool is_f32_vec(const ggml_tensor * t, int64_t n) {
    return t->type == GGML_TYPE_F32 && ggml_nelements(t) == n && ggml_is_contiguous(t);
}
static bool dense_rows(const ggml_tensor * w) {
    return w->nb[0] == ggml_type_size(w->type) && w->nb[1] == ggml_row_size(w->type, w->ne[0]) && w->ne[2] == 1 && w->ne[3] == 1;
}

// pointer from which nelements(t) values can be read in t's logical order, looking through layout-only
// nodes and plain dense copies (cont of an already contiguous tensor); NULL when t is not contiguous
static const char * resolve_dense(const ggml_tensor * t) {
    if (!t->data || !ggml_is_contiguous(t)) return nullptr;
    const ggml_tensor * cur = t;
    int64_t delta = 0;
    while (cur->op == GGML_OP_VIEW || cur->op == GGML_OP_RESHAPE || cur->op == GGML_OP_PERMUTE || cur->op == GGML_OP_TRANSPOSE) {
        const ggml_tensor * s = cur->src[0];
        if (!s->data) return nullptr;
        delta += (const char *) cur->data - (const char *) s->data;
        cur = s;
    }
    // cur owns the storage; t's contents are the contiguous byte range cur->data + delta
    if ((cur->op == GGML_OP_CONT || cur->op == GGML_OP_DUP) && cur->src[0]->type == cur->type && ggml_is_contiguous(cur) &&
        ggml_is_contiguous(cur->src[0]) && ggml_nelements(cur->src[0]) == ggml_nelements(cur)) {
        const char * base = resolve_dense(cur->src[0]);   // same bytes, one copy earlier
        if (base) return base + delta;
    }
    return (const char *) cur->data + delta;
}

// Strided view of a tensor's elements in the storage of the earliest tensor that still holds the same values:
// walks through layout-only nodes and through cont/dup copies (mapping strides across the copy). On return element
// (i0,i1,i2,i3) of `t` lives at base + sum i_k * nb[k].
struct sview { const char * base; int64_t ne[4]; int64_t nb[4]; };

static bool strided_resolve(const ggml_tensor * t, sview & sv) {
    if (!t->data) return false;
    sv.base = (const char *) t->data;
    for (int i = 0; i < 4; i++) { sv.ne[i] = t->ne[i]; sv.nb[i] = (int64_t) t->nb[i]; }
    const int64_t es = (int64_t) ggml_type_size(t->type);
    const ggml_tensor * cur = t;
    for (int guard = 0; guard < 8; guard++) {
        while (cur->op == GGML_OP_VIEW || cur->op == GGML_OP_RESHAPE || cur->op == GGML_OP_PERMUTE || cur->op == GGML_OP_TRANSPOSE) cur = cur->src[0];
        if (!(cur->op == GGML_OP_CONT || cur->op == GGML_OP_DUP) || cur->src[0]->type != cur->type || !ggml_is_contiguous(cur) || !cur->data || !cur->src[0]->data) break;
        const ggml_tensor * src = cur->src[0];
        // position of our base inside the dense copy -> logical coordinates of the copy
        int64_t eo = (sv.base - (const char *) cur->data) / es;
        if (eo < 0 || eo >= ggml_nelements(cur)) break;
        int64_t P[5] = { 1, cur->ne[0], cur->ne[0] * cur->ne[1], cur->ne[0] * cur->ne[1] * cur->ne[2], ggml_nelements(cur) };
        int64_t c[4];
        for (int j = 3; j >= 0; j--) { c[j] = eo / P[j]; eo -= c[j] * P[j]; }
        int64_t nb2[4];
        bool ok = true;
        for (int k = 0; k < 4 && ok; k++) {
            nb2[k] = 0;
            if (sv.ne[k] <= 1) continue;
            if (sv.nb[k] % es != 0) { ok = false; break; }
            const int64_t E = sv.nb[k] / es;
            int j = 3;
            while (j > 0 && (P[j] > E || E % P[j] != 0)) j--;
            const int64_t m = E / P[j];
            if (c[j] + m * (sv.ne[k] - 1) >= cur->ne[j]) { ok = false; break; }   // the dim must stay inside one dim of the copy
            nb2[k] = m * (int64_t) src->nb[j];
        }
        if (!ok) break;
        sv.base = (const char *) src->data + c[0] * (int64_t) src->nb[0] + c[1] * (int64_t) src->nb[1] + c[2] * (int64_t) src->nb[2] + c[3] * (int64_t) src->nb[3];
        for (int k = 0; k < 4; k++) sv.nb[k] = nb2[k];
        cur = src;
    }
    return true;
}

// ---- emission ------------------------------------------------------------------------------------------
struct emitter {
    hip_ctx * c;
    plan_t * p;
    void * ws(size_t n) {
        size_t actual = 0;
        void * w = pool_alloc(c, n, &actual);
        GGML_ASSERT(w);
        p->workspaces.push_back({ w, actual });
        return w;
    }
    void push(step_fn f) { p->steps.push_back(pstep(std::move(f))); }
    // Codec convolutions without an im2col launch (match_conv): the output tensor of a conv / transposed-conv group -> a slot its launch reads when it runs.
    // A later conv whose input that tensor is claims the slot (fills it in): the producing launch then also writes the consumer's F16 im2col panel
    // (conv_scatter, hip_common.h), and the consumer starts from the panel.
    std::map<const ggml_tensor *, conv_scatter *> scatter_of;
    conv_scatter * scatter_slot(const ggml_tensor * out) { p->scatter_slots.emplace_back(new conv_scatter()); memset(p->scatter_slots.back().get(), 0, sizeof(conv_scatter)); scatter_of[out] = p->scatter_slots.back().get(); return scatter_of[out]; }
};

static bool is_qblock(enum ggml_type t) { return t == GGML_TYPE_Q4_K || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_0; }

static bool fill_matvec_base(mv_args & a, const ggml_tensor * mm) {
    const ggml_tensor * w = mm->src[0], * b = mm->src[1];
    if (!dense_rows(w) || !k_matvec_supported(w->type, w->ne[0], w->ne[1])) return false;
    if (b->type != GGML_TYPE_F32 || b->ne[2] != 1 || b->ne[3] != 1 || b->ne[1] < 1) return false;
    if (b->ne[1] > (is_qblock(w->type) ? 1 : MV_MAX_COLS)) return false;
    if (b->nb[0] != 4 || mm->nb[0] != 4) return false;
    memset(&a, 0, sizeof(a));
    a.ncols = (int) b->ne[1];
    a.x_cs = (int64_t) b->nb[1] / 4;
    a.y_cs = (int64_t) mm->nb[1] / 4;
    a.wtype = w->type;
    a.w = (const char *) w->data;
    a.row_bytes = (int64_t) w->nb[1];
    a.K = w->ne[0];
    a.M = w->ne[1];
    a.prologue = MV_PLAIN;
    a.y = (float *) mm->data;
    return true;
}

static void emit_generic(emitter & em, ggml_tensor * n) {
    const tdesc d = make_tdesc(n);
    switch (n->op) {
        case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return;
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: {
            const tdesc a = make_tdesc(n->src[0]), b = make_tdesc(n->src[1]); const int op = n->op;
            em.push([=](hipStream_t s) { k_binary(s, op, d, a, b); });
        } return;
        case GGML_OP_UNARY: {
            const tdesc a = make_tdesc(n->src[0]); const int uop = n->op_params[0];
            em.push([=](hipStream_t s) { k_unary(s, uop, d, a); });
        } return;
        case GGML_OP_SCALE: {
            const tdesc a = make_tdesc(n->src[0]); const float sc = ggml_get_op_params_f32(n, 0), bi = ggml_get_op_params_f32(n, 1);
            em.push([=](hipStream_t s) { k_scale(s, d, a, sc, bi); });
        } return;
        case GGML_OP_CLAMP: {
            const tdesc a = make_tdesc(n->src[0]); const float mn = ggml_get_op_params_f32(n, 0), mx = ggml_get_op_params_f32(n, 1);
            em.push([=](hipStream_t s) { k_clamp(s, d, a, mn, mx); });
        } return;
        case GGML_OP_SUM: { const tdesc a = make_tdesc(n->src[0]); em.push([=](hipStream_t s) { k_sum_all(s, d, a); }); } return;
        case GGML_OP_SUM_ROWS: { const tdesc a = make_tdesc(n->src[0]); em.push([=](hipStream_t s) { k_sum_rows(s, d, a); }); } return;
        case GGML_OP_ARGMAX: { const tdesc a = make_tdesc(n->src[0]); em.push([=](hipStream_t s) { k_argmax(s, d, a); }); } return;
        case GGML_OP_ARGSORT: case GGML_OP_TOP_K: {
            const tdesc a = make_tdesc(n->src[0]); const int desc = n->op_params[0] == GGML_SORT_ORDER_DESC;
            em.push([=](hipStream_t s) { k_argsort(s, d, a, desc); });
        } return;
        case GGML_OP_ARANGE: {
            const float st = ggml_get_op_params_f32(n, 0), step = ggml_get_op_params_f32(n, 2);
            em.push([=](hipStream_t s) { k_arange(s, d, st, step); });
        } return;
        case GGML_OP_REPEAT: { const tdesc a = make_tdesc(n->src[0]); em.push([=](hipStream_t s) { k_repeat(s, d, a); }); } return;
        case GGML_OP_PAD: { const tdesc a = make_tdesc(n->src[0]); em.push([=](hipStream_t s) { k_pad(s, d, a); }); } return;
        case GGML_OP_CONCAT: {
            const tdesc a = make_tdesc(n->src[0]), b = make_tdesc(n->src[1]); const int dim = n->op_params[0];
            em.push([=](hipStream_t s) { k_concat(s, d, a, b, dim); });
        } return;
        case GGML_OP_NORM: case GGML_OP_RMS_NORM: {
            const tdesc a = make_tdesc(n->src[0]); const float eps = ggml_get_op_params_f32(n, 0); const int rms = n->op == GGML_OP_RMS_NORM;
            em.push([=](hipStream_t s) { k_norm(s, d, a, eps, rms); });
        } return;
        case GGML_OP_SOFT_MAX: {
            const tdesc a = make_tdesc(n->src[0]); const int has_mask = n->src[1] != NULL;
            const tdesc m = has_mask ? make_tdesc(n->src[1]) : a; const float sc = ggml_get_op_params_f32(n, 0);
            GGML_ASSERT(ggml_get_op_params_f32(n, 1) == 0.0f && "ALiBi (max_bias) is not used by moshi.cpp");
            em.push([=](hipStream_t s) { k_soft_max(s, d, a, m, has_mask, sc); });
        } return;
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            const tdesc a = make_tdesc(n->src[0]);
            em.push([=](hipStream_t s) { k_cpy(s, d, a); });
        } return;
        case GGML_OP_GET_ROWS: {
            const tdesc a = make_tdesc(n->src[0]), idx = make_tdesc(n->src[1]);
            em.push([=](hipStream_t s) { k_get_rows(s, d, a, idx); });
        } return;
        case GGML_OP_SET_ROWS: {
            const tdesc src = make_tdesc(n->src[0]), idx = make_tdesc(n->src[1]);
            em.push([=](hipStream_t s) { k_set_rows(s, d, src, idx); });
        } return;
        case GGML_OP_IM2COL: {
            const tdesc x = make_tdesc(n->src[1]); const int64_t K = n->src[0]->ne[0];
            const int s0 = n->op_params[0], p0 = n->op_params[2], d0 = n->op_params[4];
            em.push([=](hipStream_t s) { k_im2col(s, d, x, K, s0, p0, d0); });
        } return;
        case GGML_OP_CONV_TRANSPOSE_1D: {
            const tdesc w = make_tdesc(n->src[0]), x = make_tdesc(n->src[1]); const int s0 = n->op_params[0];
            void * ws = em.ws(k_conv_transpose_1d_ws_size(n->src[0], n->src[1]));
            em.push([=](hipStream_t s) { k_conv_transpose_1d(s, d, w, x, s0, ws); });
        } return;
        case GGML_OP_TIMESTEP_EMBEDDING: {
            const tdesc ts = make_tdesc(n->src[0]); const int dim = n->op_params[0], mp = n->op_params[1];
            em.push([=](hipStream_t s) { k_timestep_embedding(s, d, ts, dim, mp); });
        } return;
        case GGML_OP_MUL_MAT: {
            mv_args mv;
            if (!(em.c->flags & 1) && fill_matvec_base(mv, n)) {
                mv.x = (const float *) n->src[1]->data;
                em.push([=](hipStream_t s) { k_matvec(s, mv); });
                return;
            }
            {   // several activation rows against Q4_K weights (batched prompt prefill): int8 MFMA tiles
                const ggml_tensor * w0 = n->src[0], * x1 = n->src[1];
                const int64_t Tn = ggml_nelements(x1) / x1->ne[0];
                static const bool no_batched = getenv("MI355X_NO_BATCHED_MM") != nullptr;
                if (!no_batched && x1->type == GGML_TYPE_F32 && ggml_is_contiguous(x1) && ggml_is_contiguous(n) && ggml_is_contiguous(w0) && w0->ne[2] == 1 && w0->ne[3] == 1 &&
                    k_mm_q4k_batched_supported(w0->type, w0->ne[0], w0->ne[1], Tn)) {
                    const char * wp = (const char *) w0->data; const int64_t rb = (int64_t) w0->nb[1], K = w0->ne[0], M = w0->ne[1];
                    const float * xp = (const float *) x1->data; float * yp = (float *) n->data;
                    void * ws = em.ws(k_mm_q4k_batched_ws_size(K, Tn));
                    const int wt = (int) w0->type;
                    em.push([=](hipStream_t s) { k_mm_q4k_batched(s, wt, wp, rb, K, M, Tn, xp, K, ws, yp, M, nullptr, 0); });
                    return;
                }
            }
            const tdesc a = make_tdesc(n->src[0]), b = make_tdesc(n->src[1]);
            void * w = em.ws(k_mul_mat_ws_size(n->src[0], n->src[1]));
            em.push([=](hipStream_t s) { k_mul_mat(s, d, a, b, w); });
        } return;
        default:
            GGML_ABORT("mi355x backend: unsupported op %s (node %s)", ggml_op_name(n->op), n->name);
    }
}

// ---- fusion matchers --------------------------------------------------------------------------------------

// A. mat-vec with fused activation prologue and residual epilogue. Returns the position at which the
//    fused kernel must be emitted (the last node of the group), or -1.
struct mv_group { mv_args a; int emit_pos; std::vector<int> members; };
static bool match_embed_term(const analysis & an, const ggml_tensor * e, embed_src & out, std::vector<int> & members);

static bool writes_through_alias(const ggml_tensor * n) {
    return n->view_src != NULL && !is_view_op(n->op);
}

static bool match_matvec(const analysis & an, int pos, mv_group & grp) {
    ggml_tensor * mm = an.g->nodes[pos];
    if (mm->op != GGML_OP_MUL_MAT || !fill_matvec_base(grp.a, mm)) return false;
    mv_args & a = grp.a;
    const ggml_tensor * b = mm->src[1];
    const int64_t K = a.K;
    grp.members.clear();
    grp.members.push_back(pos);
    grp.emit_pos = pos;
    bool have_x = false;
    // per-step weight selection wraps the activation in a whole-tensor view (transformer.h:85-91): look through it
    while ((b->op == GGML_OP_VIEW || b->op == GGML_OP_RESHAPE) && uses_of(an, b) == 1 && b->src[0]->data == b->data &&
           ggml_nelements(b->src[0]) == ggml_nelements(b) && ggml_is_contiguous(b->src[0])) b = b->src[0];

    // prologue 1: b = alpha * rms_norm(x), private to this mat-vec - or shared with exactly one cpy(b -> contiguous F32 tensor) issued right before the
    // mat-vec (the Temporal head: out_norm feeds text_linear and is kept as the `transformer_out` state, lm.h:434): workgroup 0 writes the copy
    const ggml_tensor * side_cpy = nullptr;
    if (a.ncols == 1 && b->op == GGML_OP_MUL && uses_of(an, b) == 2 && pos > 0 && is_qblock(mm->src[0]->type)) {
        const ggml_tensor * cp = an.g->nodes[pos - 1];
        if (cp->op == GGML_OP_CPY && cp->src[0] == b && cp->type == GGML_TYPE_F32 && ggml_is_contiguous(cp) && ggml_nelements(cp) == K && cp->data &&
            !an.skip[(size_t) (pos - 1)]) side_cpy = cp;
    }
    if (a.ncols == 1 && b->op == GGML_OP_MUL && is_f32_vec(b, K) && (uses_of(an, b) == 1 || side_cpy) && (!is_qblock(mm->src[0]->type) || K <= 4096)) {
        const ggml_tensor * al = b->src[0], * nr = b->src[1];
        if (nr->op != GGML_OP_RMS_NORM) std::swap(al, nr);
        if (nr->op == GGML_OP_RMS_NORM && uses_of(an, nr) == 1 && is_f32_vec(al, K) && is_f32_vec(nr->src[0], K)) {
            a.prologue = MV_RMSNORM;
            a.x = (const float *) nr->src[0]->data;
            a.alpha = (const float *) al->data;
            a.eps = ggml_get_op_params_f32(nr, 0);
            grp.members.push_back(pos_of(an, b)); grp.members.push_back(pos_of(an, nr));
            if (side_cpy) { a.x_out = (float *) side_cpy->data; grp.members.push_back(pos - 1); }
            have_x = true;
        }
    }
    // prologue 1b: b = norm(x) * w (+ bias) (torch_nn_layer_norm, torch.h:49-60), any column count
    if (!have_x && !is_qblock(mm->src[0]->type) && uses_of(an, b) == 1 && (b->op == GGML_OP_ADD || b->op == GGML_OP_MUL)) {
        const ggml_tensor * mulw = b, * bias = nullptr;
        if (b->op == GGML_OP_ADD && b->src[0]->op == GGML_OP_MUL && uses_of(an, b->src[0]) == 1) { mulw = b->src[0]; bias = b->src[1]; }
        if (mulw->op == GGML_OP_MUL && mulw->src[0]->op == GGML_OP_NORM && uses_of(an, mulw->src[0]) == 1) {
            const ggml_tensor * nr = mulw->src[0], * wgt = mulw->src[1], * xin = nr->src[0];
            if (is_f32_vec(wgt, K) && (!bias || is_f32_vec(bias, K)) && xin->type == GGML_TYPE_F32 && xin->ne[0] == K && xin->ne[1] == a.ncols &&
                xin->nb[0] == 4 && ggml_nelements(xin) == K * a.ncols && ggml_are_same_shape(b, xin)) {
                a.prologue = MV_LAYERNORM;
                a.x = (const float *) xin->data;
                a.x_cs = (int64_t) xin->nb[1] / 4;
                a.alpha = (const float *) wgt->data;
                a.beta = bias ? (const float *) bias->data : nullptr;
                a.eps = ggml_get_op_params_f32(nr, 0);
                grp.members.push_back(pos_of(an, nr)); grp.members.push_back(pos_of(an, mulw));
                if (bias) grp.members.push_back(pos_of(an, b));
                have_x = true;
            }
        }
    }
    // prologue 1c: b = gelu(h)
    if (!have_x && !is_qblock(mm->src[0]->type) && b->op == GGML_OP_UNARY && b->op_params[0] == GGML_UNARY_OP_GELU && uses_of(an, b) == 1 &&
        !an.skip[(size_t) pos_of(an, b)]) {   // (already produced by the previous mat-vec's epilogue otherwise)
        const ggml_tensor * hin = b->src[0];
        if (hin->type == GGML_TYPE_F32 && hin->nb[0] == 4 && ggml_are_same_shape(hin, b) && hin->ne[2] == 1 && hin->ne[3] == 1) {
            a.prologue = MV_GELU;
            a.x = (const float *) hin->data;
            a.x_cs = (int64_t) hin->nb[1] / 4;
            grp.members.push_back(pos_of(an, b));
            have_x = true;
        }
    }
    // prologue 2: b = silu(left(h)) * right(h)
    if (!have_x && a.ncols == 1 && b->op == GGML_OP_MUL && uses_of(an, b) == 1 && ggml_nelements(b) == K) {
        const ggml_tensor * sl = b->src[0], * r = b->src[1];
        if (sl->op == GGML_OP_UNARY && sl->op_params[0] == GGML_UNARY_OP_SILU && uses_of(an, sl) == 1 &&
            sl->src[0]->op == GGML_OP_VIEW && r->op == GGML_OP_VIEW && uses_of(an, sl->src[0]) == 1 && uses_of(an, r) == 1) {
            const ggml_tensor * l = sl->src[0];
            const ggml_tensor * h = l->src[0];
            if (r->src[0] == h && is_f32_vec(h, 2 * K) && l->data == h->data && (const char *) r->data == (const char *) h->data + K * 4 &&
                ggml_is_contiguous(l) && ggml_is_contiguous(r)) {
                a.prologue = MV_GATE_SILU;
                a.x = (const float *) h->data;
                grp.members.push_back(pos_of(an, b)); grp.members.push_back(pos_of(an, sl));
                have_x = true;
            }
        }
    }
    if (!have_x) {   // plain activation: address it through the mat-vec's own operand (a looked-through view may be shaped differently)
        a.x = (const float *) mm->src[1]->data;
        a.x_cs = (int64_t) mm->src[1]->nb[1] / 4;
    }
    // epilogue 0: gelu(W x) (the activation is evaluated once per output element here, not once per consuming workgroup)
    {
        const ggml_tensor * g = sole_consumer(an, mm);
        if (g && g->op == GGML_OP_UNARY && g->op_params[0] == GGML_UNARY_OP_GELU && !is_qblock(mm->src[0]->type) && g->view_src == NULL &&
            ggml_are_same_shape(g, mm) && g->nb[0] == 4 && g->type == GGML_TYPE_F32) {
            a.out_act = 1;
            a.y = (float *) g->data;
            a.y_cs = (int64_t) g->nb[1] / 4;
            grp.members.push_back(pos_of(an, g));
            grp.emit_pos = pos_of(an, g);
            return true;
        }
    }
    // epilogue: (optional per-row layer_scale, then) the only consumer adds a same-shaped F32 tensor
    const ggml_tensor * cons = sole_consumer(an, mm);
    const ggml_tensor * scaled = nullptr;
    if (cons && cons->op == GGML_OP_MUL && cons->view_src == NULL && cons->src[0] == mm && !is_qblock(mm->src[0]->type) &&
        is_f32_vec(cons->src[1], a.M) && ggml_are_same_shape(cons, mm) && cons->nb[0] == 4) {
        const ggml_tensor * c2 = sole_consumer(an, cons);
        if (c2 && c2->op == GGML_OP_ADD) { scaled = cons; cons = c2; }
    }
    const ggml_tensor * prod = scaled ? scaled : mm;
    if (cons && cons->op == GGML_OP_ADD && cons->view_src == NULL) {
        const ggml_tensor * other = cons->src[0] == prod ? cons->src[1] : cons->src[0];
        const bool shapes = other != prod && other->type == GGML_TYPE_F32 && ggml_are_same_shape(other, mm) && ggml_are_same_shape(cons, mm) &&
                            other->nb[0] == 4 && cons->nb[0] == 4 && cons->type == GGML_TYPE_F32;
        // the addend is one embedding row (moshi_scaled_embedding_chained + cast, lm.h:446-475): gather it in the epilogue
        std::vector<int> emb_members;
        embed_src es;
        const ggml_tensor * term = other;
        bool embed = false;
        if (shapes && !scaled && is_qblock((enum ggml_type) a.wtype) && a.ncols == 1 && uses_of(an, other) == 1) {
            if (term->op == GGML_OP_CPY && term->view_src == NULL && term->src[0]->type == GGML_TYPE_F32 && ggml_are_same_shape(term->src[0], term) && uses_of(an, term->src[0]) == 1) {
                emb_members.push_back(pos_of(an, term));
                term = term->src[0];
            }
            embed = (term->op == GGML_OP_GET_ROWS || term->op == GGML_OP_MUL) && term->ne[1] == 1 && match_embed_term(an, term, es, emb_members);
            for (int m : emb_members) if (m < 0 || an.skip[(size_t) m]) embed = false;
        }
        if (shapes && embed) for (int i = pos + 1; i < pos_of(an, cons); i++) if (writes_through_alias(an.g->nodes[i])) embed = false;
        if (shapes && embed) {
            const int cpos = pos_of(an, cons);
            a.res_embed = es;
            a.y = (float *) cons->data;
            a.y_cs = (int64_t) cons->nb[1] / 4;
            for (int m : emb_members) grp.members.push_back(m);
            grp.members.push_back(cpos);
            grp.emit_pos = cpos;
        } else if (shapes) {
            const int cpos = pos_of(an, cons);
            bool hazard = false;
            for (int i = pos + 1; i < cpos; i++) if (writes_through_alias(an.g->nodes[i])) hazard = true;
            if (!hazard) {
                a.residual = (const float *) other->data;
                a.r_cs = (int64_t) other->nb[1] / 4;
                a.y = (float *) cons->data;
                a.y_cs = (int64_t) cons->nb[1] / 4;
                if (scaled) { a.out_scale = (const float *) scaled->src[1]->data; grp.members.push_back(pos_of(an, scaled)); }
                grp.members.push_back(cpos);
                grp.emit_pos = cpos;
            }
        }
    }
    return true;
}

// B. single-token attention block
struct attn_group { attn_args a; int emit_pos; std::vector<int> members; const ggml_tensor * mask_node; };

static const ggml_tensor * strip_views(const ggml_tensor * t) {
    while (t && (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE)) t = t->src[0];
    return t;
}

// rotated = concat(sub(mul(r, rotr), mul(i, roti)), add(mul(r, roti), mul(i, rotr))); returns the un-rotated source
static bool match_rope(const ggml_tensor * rotated, const ggml_tensor ** src_out, const ggml_tensor ** rotr_out, const ggml_tensor ** roti_out) {
    if (rotated->op != GGML_OP_CONCAT || rotated->op_params[0] != 0) return false;
    const ggml_tensor * re = rotated->src[0], * im = rotated->src[1];
    if (re->op != GGML_OP_SUB || im->op != GGML_OP_ADD) return false;
    const ggml_tensor * m0 = re->src[0], * m1 = re->src[1], * m2 = im->src[0], * m3 = im->src[1];
    if (m0->op != GGML_OP_MUL || m1->op != GGML_OP_MUL || m2->op != GGML_OP_MUL || m3->op != GGML_OP_MUL) return false;
    const ggml_tensor * xr = m0->src[0], * rotr = m0->src[1], * xi = m1->src[0], * roti = m1->src[1];
    if (m2->src[0] != xr || m2->src[1] != roti || m3->src[0] != xi || m3->src[1] != rotr) return false;
    // xr / xi = reshape(view(Z, half 0 / 1)), Z = cont(permute(reshape(cont(src), 2, D/2, T, BH), 3, 0, 1, 2))
    if (xr->op != GGML_OP_RESHAPE || xi->op != GGML_OP_RESHAPE) return false;
    const ggml_tensor * vr = xr->src[0], * vi = xi->src[0];
    if (vr->op != GGML_OP_VIEW || vi->op != GGML_OP_VIEW || vr->src[0] != vi->src[0]) return false;
    const ggml_tensor * Z = vr->src[0];
    if (Z->op != GGML_OP_CONT || vr->data != Z->data || (const char *) vi->data != (const char *) Z->data + ggml_nbytes(Z) / 2) return false;
    const ggml_tensor * P = Z->src[0];
    if (P->op != GGML_OP_PERMUTE || P->op_params[0] != 3 || P->op_params[1] != 0 || P->op_params[2] != 1 || P->op_params[3] != 2) return false;
    const ggml_tensor * R = P->src[0];
    if (R->op != GGML_OP_RESHAPE || R->ne[0] != 2) return false;
    *src_out = R->src[0];
    *rotr_out = rotr;
    *roti_out = roti;
    return true;
}

static bool match_attention(const analysis & an, int pos, attn_group & grp) {
    const ggml_tensor * sm = an.g->nodes[pos];
    if (sm->op != GGML_OP_SOFT_MAX || !sm->src[1]) return false;
    const ggml_tensor * kq = sm->src[0], * mask = sm->src[1];
    if (kq->op != GGML_OP_MUL_MAT || uses_of(an, kq) != 1 || uses_of(an, sm) != 1) return false;
    const ggml_tensor * kc = kq->src[0], * qo = kq->src[1];
    if (kc->op != GGML_OP_SET_ROWS || kc->type != GGML_TYPE_BF16) return false;
    const ggml_tensor * pv = sole_consumer(an, sm);
    if (!pv || pv->op != GGML_OP_MUL_MAT || pv->src[1] != sm) return false;
    const ggml_tensor * vt = pv->src[0];
    if (vt->op != GGML_OP_CONT || vt->src[0]->op != GGML_OP_TRANSPOSE) return false;
    const ggml_tensor * vc = vt->src[0]->src[0];
    if (vc->op != GGML_OP_SET_ROWS || vc->type != GGML_TYPE_BF16) return false;
    const ggml_tensor * x1 = sole_consumer(an, pv);
    if (!x1 || x1->op != GGML_OP_PERMUTE) return false;
    const ggml_tensor * x2 = sole_consumer(an, x1);
    if (!x2 || x2->op != GGML_OP_CONT) return false;

    const int64_t D = kc->ne[0], C = kc->ne[1], H = kc->ne[2], Tn = qo->ne[1];
    if (qo->ne[0] != D || Tn < 1 || Tn > 64 || qo->ne[2] != H || qo->ne[3] != 1) return false;   // B == 1; blocks longer than 4 rows (prompt prefill) run as consecutive launches of 4
    if (vc->ne[0] != D || vc->ne[1] != C || vc->ne[2] != H) return false;
    if (mask->type != GGML_TYPE_F32 || mask->ne[0] != C || mask->ne[1] != Tn || !ggml_is_contiguous(mask)) return false;
    if (D % 8 != 0 || 64 % (D / 8) != 0 || D > 256) return false;
    if (kc->src[1] != vc->src[1]) return false;
    const ggml_tensor * idx = kc->src[1];
    if (idx->type != GGML_TYPE_I32 || ggml_nelements(idx) != Tn || !ggml_is_contiguous(idx)) return false;
    if (kc->nb[0] != 2 || vc->nb[0] != 2) return false;

    const ggml_tensor * ko = kc->src[0], * vrow = vc->src[0];
    const ggml_tensor * qsrc = qo, * ksrc = ko, * rotr = nullptr, * roti = nullptr;
    if (qo->op == GGML_OP_CONCAT) {
        const ggml_tensor * rotr2, * roti2;
        if (!match_rope(qo, &qsrc, &rotr, &roti) || !match_rope(ko, &ksrc, &rotr2, &roti2) || rotr != rotr2 || roti != roti2) return false;
        if (rotr->type != GGML_TYPE_F32 || rotr->ne[0] != D / 2 || rotr->ne[1] != Tn || (const char *) roti->data != (const char *) rotr->data + D / 2 * 4) return false;
        if (rotr->nb[0] != 4 || (Tn > 1 && (int64_t) rotr->nb[1] != D * 4)) return false;
    }
    if (qsrc->type != GGML_TYPE_F32 || ksrc->type != GGML_TYPE_F32 || vrow->type != GGML_TYPE_F32) return false;
    for (const ggml_tensor * t : { qsrc, ksrc, vrow }) if (t->ne[0] != D || t->ne[1] != Tn || t->ne[2] != H || t->ne[3] != 1) return false;
    sview qs, ks, vs;
    if (!strided_resolve(qsrc, qs) || !strided_resolve(ksrc, ks) || !strided_resolve(vrow, vs)) return false;
    for (const sview * v : { &qs, &ks, &vs }) if (v->nb[0] != 4 || v->nb[1] % 4 != 0 || v->nb[2] % 4 != 0) return false;
    const ggml_tensor * out_t = x2;
    if (out_t->ne[0] != D || out_t->ne[1] != H || out_t->ne[2] != Tn || !ggml_is_contiguous(out_t)) return false;

    // collect the interior of the block: everything reachable from x2 down to the block inputs
    std::vector<const ggml_tensor *> stack = { x2 };
    std::vector<int> members;
    std::unordered_map<const ggml_tensor *, bool> seen;
    int n_mm = 0, n_sm = 0, n_sr = 0;
    const int lo = pos - 200 > 0 ? pos - 200 : 0;
    while (!stack.empty()) {
        const ggml_tensor * t = stack.back(); stack.pop_back();
        if (seen[t]) continue;
        seen[t] = true;
        const int p = pos_of(an, t);
        if (p < 0) continue;                                  // leaf: cache, mask, indices, weights
        if (t == mask || t == idx || t == rotr || t == roti) continue;
        if ((const char *) t->data == nullptr) return false;
        // the projection output (and anything before it) is an input of the block, not part of it
        const bool is_input = (t->op == GGML_OP_MUL_MAT && t != kq && t != pv) || t->op == GGML_OP_TIMESTEP_EMBEDDING;
        if (is_input) continue;
        if (p < lo) return false;                              // wandered out of the layer: refuse
        switch (t->op) {
            case GGML_OP_MUL_MAT: n_mm++; break;
            case GGML_OP_SOFT_MAX: n_sm++; break;
            case GGML_OP_SET_ROWS: n_sr++; break;
            case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: case GGML_OP_CONT:
            case GGML_OP_MUL: case GGML_OP_SUB: case GGML_OP_ADD: case GGML_OP_CONCAT: break;
            default: return false;
        }
        members.push_back(p);
        for (int s = 0; s < GGML_MAX_SRC; s++) if (t->src[s]) stack.push_back(t->src[s]);
    }
    if (n_mm != 2 || n_sm != 1 || n_sr != 2) return false;
    // no interior value may be consumed outside the block (x2 is the block's output)
    std::unordered_map<int, bool> inside;
    for (int p : members) inside[p] = true;
    for (int i = 0; i < an.g->n_nodes; i++) {
        if (inside.count(i)) continue;
        const ggml_tensor * c = an.g->nodes[i];
        for (int s = 0; s < GGML_MAX_SRC; s++) {
            const ggml_tensor * t = c->src[s];
            if (!t || t == x2) continue;
            const int tp = pos_of(an, t);
            if (tp >= 0 && inside.count(tp)) return false;
        }
    }
    attn_args & a = grp.a;
    memset(&a, 0, sizeof(a));
    a.q = (const float *) qs.base; a.k = (const float *) ks.base; a.v = (const float *) vs.base;
    a.q_ts = qs.nb[1] / 4; a.q_hs = qs.nb[2] / 4; a.k_ts = ks.nb[1] / 4; a.k_hs = ks.nb[2] / 4; a.v_ts = vs.nb[1] / 4; a.v_hs = vs.nb[2] / 4;
    a.rot = rotr ? (const float *) rotr->data : nullptr;
    a.mask = (const float *) mask->data;
    grp.mask_node = nullptr;
    if (mask->op == GGML_OP_CONT && pos_of(an, mask) >= 0) {   // a dense copy of an already dense row: read the source
        const char * base = resolve_dense(mask);
        if (base && base != (const char *) mask->data) { a.mask = (const float *) base; grp.mask_node = mask; }
    }
    a.index = (const int32_t *) idx->data;
    a.kcache = (char *) kc->data; a.vcache = (char *) vc->data;
    a.k_nb1 = (int64_t) kc->nb[1]; a.k_nb2 = (int64_t) kc->nb[2];
    a.v_nb1 = (int64_t) vc->nb[1]; a.v_nb2 = (int64_t) vc->nb[2];
    a.H = (int) H; a.D = (int) D; a.C = (int) C; a.T = (int) Tn;
    a.scale = ggml_get_op_params_f32(sm, 0);
    a.out = (float *) x2->data;
    a.out_ts = H * D;
    grp.members = members;
    grp.emit_pos = pos_of(an, x2);
    return true;
}

// A'. several activation rows against Q4_K weights (batched prompt prefill): the int8-MFMA mat-mul with the activation producer
//     (alpha * rms_norm(x), or silu(h[:n]) * h[n:]) folded into its row quantiser and the residual add into its epilogue
struct bmm_group { int emit_pos; std::vector<int> members; std::function<void(hipStream_t)> run; };
static bool match_batched_mm(const analysis & an, int pos, emitter & em, bmm_group & grp) {
    const ggml_tensor * mm = an.g->nodes[pos];
    if (mm->op != GGML_OP_MUL_MAT) return false;
    const ggml_tensor * w0 = mm->src[0], * x1 = mm->src[1];
    if (!is_qblock(w0->type) || !ggml_is_contiguous(w0) || w0->ne[2] != 1 || w0->ne[3] != 1) return false;
    if (x1->type != GGML_TYPE_F32 || !ggml_is_contiguous(x1) || mm->type != GGML_TYPE_F32 || !ggml_is_contiguous(mm) || mm->view_src) return false;
    const int64_t K = w0->ne[0], M = w0->ne[1], Tn = ggml_nelements(x1) / K;
    if (!k_mm_q4k_batched_supported(w0->type, K, M, Tn)) return false;
    grp.members.assign(1, pos);
    grp.emit_pos = pos;
    int prologue = MV_PLAIN;
    const float * xp = (const float *) x1->data, * alpha = nullptr;
    int64_t x_cs = K;
    float eps = 0.f;
    const ggml_tensor * b = x1;
    while ((b->op == GGML_OP_VIEW || b->op == GGML_OP_RESHAPE) && uses_of(an, b) == 1 && b->src[0]->data == b->data &&
           ggml_nelements(b->src[0]) == ggml_nelements(b) && ggml_is_contiguous(b->src[0])) b = b->src[0];
    if (b->op == GGML_OP_MUL && uses_of(an, b) == 1 && b->view_src == NULL) {
        const ggml_tensor * s0 = b->src[0], * s1 = b->src[1];
        const ggml_tensor * nr = s0->op == GGML_OP_RMS_NORM ? s0 : s1->op == GGML_OP_RMS_NORM ? s1 : nullptr;
        const ggml_tensor * al = nr == s0 ? s1 : s0;
        if (nr && uses_of(an, nr) == 1 && is_f32_vec(al, K) && is_f32_vec(nr->src[0], K * Tn) && nr->src[0]->ne[0] == K) {
            prologue = MV_RMSNORM;
            xp = (const float *) nr->src[0]->data;
            alpha = (const float *) al->data;
            eps = ggml_get_op_params_f32(nr, 0);
            grp.members.push_back(pos_of(an, b)); grp.members.push_back(pos_of(an, nr));
        } else if (s0->op == GGML_OP_UNARY && s0->op_params[0] == GGML_UNARY_OP_SILU && uses_of(an, s0) == 1 && s0->src[0]->op == GGML_OP_VIEW &&
                   s1->op == GGML_OP_VIEW && uses_of(an, s0->src[0]) == 1 && uses_of(an, s1) == 1) {
            const ggml_tensor * l = s0->src[0], * r = s1, * h = l->src[0];
            if (r->src[0] == h && is_f32_vec(h, 2 * K * Tn) && h->ne[0] == 2 * K && l->data == h->data && (const char *) r->data == (const char *) h->data + K * 4 &&
                l->ne[0] == K && r->ne[0] == K && ggml_nelements(l) == K * Tn && ggml_nelements(r) == K * Tn &&
                l->ne[1] == 1 && r->ne[1] == 1 && l->nb[2] == h->nb[1] && r->nb[2] == h->nb[1] && l->ne[2] == Tn && r->ne[2] == Tn) {
                prologue = MV_GATE_SILU;
                xp = (const float *) h->data;
                x_cs = 2 * K;
                grp.members.push_back(pos_of(an, b)); grp.members.push_back(pos_of(an, s0));
            }
        }
    }
    float * yp = (float *) mm->data;
    const float * res = nullptr;
    {   // residual add, possibly behind the gating's shape fold
        const ggml_tensor * cur = mm, * c = sole_consumer(an, mm);
        while (c && (c->op == GGML_OP_RESHAPE || c->op == GGML_OP_VIEW) && c->data == cur->data && ggml_is_contiguous(c) && ggml_nelements(c) == ggml_nelements(cur)) { cur = c; c = sole_consumer(an, c); }
        if (c && c->op == GGML_OP_ADD && c->type == GGML_TYPE_F32 && ggml_is_contiguous(c) && c->view_src == NULL && ggml_nelements(c) == M * Tn && (c->src[0] == cur || c->src[1] == cur)) {
            const ggml_tensor * other = c->src[0] == cur ? c->src[1] : c->src[0];
            if (other != cur && is_f32_vec(other, M * Tn)) {
                res = (const float *) other->data;
                yp = (float *) c->data;
                grp.members.push_back(pos_of(an, c));
                grp.emit_pos = pos_of(an, c);
            }
        }
    }
    for (int m : grp.members) if (m < 0) return false;
    const char * wp = (const char *) w0->data;
    const int64_t rb = (int64_t) w0->nb[1];
    void * ws = em.ws(k_mm_q4k_batched_ws_size(K, Tn));
    const int wt = (int) w0->type;
    grp.run = [=](hipStream_t s) { k_mm_q4k_batched(s, wt, wp, rb, K, M, Tn, xp, x_cs, ws, yp, M, res, M, prologue, alpha, eps); };
    return true;
}

// B'. cross-attention over a cached condition, ending at x2 = cont(permute(mul_mat(cont(transpose(V)), soft_max(mul_mat(K, q)))))
struct xattn_group { xattn_args a; int emit_pos; std::vector<int> members; };
static bool match_cross_attention(const analysis & an, int pos, xattn_group & grp) {
    const ggml_tensor * x2 = an.g->nodes[pos];
    if (x2->op != GGML_OP_CONT || x2->type != GGML_TYPE_F32 || x2->src[0]->op != GGML_OP_PERMUTE || uses_of(an, x2->src[0]) != 1) return false;
    const ggml_tensor * x1 = x2->src[0], * pv = x1->src[0];
    if (pv->op != GGML_OP_MUL_MAT || uses_of(an, pv) != 1) return false;
    const ggml_tensor * vt = pv->src[0], * sm = pv->src[1];
    if (vt->op != GGML_OP_CONT || uses_of(an, vt) != 1 || vt->src[0]->op != GGML_OP_TRANSPOSE || uses_of(an, vt->src[0]) != 1) return false;
    const ggml_tensor * vx = vt->src[0]->src[0];
    if (sm->op != GGML_OP_SOFT_MAX || uses_of(an, sm) != 1 || sm->src[1] != NULL || ggml_get_op_params_f32(sm, 1) != 0.0f) return false;
    const ggml_tensor * kq = sm->src[0];
    if (kq->op != GGML_OP_MUL_MAT || uses_of(an, kq) != 1) return false;
    const ggml_tensor * kx = kq->src[0], * qp = kq->src[1];
    if (kx->type != GGML_TYPE_F32 || vx->type != GGML_TYPE_F32 || !kx->data || !vx->data || pos_of(an, kx) >= 0 || pos_of(an, vx) >= 0) return false;   // cached state, not graph values
    const int64_t D = kx->ne[0], Tc = kx->ne[1], H = kx->ne[2];
    if (vx->ne[0] != D || vx->ne[1] != Tc || vx->ne[2] != H || kx->ne[3] != 1 || kx->nb[0] != 4 || vx->nb[0] != 4) return false;
    // q: [D, 1, H] permuted view of a dense [D * H] vector
    if (qp->ne[0] != D || qp->ne[1] != 1 || qp->ne[2] != H || qp->ne[3] != 1 || qp->type != GGML_TYPE_F32 || qp->nb[0] != 4 || (int64_t) qp->nb[2] != D * 4) return false;
    const ggml_tensor * qsrc = qp;
    while (is_view_op(qsrc->op) && qsrc->op != GGML_OP_NONE && qsrc->src[0]) { if (uses_of(an, qsrc) != 1) return false; qsrc = qsrc->src[0]; }
    if (!is_f32_vec(qsrc, D * H) || qsrc->data != qp->data) return false;
    if (x2->ne[0] != D || x2->ne[1] != H || ggml_nelements(x2) != D * H || !ggml_is_contiguous(x2)) return false;
    xattn_args & a = grp.a;
    a.q = (const float *) qp->data;
    a.k = (const char *) kx->data; a.v = (const char *) vx->data;
    a.k_nb1 = (int64_t) kx->nb[1]; a.k_nb2 = (int64_t) kx->nb[2]; a.v_nb1 = (int64_t) vx->nb[1]; a.v_nb2 = (int64_t) vx->nb[2];
    a.H = (int) H; a.D = (int) D; a.Tc = (int) Tc;
    a.scale = ggml_get_op_params_f32(sm, 0);
    a.out = (float *) x2->data;
    grp.members = { pos, pos_of(an, pv), pos_of(an, vt), pos_of(an, sm), pos_of(an, kq) };
    for (int m : grp.members) if (m < 0) return false;
    grp.emit_pos = pos;
    return true;
}

// C. left-deep sum of (scaled) embedding rows ending at node `pos`
struct embed_group { embed_sum_args a; std::vector<int> members; };

static bool match_embed_term(const analysis & an, const ggml_tensor * e, embed_src & out, std::vector<int> & members) {
    const ggml_tensor * gr = e, * scale = nullptr;
    if (e->op == GGML_OP_MUL) {
        gr = e->src[0]; scale = e->src[1];
        if (scale->type != GGML_TYPE_F32 || ggml_nelements(scale) != 1 || uses_of(an, gr) != 1) return false;
        members.push_back(pos_of(an, e));
    } else if (e->op == GGML_OP_CONT && e->src[0]->op == GGML_OP_PERMUTE && e->src[0]->src[0]->op == GGML_OP_GET_ROWS) {
        // a single gathered row viewed as [1, K] (moshi_vq_decode, core_vq.h:100-109): same bytes
        const ggml_tensor * pm = e->src[0];
        gr = pm->src[0];
        if (gr->ne[1] != 1 || ggml_nelements(gr) != gr->ne[0] || uses_of(an, pm) != 1 || uses_of(an, gr) != 1) return false;
        members.push_back(pos_of(an, e)); members.push_back(pos_of(an, pm));
    }
    if (gr->op != GGML_OP_GET_ROWS) return false;
    const ggml_tensor * tab = gr->src[0], * idx = gr->src[1];
    if (ggml_nelements(idx) != 1 || idx->type != GGML_TYPE_I32 || !dense_rows(tab)) return false;
    switch (tab->type) { case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_Q4_0: case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: break; default: return false; }
    const int32_t * ip = (const int32_t *) idx->data;
    if (idx->op == GGML_OP_CONT && uses_of(an, idx) == 1 && idx->src[0]->type == GGML_TYPE_I32 && idx->src[0]->data && pos_of(an, idx) >= 0) {
        ip = (const int32_t *) idx->src[0]->data;   // a one-element copy: read the source
        members.push_back(pos_of(an, idx));
    }
    out = { (const char *) tab->data, (int64_t) tab->nb[1], tab->ne[1], (int) tab->type, ip, scale ? (const float *) scale->data : nullptr };
    members.push_back(pos_of(an, gr));
    return true;
}

static bool match_embed_sum(const analysis & an, int pos, embed_group & grp) {
    const ggml_tensor * top = an.g->nodes[pos];
    if (top->op != GGML_OP_ADD || top->view_src || top->type != GGML_TYPE_F32 || !ggml_is_contiguous(top)) return false;
    if (!(top->ne[1] == 1 || top->ne[0] == 1) || top->ne[2] != 1 || top->ne[3] != 1) return false;
    // walk down the left spine of the chain of adds: adds[d] = adds[d + 1] + terms[d]
    std::vector<const ggml_tensor *> terms, adds;
    const ggml_tensor * cur = top;
    while (cur->op == GGML_OP_ADD && cur->view_src == NULL) {
        const ggml_tensor * l = cur->src[0], * r = cur->src[1];
        if (cur != top && uses_of(an, cur) != 1) break;
        if (uses_of(an, r) != 1 || !ggml_are_same_shape(l, r) || !ggml_are_same_shape(cur, l)) break;
        if (pos_of(an, cur) < 0 || an.skip[(size_t) pos_of(an, cur)]) break;   // (claimed by another group: it is the base then)
        terms.push_back(r);
        adds.push_back(cur);
        cur = l;
    }
    if (terms.size() < 2) return false;
    // embedding rows from the top down; what is left below them - one more embedding row, or any F32 vector of the same shape that something else
    // computes (tts: the demuxed text embedding's two projections, lm_utils.h:48-66) - is the sum's first term
    std::vector<embed_src> srcs(terms.size());
    std::vector<std::vector<int>> mem(terms.size());
    size_t f = 0;
    while (f < terms.size() && match_embed_term(an, terms[f], srcs[f], mem[f])) f++;
    if (f < 2) return false;
    const ggml_tensor * base = f == terms.size() ? cur : adds[f];   // adds[f] = the partial sum below the first f rows
    embed_src base_src;
    std::vector<int> base_mem;
    bool base_is_row = false;
    if (f == terms.size() && uses_of(an, cur) == 1 && match_embed_term(an, cur, base_src, base_mem)) base_is_row = true;
    if (!base_is_row) {
        if (base->type != GGML_TYPE_F32 || !ggml_is_contiguous(base) || !ggml_are_same_shape(base, top)) return false;
        // row 0 of a one-row F32 "table": any index is clamped to it (embed_sum_kernel), so the first row's index pointer serves
        base_src = { (const char *) base->data, 0, 1, (int) GGML_TYPE_F32, srcs[0].index, nullptr };
    }
    if (f + 1 < 3 || f + 1 > EMBED_SUM_MAX) return false;
    memset(&grp.a, 0, sizeof(grp.a));
    grp.a.n = (int) f + 1;
    grp.a.K = ggml_nelements(top);
    grp.a.out = (float *) top->data;
    grp.a.src[0] = base_src;
    std::vector<int> members = base_mem;
    for (size_t i = 0; i < f; i++) {
        // rows were collected right-to-left (top down); the kernel adds left-to-right
        grp.a.src[1 + i] = srcs[f - 1 - i];
        members.insert(members.end(), mem[f - 1 - i].begin(), mem[f - 1 - i].end());
        members.push_back(pos_of(an, adds[f - 1 - i]));
    }
    grp.members = members;
    return true;
}

// D. streaming / stateless conv1d (conv.h:50-96, 137-161): [elu] -> concat(prev, x) -> {tail cpy, im2col} -> mul_mat -> reshape
//    [-> + bias] [-> residual + y]  becomes  stream_im2col + conv_tail + one product with the bias/residual epilogue.
struct step_group { std::vector<step_fn> steps; int emit_pos; std::vector<int> members; sample_args smp; vq_level_args vq; bool has_vq = false; };   // smp / vq: match_sampler's / match_vq_level's arguments

static bool is_elu(const ggml_tensor * t) { return t->op == GGML_OP_UNARY && t->op_params[0] == GGML_UNARY_OP_ELU && t->view_src == NULL; }

static bool match_conv(const analysis & an, int pos, step_group & grp, emitter & em) {
    const ggml_tensor * mm = an.g->nodes[pos];
    if (mm->op != GGML_OP_MUL_MAT) return false;
    const ggml_tensor * ra = mm->src[0], * rw = mm->src[1];
    if (ra->op != GGML_OP_RESHAPE || rw->op != GGML_OP_RESHAPE) return false;
    const ggml_tensor * im = ra->src[0], * w = rw->src[0];
    if (im->op != GGML_OP_IM2COL || im->src[0] != w || im->type != GGML_TYPE_F16 || w->type != GGML_TYPE_F16 || !ggml_is_contiguous(w)) return false;
    if (uses_of(an, im) != 1 || uses_of(an, ra) != 1 || uses_of(an, mm) != 1) return false;
    const int s0 = im->op_params[0], p0 = im->op_params[2], d0 = im->op_params[4];
    if (p0 != 0 || d0 != 1 || im->ne[2] != 1 || im->ne[3] != 1) return false;
    const ggml_tensor * cat = im->src[1];
    const int Kw = (int) w->ne[0], Cin = (int) w->ne[1];
    std::vector<int> members = { pos_of(an, im), pos_of(an, ra), pos_of(an, rw), pos };
    const ggml_tensor * prev = nullptr, * xin = cat;
    int TP = 0;
    if (cat->op == GGML_OP_CONCAT && cat->op_params[0] == 0 && cat->src[0]->op == GGML_OP_NONE) {
        prev = cat->src[0]; xin = cat->src[1];
        TP = (int) prev->ne[0];
        if (prev->type != GGML_TYPE_F32 || !ggml_is_contiguous(prev) || prev->ne[1] != Cin || ggml_nelements(prev) != (int64_t) TP * Cin || !prev->data) return false;
        if (TP > 32 || uses_of(an, cat) != 2) return false;
        // the tail update: cpy(view(cat, last TP), prev)
        const ggml_tensor * tv = nullptr, * tc = nullptr;
        for (int i = pos_of(an, cat) + 1; i < pos; i++) {
            const ggml_tensor * n = an.g->nodes[i];
            if (n->op == GGML_OP_VIEW && n->src[0] == cat) tv = n;
            if (n->op == GGML_OP_CPY && tv && n->src[0] == tv && n->src[1] == prev) tc = n;
        }
        if (!tv || !tc || uses_of(an, tv) != 1 || uses_of(an, tc) != 0) return false;
        if (tv->ne[0] != TP || tv->ne[1] != Cin || (const char *) tv->data - (const char *) cat->data != (int64_t) (cat->ne[0] - TP) * 4) return false;
        members.push_back(pos_of(an, cat)); members.push_back(pos_of(an, tv)); members.push_back(pos_of(an, tc));
    }
    if (xin->type != GGML_TYPE_F32 || xin->ne[1] != Cin || xin->ne[2] != 1 || xin->ne[3] != 1) return false;
    int pre_elu = 0;
    if (is_elu(xin) && uses_of(an, xin) == 1 && pos_of(an, xin) >= 0) { pre_elu = 1; members.push_back(pos_of(an, xin)); xin = xin->src[0]; }
    if (!xin->data) return false;
    // epilogue
    const ggml_tensor * out = sole_consumer(an, mm);
    if (!out || out->op != GGML_OP_RESHAPE || out->ne[2] != 1) return false;
    members.push_back(pos_of(an, out));
    const int64_t OL = out->ne[0], Cout = out->ne[1];
    mm_epilogue epi;
    memset(&epi, 0, sizeof(epi));
    const ggml_tensor * nx = sole_consumer(an, out);
    if (nx && nx->op == GGML_OP_ADD && !nx->view_src && nx->src[0] == out && nx->src[1]->op == GGML_OP_NONE && nx->src[1]->type == GGML_TYPE_F32 && nx->src[1]->ne[0] == 1 &&
        nx->src[1]->ne[1] == Cout && ggml_nelements(nx->src[1]) == Cout && ggml_is_contiguous(nx->src[1])) {
        epi.bias = (const float *) nx->src[1]->data;
        out = nx; members.push_back(pos_of(an, out));
        nx = sole_consumer(an, out);
    }
    if (nx && nx->op == GGML_OP_ADD && !nx->view_src && nx->src[1] == out && nx->src[0] != out && nx->src[0]->type == GGML_TYPE_F32 &&
        ggml_are_same_shape(nx->src[0], out) && nx->src[0]->data) {
        epi.residual = (const char *) nx->src[0]->data; epi.res_nb0 = (int64_t) nx->src[0]->nb[0]; epi.res_nb1 = (int64_t) nx->src[0]->nb[1];
        out = nx; members.push_back(pos_of(an, out));
    }
    if (!ggml_is_contiguous(out) || !out->data || !im->data) return false;
    // cont(transpose(out)) behind a few-position conv (the encoder's last conv feeding the codec transformer): the product, which stores through its
    // destination's strides at that shape, writes the transposed copy in place of its own output - one launch less, the same values
    const ggml_tensor * tcont = nullptr;
    {
        const ggml_tensor * trn = uses_of(an, out) == 1 ? sole_consumer(an, out) : nullptr;
        const ggml_tensor * ct = trn && trn->op == GGML_OP_TRANSPOSE && uses_of(an, trn) == 1 ? sole_consumer(an, trn) : nullptr;
        if (ct && ct->op == GGML_OP_CONT && ct->type == GGML_TYPE_F32 && ggml_is_contiguous(ct) && ct->data && OL <= 8 && ct->ne[0] == Cout && ct->ne[1] == OL &&
            ggml_nelements(ct) == OL * Cout && pos_of(an, trn) >= 0 && pos_of(an, ct) >= 0 && !(getenv("MI355X_NO_CONV_TRANSPOSE_FOLD"))) {
            tcont = ct;
            members.push_back(pos_of(an, trn)); members.push_back(pos_of(an, ct));
        }
    }
    for (int m : members) if (m < 0) return false;
    // The panel hand-off below is a side effect on the emitter (this group registers a slot for its output and may claim its producer's): it must only happen
    // for a group build_plan will accept, so the clash check it runs on the returned members is made here first.
    for (int m : members) if (an.skip[(size_t) m]) return false;
    tdesc d_im = make_tdesc(ra), d_w = make_tdesc(rw), d_x = make_tdesc(xin), d_out = make_tdesc(out);
    d_out.ne[0] = OL; d_out.ne[1] = Cout; d_out.ne[2] = d_out.ne[3] = 1;
    if (tcont) { d_out.data = (char *) tcont->data; d_out.nb[0] = (int64_t) tcont->nb[1]; d_out.nb[1] = (int64_t) tcont->nb[0]; }
    float * pv = prev ? (float *) prev->data : nullptr;
    grp.steps.clear();
    if (TP > 0) {   // the tail update rides on the product kernel (it must follow the im2col, which reads the old tail)
        epi.tail_prev = pv; epi.tail_TP = TP; epi.tail_pre_elu = pre_elu; epi.tail_L = (int) xin->ne[0]; epi.tail_C = Cin;
        epi.tail_x = (const char *) xin->data; epi.tail_nb0 = (int64_t) xin->nb[0]; epi.tail_nb1 = (int64_t) xin->nb[1];
    }
    // Without an im2col launch: where x is the output of a conv / transposed-conv group planned before this one, THAT launch writes this conv's F16 operand
    // (the values stream_im2col_kernel would store, at the same places of a plan-owned panel): same products in the same order, bit-identical. Flag 64 or
    // MI355X_CONV_SCATTER=0 keep the im2col launch. (The gathering form - the product reading (tail | act(x)) itself - was built and measured slower than the
    // launch it removes: profiles/r05_ab_implicit_im2col_gather.txt.)
    static const int scatter_on = getenv("MI355X_CONV_SCATTER") ? atoi(getenv("MI355X_CONV_SCATTER")) : 1;
    bool fused = false;
    tdesc d_a = d_im;
    if (scatter_on && !(em.c->flags & 64)) {
        auto it = em.scatter_of.find(xin);
        if (it != em.scatter_of.end() && it->second->panel == nullptr && xin->nb[0] == 4 && (int64_t) xin->nb[1] == xin->ne[0] * 4 && d_im.nb[0] == 2 && d_im.nb[1] == d_im.ne[0] * 2 &&
            OL == d_im.ne[1] && (int64_t) Kw * Cin == d_im.ne[0] && (int64_t) (OL - 1) * s0 + Kw <= TP + xin->ne[0]) {
            conv_scatter * sc = it->second;
            sc->panel = (uint16_t *) em.ws((size_t) d_im.ne[0] * d_im.ne[1] * 2 + 256);
            sc->prev = pv; sc->K = (int) d_im.ne[0]; sc->Kw = Kw; sc->s0 = s0; sc->TP = TP; sc->M = (int) OL; sc->C = Cin; sc->elu = pre_elu;
            d_a.data = (char *) sc->panel;
            fused = true;
        }
    }
    if (!fused && scatter_on && !(em.c->flags & 64) && Kw == 1 && s0 == 1 && TP == 0 && k_mul_mat_is_few_rows(d_im, d_w) && d_im.nb[1] == d_im.ne[0] * 2) {
        // a 1-tap conv over a few positions fed by something else than a conv launch (the RVQ projections): the product converts its rows itself
        epi.af_x = (const char *) xin->data; epi.af_nb0 = (int64_t) xin->nb[0]; epi.af_nb1 = (int64_t) xin->nb[1]; epi.af_elu = pre_elu;
        fused = true;
    }
    if (!fused) grp.steps.push_back([=](hipStream_t s) { k_stream_im2col(s, d_im, pv, TP, d_x, Kw, s0, pre_elu); });
    const conv_scatter * slot = em.scatter_slot(out);
    grp.steps.push_back([=](hipStream_t s) { mm_epilogue e = epi; e.sc = *slot; k_mul_mat(s, d_out, d_a, d_w, nullptr, &e); });
    grp.members = members;
    grp.emit_pos = pos_of(an, tcont ? tcont : out);
    return true;
}

// E. streaming conv_transpose_1d (conv.h:240-310): [elu] -> conv_transpose_1d -> add_inplace(view lower, view prev tail) -> view full
//    -> cpy(., prev) [-> + bias] -> view(window) -> cont  becomes the partial products + one finishing kernel.
struct convtr_tail { const ggml_tensor * prev, * out; const float * bias; int PT; std::vector<int> members; };

// the streaming tail shared by the dense and the depthwise transposed conv: y is the freshly computed [OLf, OC] block
static bool match_convtr_tail(const analysis & an, const ggml_tensor * y, convtr_tail & t) {
    if (uses_of(an, y) != 1) return false;
    const int64_t OLf = y->ne[0], OC = y->ne[1];
    const ggml_tensor * lower = sole_consumer(an, y);
    if (!lower || lower->op != GGML_OP_VIEW || lower->data != y->data || lower->ne[1] != OC) return false;
    const int64_t PT = lower->ne[0];
    const ggml_tensor * ai = sole_consumer(an, lower);
    if (!ai || ai->op != GGML_OP_ADD || ai->src[0] != lower || ai->data != lower->data) return false;
    const ggml_tensor * partial = ai->src[1];
    if (partial->op != GGML_OP_VIEW || partial->src[0]->op != GGML_OP_NONE || uses_of(an, partial) != 1) return false;
    const ggml_tensor * prev = partial->src[0];
    if (prev->type != GGML_TYPE_F32 || !ggml_is_contiguous(prev) || prev->ne[0] != OLf || prev->ne[1] != OC || ggml_nelements(prev) != OLf * OC) return false;
    if (partial->ne[0] != PT || partial->ne[1] != OC || (const char *) partial->data - (const char *) prev->data != (OLf - PT) * 4 || partial->nb[1] != prev->nb[1]) return false;
    const ggml_tensor * full = sole_consumer(an, ai);
    if (!full || full->op != GGML_OP_VIEW || full->data != y->data || full->ne[0] != OLf || full->ne[1] != OC) return false;
    const ggml_tensor * cp = sole_consumer(an, full);
    if (!cp || cp->op != GGML_OP_CPY || cp->src[1] != prev) return false;
    const ggml_tensor * cur = sole_consumer(an, cp);
    t.bias = nullptr;
    t.members = { pos_of(an, lower), pos_of(an, partial), pos_of(an, ai), pos_of(an, full), pos_of(an, cp) };
    if (cur && cur->op == GGML_OP_ADD && !cur->view_src && cur->src[0] == cp && cur->src[1]->op == GGML_OP_NONE && cur->src[1]->type == GGML_TYPE_F32 &&
        cur->src[1]->ne[0] == 1 && ggml_nelements(cur->src[1]) == OC && ggml_is_contiguous(cur->src[1])) {
        t.bias = (const float *) cur->src[1]->data;
        t.members.push_back(pos_of(an, cur));
        cur = sole_consumer(an, cur);
    }
    if (!cur || cur->op != GGML_OP_VIEW || cur->ne[0] != OLf - PT || cur->ne[1] != OC || cur->data != cur->src[0]->data) return false;
    const ggml_tensor * out = sole_consumer(an, cur);
    if (!out || out->op != GGML_OP_CONT || !ggml_is_contiguous(out) || !out->data) return false;
    t.members.push_back(pos_of(an, cur)); t.members.push_back(pos_of(an, out));
    t.prev = prev; t.out = out; t.PT = (int) PT;
    return true;
}

static bool match_convtr(const analysis & an, int pos, step_group & grp, emitter & em) {
    const ggml_tensor * ct = an.g->nodes[pos];
    if (ct->op != GGML_OP_CONV_TRANSPOSE_1D) return false;
    const ggml_tensor * w = ct->src[0], * xin = ct->src[1];
    const int K = (int) w->ne[0], OC = (int) w->ne[1], s0 = ct->op_params[0], L = (int) xin->ne[0], PT = K - s0;
    if (PT <= 0 || PT > L * s0 || xin->type != GGML_TYPE_F32 || !(w->type == GGML_TYPE_F16 || w->type == GGML_TYPE_F32)) return false;
    if ((int64_t) w->nb[1] != (int64_t) w->nb[0] * K) return false;
    convtr_tail t;
    if (!match_convtr_tail(an, ct, t) || t.PT != PT) return false;
    std::vector<int> members = t.members;
    members.push_back(pos);
    int pre_elu = 0;
    if (is_elu(xin) && uses_of(an, xin) == 1 && pos_of(an, xin) >= 0) { pre_elu = 1; members.push_back(pos_of(an, xin)); xin = xin->src[0]; }
    if (!xin->data) return false;
    for (int m : members) if (m < 0) return false;
    for (int m : members) if (an.skip[(size_t) m]) return false;   // (as in match_conv: no slot is registered for a group build_plan would drop)
    void * ws = em.ws(k_conv_transpose_1d_ws_size(w, xin));
    const tdesc d_w = make_tdesc(w), d_x = make_tdesc(xin), d_out = make_tdesc(t.out);
    float * pv = (float *) t.prev->data;
    const float * bias = t.bias;
    grp.steps.clear();
    const conv_scatter * slot = ggml_is_contiguous(t.out) && t.out->type == GGML_TYPE_F32 ? em.scatter_slot(t.out) : nullptr;
    grp.steps.push_back([=](hipStream_t s) {
        const int nsplit = k_conv_transpose_1d_partial(s, d_w, d_x, ws, pre_elu);
        k_convtr_finish(s, d_out, pv, bias, ws, K, OC, L, s0, nsplit, slot);
    });
    grp.members = members;
    grp.emit_pos = pos_of(an, t.out);
    return true;
}

// E2. depthwise streaming conv_transpose_1d on a single frame (the 12.5 -> 25 Hz upsampler, conv.h:262-278): one multiply per
//     kernel tap, concatenated, then the same streaming tail
static bool match_dw_convtr(const analysis & an, int pos, step_group & grp) {
    const ggml_tensor * y = an.g->nodes[pos];
    if (y->op != GGML_OP_CONCAT || y->op_params[0] != 0 || y->type != GGML_TYPE_F32) return false;
    std::vector<const ggml_tensor *> pieces;
    std::vector<int> members;
    const ggml_tensor * cur = y;
    while (cur->op == GGML_OP_CONCAT && cur->op_params[0] == 0) {
        if (cur != y && uses_of(an, cur) != 1) return false;
        pieces.insert(pieces.begin(), cur->src[1]);
        members.push_back(pos_of(an, cur));
        cur = cur->src[0];
    }
    pieces.insert(pieces.begin(), cur);
    const int K = (int) pieces.size();
    if (K < 2 || K > 16 || y->ne[0] != K) return false;
    const int64_t C = y->ne[1];
    const ggml_tensor * x = nullptr, * w = nullptr;
    for (int k = 0; k < K; k++) {
        const ggml_tensor * pc = pieces[(size_t) k];
        if (pc->op != GGML_OP_MUL || pc->view_src || uses_of(an, pc) != 1 || pc->ne[0] != 1 || pc->ne[1] != C) return false;
        const ggml_tensor * sub = pc->src[1];
        if (sub->op != GGML_OP_VIEW || sub->type != GGML_TYPE_F32 || sub->ne[0] != 1 || sub->ne[1] != C) return false;
        if (k == 0) { x = pc->src[0]; w = sub->src[0]; }
        if (pc->src[0] != x || sub->src[0] != w || (const char *) sub->data - (const char *) w->data != (int64_t) k * (int64_t) w->nb[0] || sub->nb[1] != w->nb[2]) return false;
        members.push_back(pos_of(an, pc)); members.push_back(pos_of(an, sub));
    }
    if (!x || x->type != GGML_TYPE_F32 || x->ne[0] != 1 || x->ne[1] != C || !x->data || w->ne[0] != K || w->nb[0] != 4) return false;
    convtr_tail t;
    if (!match_convtr_tail(an, y, t) || t.PT >= K) return false;
    for (int m : t.members) members.push_back(m);
    for (int m : members) if (m < 0) return false;
    const char * xp = (const char *) x->data; const int64_t x_cs = (int64_t) x->nb[1];
    const char * wp = (const char *) w->data; const int64_t w_cs = (int64_t) w->nb[2];
    float * pv = (float *) t.prev->data; const float * bias = t.bias; float * out = (float *) t.out->data;
    const int PT = t.PT, Cn = (int) C;
    // cont(transpose(out)) behind it (the decoder's upsampler feeding the codec transformer): written directly, as above
    int64_t out_cs = K - PT, out_ks = 1;
    const ggml_tensor * emit = t.out;
    {
        const ggml_tensor * trn = uses_of(an, t.out) == 1 ? sole_consumer(an, t.out) : nullptr;
        const ggml_tensor * ct = trn && trn->op == GGML_OP_TRANSPOSE && uses_of(an, trn) == 1 ? sole_consumer(an, trn) : nullptr;
        if (ct && ct->op == GGML_OP_CONT && ct->type == GGML_TYPE_F32 && ggml_is_contiguous(ct) && ct->data && ct->ne[0] == C && ct->ne[1] == K - PT && ggml_nelements(ct) == C * (K - PT) &&
            pos_of(an, trn) >= 0 && pos_of(an, ct) >= 0 && !(getenv("MI355X_NO_CONV_TRANSPOSE_FOLD"))) {
            out = (float *) ct->data; out_cs = 1; out_ks = C; emit = ct;
            members.push_back(pos_of(an, trn)); members.push_back(pos_of(an, ct));
        }
    }
    grp.steps.clear();
    grp.steps.push_back([=](hipStream_t s) { k_dw_convtr_frame(s, out, pv, bias, xp, x_cs, wp, w_cs, K, PT, Cn, out_cs, out_ks); });
    grp.members = members;
    grp.emit_pos = pos_of(an, emit);
    return true;
}

// F. one level of the residual-VQ encoder (core_vq.h:27-56 + 171-194): the whole distance / argmax / gather / residual chain
static const ggml_tensor * find_consumer(const analysis & an, const ggml_tensor * t, enum ggml_op op, int nth = 0) {
    for (int i = pos_of(an, t) + 1; i < an.g->n_nodes; i++) {
        const ggml_tensor * n = an.g->nodes[i];
        if (n->op != op) continue;
        bool uses = false;
        for (int s = 0; s < GGML_MAX_SRC; s++) if (n->src[s] == t && n != t) uses = true;
        if (uses && nth-- == 0) return n;
    }
    return nullptr;
}

static bool match_vq_level(const analysis & an, int pos, step_group & grp, emitter & em) {
    const ggml_tensor * am = an.g->nodes[pos];
    if (am->op != GGML_OP_ARGMAX || ggml_nelements(am) != 1) return false;
    auto one_use = [&](const ggml_tensor * t, enum ggml_op op) { return t && t->op == op && !t->view_src && uses_of(an, t) == 1; };
    const ggml_tensor * dv = am->src[0];
    if (!one_use(dv, GGML_OP_DIV)) return false;
    const ggml_tensor * num = dv->src[0], * ad = dv->src[1];
    if (!one_use(ad, GGML_OP_ADD)) return false;
    const ggml_tensor * rs = ad->src[0], * addc = ad->src[1];
    if (rs->op != GGML_OP_RESHAPE || uses_of(an, rs) != 1) return false;
    const ggml_tensor * sr = rs->src[0];
    if (!one_use(sr, GGML_OP_SUM_ROWS)) return false;
    const ggml_tensor * sq = sr->src[0];
    if (!one_use(sq, GGML_OP_MUL) || sq->src[0] != sq->src[1]) return false;
    const ggml_tensor * df = sq->src[0];
    if (!one_use(df, GGML_OP_SUB)) return false;
    const ggml_tensor * rb = df->src[0], * ra2 = df->src[1];
    if (!one_use(rb, GGML_OP_REPEAT) || ra2->op != GGML_OP_RESHAPE || uses_of(an, ra2) != 1) return false;
    const ggml_tensor * emb = rb->src[0], * ra = ra2->src[0];
    if (!one_use(ra, GGML_OP_REPEAT)) return false;
    const ggml_tensor * ra0 = ra->src[0];
    if (ra0->op != GGML_OP_RESHAPE || uses_of(an, ra0) != 1) return false;
    const ggml_tensor * ac = ra0->src[0];
    if (!one_use(ac, GGML_OP_CONT)) return false;
    const ggml_tensor * xp = ac->src[0];
    if (xp->op != GGML_OP_PERMUTE || uses_of(an, xp) != 1) return false;
    const ggml_tensor * resid = xp->src[0];
    const int64_t D = emb->ne[0], NC = emb->ne[1];
    if (emb->op != GGML_OP_NONE || emb->type != GGML_TYPE_F32 || !dense_rows(emb) || D != 256 || NC > 4096) return false;
    if (resid->type != GGML_TYPE_F32 || resid->ne[0] != 1 || resid->ne[1] != D || ggml_nelements(resid) != D || !resid->data) return false;
    if (ggml_nelements(rb) != D * NC || ggml_nelements(ra) != D * NC || ggml_nelements(dv) != NC) return false;
    if (num->op != GGML_OP_NONE || num->type != GGML_TYPE_F32 || ggml_nelements(num) != NC || !ggml_is_contiguous(num)) return false;
    if (addc->op != GGML_OP_NONE || addc->type != GGML_TYPE_F32 || ggml_nelements(addc) != 1) return false;
    std::vector<int> members = { pos, pos_of(an, dv), pos_of(an, ad), pos_of(an, rs), pos_of(an, sr), pos_of(an, sq), pos_of(an, df), pos_of(an, rb),
                                 pos_of(an, ra2), pos_of(an, ra), pos_of(an, ra0), pos_of(an, ac), pos_of(an, xp) };
    // consumers of the code: the F32 cast (always) and the centroid gather feeding the next residual (all but the last level)
    const ggml_tensor * cs = find_consumer(an, am, GGML_OP_CPY);
    if (!cs || cs->type != GGML_TYPE_F32 || cs->src[0] != am || cs->view_src) return false;
    members.push_back(pos_of(an, cs));
    int last = pos_of(an, cs);
    float * resid_out = nullptr;
    const ggml_tensor * ci = find_consumer(an, am, GGML_OP_CONT);
    if (ci) {
        if (uses_of(an, am) != 2 || uses_of(an, ci) != 1) return false;
        const ggml_tensor * gr = sole_consumer(an, ci);
        if (!one_use(gr, GGML_OP_GET_ROWS) || gr->src[0] != emb || gr->src[1] != ci) return false;
        const ggml_tensor * pm = sole_consumer(an, gr);
        if (!pm || pm->op != GGML_OP_PERMUTE || uses_of(an, pm) != 1) return false;
        const ggml_tensor * qc = sole_consumer(an, pm);
        if (!one_use(qc, GGML_OP_CONT)) return false;
        const ggml_tensor * sb = sole_consumer(an, qc);
        if (!sb || sb->op != GGML_OP_SUB || sb->view_src || sb->src[0] != resid || sb->src[1] != qc || !ggml_is_contiguous(sb) || !sb->data) return false;
        if (uses_of(an, resid) != 2) return false;
        resid_out = (float *) sb->data;
        for (const ggml_tensor * t : { ci, gr, pm, qc, sb }) members.push_back(pos_of(an, t));
        if (pos_of(an, sb) > last) last = pos_of(an, sb);
    } else if (uses_of(an, am) != 1) return false;
    for (int m : members) if (m < 0) return false;
    // Emitted where the F32 code is produced: everything the kernel reads precedes the argmax, and the next residual is
    // only consumed after its own (later) position; tensors never share storage (ggml_backend_alloc_ctx_tensors).
    last = pos_of(an, cs);
    if (last < pos) return false;
    char * ws = (char *) em.ws(VQ_LEVEL_WS_BYTES);
    HIP_CHECK(hipMemsetAsync(ws, 0, VQ_LEVEL_WS_BYTES, em.c->stream));   // stream-ordered behind the block's previous user (plans are built outside capture)
    vq_level_args a;
    a.emb = (const char *) emb->data; a.emb_row_bytes = (int64_t) emb->nb[1]; a.D = (int) D; a.NC = (int) NC;
    a.resid = (const char *) resid->data; a.resid_stride = (int64_t) resid->nb[1];
    a.add_c = (const float *) addc->data; a.num = (const float *) num->data;
    a.resid_out = resid_out; a.idx_f = (float *) cs->data; a.idx_i = (int32_t *) am->data;
    a.cand_val = (float *) ws; a.cand_idx = (int32_t *) (ws + 1024); a.counter = (unsigned *) (ws + 2048);
    grp.steps.clear();
    grp.steps.push_back([=](hipStream_t s) { k_vq_level(s, a); });
    grp.members = members;
    grp.emit_pos = last;
    grp.vq = a; grp.has_vq = true;
    return true;
}

// G. a tree of concats / layout nodes over one-element F32 tensors, optionally cast to I32 at the top (the RVQ encoder's code
//    vector: vq.h:32-45, 97-114): one gather instead of a copy per concat
static bool vector_like(const ggml_tensor * t) {
    int big = 0;
    for (int i = 0; i < 4; i++) if (t->ne[i] > 1) big++;
    return big <= 1;
}
static bool collect_scalar_leaves(const analysis & an, const ggml_tensor * t, bool top, std::vector<const ggml_tensor *> & leaves, std::vector<int> & members) {
    if (!vector_like(t) || t->type != GGML_TYPE_F32) return false;
    if (ggml_nelements(t) == 1 && t->op != GGML_OP_CONCAT) {
        const ggml_tensor * base = t;
        while ((base->op == GGML_OP_PERMUTE || base->op == GGML_OP_RESHAPE || base->op == GGML_OP_VIEW) && pos_of(an, base) >= 0) {
            if (uses_of(an, base) != 1) break;
            members.push_back(pos_of(an, base));
            base = base->src[0];
            if (ggml_nelements(base) != 1) return false;
        }
        if (!t->data) return false;
        leaves.push_back(t);
        return true;
    }
    if (!top && uses_of(an, t) != 1) return false;
    if (pos_of(an, t) < 0) return false;
    if (t->op == GGML_OP_CONCAT) {
        if (t->view_src) return false;
        members.push_back(pos_of(an, t));
        return collect_scalar_leaves(an, t->src[0], false, leaves, members) && collect_scalar_leaves(an, t->src[1], false, leaves, members);
    }
    if (t->op == GGML_OP_PERMUTE || t->op == GGML_OP_RESHAPE) {
        members.push_back(pos_of(an, t));
        return collect_scalar_leaves(an, t->src[0], false, leaves, members);
    }
    return false;
}
static bool match_scalar_gather(const analysis & an, int pos, step_group & grp) {
    const ggml_tensor * top = an.g->nodes[pos];
    const ggml_tensor * tree = top;
    std::vector<int> members;
    if (top->op == GGML_OP_CPY && top->view_src == NULL && top->type == GGML_TYPE_I32 && top->src[0]->type == GGML_TYPE_F32 && ggml_is_contiguous(top)) {
        members.push_back(pos);
        tree = top->src[0];
        if (uses_of(an, tree) != 1) return false;
    } else if (top->op != GGML_OP_CONCAT || !ggml_is_contiguous(top)) return false;
    if (!vector_like(top) || !top->data) return false;
    std::vector<const ggml_tensor *> leaves;
    if (!collect_scalar_leaves(an, tree, tree == top, leaves, members)) return false;
    if ((int64_t) leaves.size() != ggml_nelements(top) || leaves.size() < 3 || leaves.size() > GATHER_MAX) return false;
    for (int m : members) if (m < 0) return false;
    gather_args a;
    memset(&a, 0, sizeof(a));
    a.n = (int) leaves.size();
    for (int i = 0; i < a.n; i++) a.src[i] = (const float *) leaves[(size_t) i]->data;
    a.dst = top->data;
    a.dst_type = top->type;
    grp.steps.clear();
    grp.steps.push_back([=](hipStream_t s) { k_gather_scalars(s, a); });
    grp.members = members;
    grp.emit_pos = pos;
    return true;
}

// H. the top-k sampler (moshi_sample_token with temp > 0, sampling.h:4-64), matched at its last node:
//    out = get_rows(cont(permute(idx)), reshape(argmax(div(reshape(permute(get_rows(cont(permute(p)), idx))), noise))))
//    with idx = view(argsort_desc(p), k), p = soft_max(scale(logits, 1 / temp))
static bool match_sampler(const analysis & an, int pos, step_group & grp) {
    const ggml_tensor * out = an.g->nodes[pos];
    if (out->op != GGML_OP_GET_ROWS || out->type != GGML_TYPE_I32 || ggml_nelements(out) != 1) return false;
    auto is = [&](const ggml_tensor * t, enum ggml_op op, int uses) { return t && t->op == op && pos_of(an, t) >= 0 && uses_of(an, t) == uses; };
    const ggml_tensor * cont2 = out->src[0], * nx = out->src[1];
    if (!is(cont2, GGML_OP_CONT, 1) || !is(cont2->src[0], GGML_OP_PERMUTE, 1)) return false;
    const ggml_tensor * irows = cont2->src[0], * idx = irows->src[0];
    if (!is(idx, GGML_OP_VIEW, 2) || !is(idx->src[0], GGML_OP_ARGSORT, 1) || idx->type != GGML_TYPE_I32 || idx->data != idx->src[0]->data) return false;
    const ggml_tensor * srt = idx->src[0], * pr = srt->src[0];
    if (srt->op_params[0] != GGML_SORT_ORDER_DESC || !is(pr, GGML_OP_SOFT_MAX, 2) || pr->src[1] != NULL) return false;
    if (ggml_get_op_params_f32(pr, 0) != 1.0f || ggml_get_op_params_f32(pr, 1) != 0.0f) return false;
    const ggml_tensor * sc = pr->src[0];
    if (!is(sc, GGML_OP_SCALE, 1) || ggml_get_op_params_f32(sc, 1) != 0.0f) return false;
    const ggml_tensor * logits = sc->src[0];
    const int64_t n = pr->ne[0], k = idx->ne[0];
    if (logits->type != GGML_TYPE_F32 || !ggml_is_contiguous(logits) || ggml_nelements(logits) != n || ggml_nelements(pr) != n || !logits->data) return false;
    if (n > SAMPLE_MAX_N || k > SAMPLE_MAX_K || k < 1 || ggml_nelements(idx) != k) return false;
    // the argmax side
    while (nx && (nx->op == GGML_OP_RESHAPE || nx->op == GGML_OP_VIEW) && pos_of(an, nx) >= 0 && uses_of(an, nx) == 1) nx = nx->src[0];
    std::vector<int> members = { pos, pos_of(an, cont2), pos_of(an, irows), pos_of(an, idx), pos_of(an, srt), pos_of(an, pr), pos_of(an, sc) };
    for (const ggml_tensor * t = out->src[1]; t != nx; t = t->src[0]) members.push_back(pos_of(an, t));
    const ggml_tensor * am = nx;
    if (!is(am, GGML_OP_ARGMAX, 1)) return false;
    const ggml_tensor * q = am->src[0];
    if (!is(q, GGML_OP_DIV, 1) || q->view_src) return false;
    const ggml_tensor * noise = q->src[1], * in2 = q->src[0];
    if (noise->type != GGML_TYPE_F32 || ggml_nelements(noise) != k || !ggml_is_contiguous(noise) || !noise->data || pos_of(an, noise) >= 0) return false;
    members.push_back(pos_of(an, am)); members.push_back(pos_of(an, q));
    const ggml_tensor * t = in2;
    while (t && (t->op == GGML_OP_RESHAPE || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_VIEW) && pos_of(an, t) >= 0 && uses_of(an, t) == 1) { members.push_back(pos_of(an, t)); t = t->src[0]; }
    if (!is(t, GGML_OP_GET_ROWS, 1) || t->src[1] != idx) return false;
    members.push_back(pos_of(an, t));
    const ggml_tensor * c1 = t->src[0];
    if (!is(c1, GGML_OP_CONT, 1) || !is(c1->src[0], GGML_OP_PERMUTE, 1) || c1->src[0]->src[0] != pr) return false;
    members.push_back(pos_of(an, c1)); members.push_back(pos_of(an, c1->src[0]));
    for (int m : members) if (m < 0) return false;
    sample_args a;
    a.logits = (const float *) logits->data; a.n = (int) n; a.scale = ggml_get_op_params_f32(sc, 0); a.k = (int) k;
    a.noise = (const float *) noise->data; a.out = (int32_t *) out->data; a.out2 = nullptr;
    // a copy of the token into an I32 slot (the chained Depth graph's token vector, lm.h:527): written by the same launch
    for (int j = pos + 1; j < an.g->n_nodes; j++) {
        const ggml_tensor * cp = an.g->nodes[j];
        if (cp->op == GGML_OP_CPY && cp->src[0] == out && cp->type == GGML_TYPE_I32 && ggml_nelements(cp) == 1 && cp->data && !an.skip[(size_t) j]) {
            a.out2 = (int32_t *) cp->data; members.push_back(j); break;
        }
    }
    grp.steps.clear();
    grp.steps.push_back([=](hipStream_t s) { k_sample_topk(s, a); });
    grp.members = members;
    grp.emit_pos = pos;
    grp.smp = a;
    return true;
}

// ---- plan construction --------------------------------------------------------------------------------------
// Kernels whose workgroups wait for each other inside a launch (persistent chains, the fused attention + out_proj launch) need their WHOLE grid resident.
// Two such launches from two streams of one device could interleave their dispatch and each keep the other's workgroups off the compute units, so only ONE
// context per device plans them: the first that asks (the LM stream; the codec stream's graphs have no such runs today). Released with the context.
static hip_ctx * g_spin_owner[64];
// (asked only when a plan is about to CONTAIN such a launch: a context that never plans one never claims the device)
static bool is_stream_context(const hip_ctx * c) { for (const hip_ctx * o : stream_contexts()) if (o == c) return true; return false; }
static bool spin_kernels_allowed(hip_ctx * c) {
    if (c->device < 0 || c->device >= 64) return false;
    // An ADDITIONAL command stream (ggml_backend_mi355x_init_stream: the codec stream of the two-stream frame loop) never claims the device: its graphs are the
    // first a frame submits (the encode half), and a claim from there would take the persistent launches away from the LM stream they were built for.
    if (is_stream_context(c)) return false;
    if (!g_spin_owner[c->device]) g_spin_owner[c->device] = c;
    return g_spin_owner[c->device] == c;
}
static bool spin_kernels_possible(const hip_ctx * c) { return c->device >= 0 && c->device < 64 && !is_stream_context(c) && (!g_spin_owner[c->device] || g_spin_owner[c->device] == c); }

static plan_t * build_plan(hip_ctx * c, ggml_cgraph * g, bool keep = true) {
    plan_t * p = new plan_t;
    p->n_nodes = g->n_nodes;
    emitter em = { c, p };
    analysis an;
    analyse(an, g);
    collect_plan_buffers(p, g);
    const bool fuse = !(c->flags & 1);

    // fused groups, keyed by the position at which they are emitted
    std::map<int, std::vector<pstep>> at_pos;
    std::vector<attn_group> attn_groups;
    static const bool no_attn_prologue = getenv("MI355X_NO_ATTN_PROLOGUE") != nullptr;
    static const bool no_argmax_epilogue = getenv("MI355X_NO_ARGMAX_EPILOGUE") != nullptr;
    if (fuse) {
        // attention blocks first (they swallow set_rows / soft_max / two mul_mats); emitted after the mat-vec pass, which may
        // absorb a short-ring attention into the projection that consumes it
        for (int i = 0; i < g->n_nodes; i++) {
            if (an.skip[(size_t) i] || g->nodes[i]->op != GGML_OP_SOFT_MAX) continue;
            attn_group grp;
            if (!match_attention(an, i, grp)) continue;
            bool clash = false;
            for (int m : grp.members) if (an.skip[(size_t) m]) clash = true;
            if (clash) continue;
            for (int m : grp.members) an.skip[(size_t) m] = 1;
            attn_groups.push_back(grp);
            p->n_fused += (int) grp.members.size();
        }
        {   // mask copies that every consumer now bypasses
            std::map<const ggml_tensor *, int> bypass;
            for (auto & ag : attn_groups) if (ag.mask_node) bypass[ag.mask_node]++;
            for (auto & kv : bypass) if (uses_of(an, kv.first) == kv.second) { an.skip[(size_t) pos_of(an, kv.first)] = 1; p->n_fused++; }
        }
        // codec convolutions
        for (int i = 0; i < g->n_nodes; i++) {
            if (an.skip[(size_t) i]) continue;
            step_group grp;
            if (g->nodes[i]->op == GGML_OP_MUL_MAT) { if (!match_conv(an, i, grp, em)) continue; }
            else if (g->nodes[i]->op == GGML_OP_CONV_TRANSPOSE_1D) { if (!match_convtr(an, i, grp, em)) continue; }
            else if (g->nodes[i]->op == GGML_OP_ARGMAX) { if (!match_vq_level(an, i, grp, em)) continue; }
            else if (g->nodes[i]->op == GGML_OP_CONCAT) { if (!match_dw_convtr(an, i, grp)) continue; }
            else continue;
            bool clash = false;
            for (int m : grp.members) if (an.skip[(size_t) m]) clash = true;
            if (clash) continue;
            for (int m : grp.members) an.skip[(size_t) m] = 1;
            for (auto & f : grp.steps) {
                at_pos[grp.emit_pos].push_back(f);
                if (grp.has_vq) { p->vq_copies.emplace_back(new vq_level_args(grp.vq)); at_pos[grp.emit_pos].back().vq = p->vq_copies.back().get(); }
            }
            p->n_fused += (int) grp.members.size();
        }
        // scalar gathers (top-most node first, so it claims the whole tree)
        for (int i = g->n_nodes - 1; i >= 0; i--) {
            if (an.skip[(size_t) i] || !(g->nodes[i]->op == GGML_OP_CPY || g->nodes[i]->op == GGML_OP_CONCAT)) continue;
            step_group grp;
            if (!match_scalar_gather(an, i, grp)) continue;
            bool clash = false;
            for (int m : grp.members) if (an.skip[(size_t) m]) clash = true;
            if (clash) continue;
            for (int m : grp.members) an.skip[(size_t) m] = 1;
            for (auto & f : grp.steps) at_pos[grp.emit_pos].push_back(f);
            p->n_fused += (int) grp.members.size();
        }
        // top-k samplers (temp > 0)
        static const bool no_sampler = getenv("MI355X_NO_SAMPLER_FUSION") != nullptr;
        for (int i = g->n_nodes - 1; i >= 0 && !no_sampler; i--) {
            if (an.skip[(size_t) i] || g->nodes[i]->op != GGML_OP_GET_ROWS || g->nodes[i]->type != GGML_TYPE_I32) continue;
            step_group grp;
            if (!match_sampler(an, i, grp)) continue;
            bool clash = false;
            for (int m : grp.members) if (an.skip[(size_t) m]) clash = true;
            if (clash) continue;
            for (int m : grp.members) an.skip[(size_t) m] = 1;
            // (a step of its own that a step program may take in: the Depth transformer's samplers sit between its steps' mat-vecs, mv_args::special = 3)
            p->sampler_copies.emplace_back(new sample_args(grp.smp));
            for (auto & f : grp.steps) at_pos[grp.emit_pos].push_back(pstep::special_step(f, 3, nullptr, nullptr, p->sampler_copies.back().get()));
            p->n_fused += (int) grp.members.size();
        }
        // cross-attention over cached conditions (tts)
        static const bool no_xattn = getenv("MI355X_NO_CROSS_ATTN_FUSION") != nullptr;
        for (int i = 0; i < g->n_nodes && !no_xattn; i++) {
            if (an.skip[(size_t) i] || g->nodes[i]->op != GGML_OP_CONT) continue;
            xattn_group xg;
            if (!match_cross_attention(an, i, xg)) continue;
            bool clash = false;
            for (int m : xg.members) if (an.skip[(size_t) m]) clash = true;
            if (clash) continue;
            for (int m : xg.members) an.skip[(size_t) m] = 1;
            const xattn_args xa = xg.a;
            at_pos[xg.emit_pos].push_back([=](hipStream_t s) { k_cross_attn(s, xa); });
            p->n_fused += (int) xg.members.size();
        }
        // one embedding row through a small Q8_0 projection (tts: low-rank Depth embeddings): get_rows -> mul_mat [-> cast] as one launch
        static const bool no_lowrank = getenv("MI355X_NO_LOWRANK_FUSION") != nullptr;
        for (int i = 0; i < g->n_nodes && !no_lowrank; i++) {
            const ggml_tensor * mm = g->nodes[i];
            if (an.skip[(size_t) i] || mm->op != GGML_OP_MUL_MAT || mm->type != GGML_TYPE_F32) continue;
            const ggml_tensor * W = mm->src[0], * gr = mm->src[1];
            if (W->type != GGML_TYPE_Q8_0 || !dense_rows(W) || W->ne[2] * W->ne[3] != 1 || gr->op != GGML_OP_GET_ROWS || gr->type != GGML_TYPE_F32) continue;
            const int pg = pos_of(an, gr);
            if (pg < 0 || an.skip[(size_t) pg] || uses_of(an, gr) != 1 || ggml_nelements(gr) != gr->ne[0] || gr->ne[0] != W->ne[0]) continue;
            const ggml_tensor * tab = gr->src[0], * idx = gr->src[1];
            if (ggml_nelements(idx) != 1 || idx->type != GGML_TYPE_I32 || !dense_rows(tab) || !idx->data) continue;
            switch (tab->type) { case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_Q4_0: case GGML_TYPE_Q8_0: break; default: continue; }
            const int64_t K = W->ne[0], M = W->ne[1];
            if (K % 32 != 0 || K > 2048 || ggml_nelements(mm) != M || !ggml_is_contiguous(mm)) continue;
            std::vector<int> members = { pg, i };
            float * out = (float *) mm->data;
            int emit = i;
            const ggml_tensor * cp = sole_consumer(an, mm);   // ggml_cast(.., F32) of an F32 tensor: a copy - write its storage directly
            if (cp && cp->op == GGML_OP_CPY && uses_of(an, mm) == 1 && cp->type == GGML_TYPE_F32 && cp->src[0] == mm && ggml_is_contiguous(cp) && cp->view_src == NULL &&
                ggml_are_same_shape(cp, mm) && pos_of(an, cp) >= 0 && !an.skip[(size_t) pos_of(an, cp)]) {
                members.push_back(pos_of(an, cp)); out = (float *) cp->data; emit = pos_of(an, cp);
            }
            for (int m2 : members) an.skip[(size_t) m2] = 1;
            const lowrank_embed_args la = { (const char *) tab->data, (int64_t) tab->nb[1], tab->ne[1], (int) tab->type, (const int32_t *) idx->data,
                                            (const char *) W->data, (int64_t) W->nb[1], (int) K, (int) M, out };
            p->lowrank_copies.emplace_back(new lowrank_embed_args(la));
            at_pos[emit].push_back(pstep::special_step([=](hipStream_t s) { k_lowrank_embed(s, la); }, 2, nullptr, p->lowrank_copies.back().get()));
            p->n_fused += (int) members.size();
        }
        // embedding sums
        for (int i = g->n_nodes - 1; i >= 0; i--) {
            if (an.skip[(size_t) i] || g->nodes[i]->op != GGML_OP_ADD) continue;
            embed_group grp;
            if (!match_embed_sum(an, i, grp)) continue;
            bool clash = false;
            for (int m : grp.members) if (m < 0 || an.skip[(size_t) m]) clash = true;
            if (clash) continue;
            for (int m : grp.members) an.skip[(size_t) m] = 1;
            const embed_sum_args a = grp.a;
            at_pos[i].push_back([=](hipStream_t s) { k_embed_sum(s, a); });
            p->n_fused += (int) grp.members.size();
        }
        // cpy(cont(x), dst): copy the strided source straight into dst (the per-step mask row: cont(view of the bias table) -> cpy into the graph input,
        // transformer.h:1259-1289 - two launches on the LM stream in front of every Temporal graph)
        for (int i = 0; i < g->n_nodes; i++) {
            const ggml_tensor * n = g->nodes[i];
            if (an.skip[(size_t) i] || n->op != GGML_OP_CPY) continue;
            const ggml_tensor * ct = n->src[0];
            const int pc = pos_of(an, ct);
            if (ct->op != GGML_OP_CONT || pc < 0 || an.skip[(size_t) pc] || uses_of(an, ct) != 1 || ct->view_src) continue;
            if (ct->type != GGML_TYPE_F32 || n->type != GGML_TYPE_F32 || ct->src[0]->type != GGML_TYPE_F32 || !ggml_are_same_shape(ct, ct->src[0]) ||
                !ggml_are_same_shape(n, ct) || !ggml_is_contiguous(n)) continue;
            // the strided source is read at the cpy's position instead of the cont's: nothing in between may write through an alias (set_rows, a cpy into
            // a view of the same table), or the fold would copy newer data than the cont saw
            bool writer_between = false;
            for (int j = pc + 1; j < i; j++) if (writes_through_alias(g->nodes[j])) writer_between = true;
            if (writer_between) continue;
            an.skip[(size_t) i] = an.skip[(size_t) pc] = 1;
            const tdesc d = make_tdesc(n), x = make_tdesc(ct->src[0]);
            at_pos[i].push_back([=](hipStream_t s) { k_cpy(s, d, x); });
            p->n_fused += 2;
        }
        // timestep table of add(a, b): one launch (the Temporal graph's RoPE phase: add(arange, offset) -> timestep_embedding)
        for (int i = 0; i < g->n_nodes; i++) {
            const ggml_tensor * n = g->nodes[i];
            if (an.skip[(size_t) i] || n->op != GGML_OP_TIMESTEP_EMBEDDING) continue;
            const ggml_tensor * ad = n->src[0];
            const int pa = pos_of(an, ad);
            if (ad->op != GGML_OP_ADD || pa < 0 || an.skip[(size_t) pa] || uses_of(an, ad) != 1 || ad->view_src || ad->type != GGML_TYPE_F32) continue;
            const ggml_tensor * x = ad->src[0], * y = ad->src[1];
            if (x->type != GGML_TYPE_F32 || y->type != GGML_TYPE_F32 || !ggml_is_contiguous(x) || !ggml_is_contiguous(y) || !ggml_are_same_shape(x, ad) ||
                ggml_nelements(ad) != ad->ne[0] || !(ggml_nelements(y) == ggml_nelements(ad) || ggml_nelements(y) == 1)) continue;
            an.skip[(size_t) i] = an.skip[(size_t) pa] = 1;
            const tdesc d = make_tdesc(n), tx = make_tdesc(x);
            const float * yp = (const float *) y->data; const int yn = (int) ggml_nelements(y);
            const int dim = n->op_params[0], mp = n->op_params[1];
            {
                pstep st([=](hipStream_t s) { k_timestep_embedding(s, d, tx, dim, mp, yp, yn); });
                if (x->op == GGML_OP_NONE && y->op == GGML_OP_NONE && !x->view_src && !y->view_src) { st.hoist_dst = (const char *) n->data; st.hoist_bytes = ggml_nbytes(n); }   // (both operands are uploaded leaves)
                at_pos[i].push_back(std::move(st));
            }
            p->n_fused += 2;
        }
        // mat-vecs with prologue / epilogue
        std::unordered_map<const void *, float *> paired_gate;   // storage of a linear_in output h (never materialised) -> g = silu(h_l) * h_r
        size_t paired_consumed = 0;
        // on by default (MI355X_PAIRED_GATE=0 turns it off): the gate kernel it removes (-2.3 us per layer) is mostly paid back by linear_out quantising its 44
        // activation blocks in every workgroup (+1.8 us, profiles/r02_frame_stamps_paired_gate.txt) - neutral while the frame waited for the host between graphs,
        // +1.1 % once the LM graphs run back to back (354.7 -> 358.7 frames/s, profiles/r02_ab_paired_gate_run_ahead.txt)
        static const bool no_pair = getenv("MI355X_PAIRED_GATE") != nullptr && atoi(getenv("MI355X_PAIRED_GATE")) == 0;
        static const bool no_batched_fusion = getenv("MI355X_NO_BATCHED_MM") != nullptr || getenv("MI355X_NO_BATCHED_FUSION") != nullptr;
        for (int i = 0; i < g->n_nodes; i++) {
            if (an.skip[(size_t) i] || g->nodes[i]->op != GGML_OP_MUL_MAT) continue;
            mv_group grp;
            if (!match_matvec(an, i, grp)) {
                bmm_group bg;
                if (!no_batched_fusion && match_batched_mm(an, i, em, bg)) {
                    bool bclash = false;
                    for (int m : bg.members) if (m != i && an.skip[(size_t) m]) bclash = true;
                    if (!bclash) {
                        for (int m : bg.members) an.skip[(size_t) m] = 1;
                        at_pos[bg.emit_pos].push_back(bg.run);
                        p->n_fused += (int) bg.members.size();
                    }
                }
                continue;
            }
            bool clash = false;
            for (int m : grp.members) if (m < 0 || (m != i && an.skip[(size_t) m])) clash = true;
            if (clash) continue;
            for (int m : grp.members) an.skip[(size_t) m] = 1;
            mv_args a = grp.a;
            if (is_qblock((enum ggml_type) a.wtype) && a.ncols == 1 && !no_argmax_epilogue) {
                // greedy sampling (sampling.h:57-63): argmax of the logits, optionally copied into the token buffer
                const ggml_tensor * ynode = g->nodes[grp.emit_pos];
                const ggml_tensor * am = nullptr;
                int n_am = 0;
                for (int j = grp.emit_pos + 1; j < g->n_nodes; j++) if (g->nodes[j]->op == GGML_OP_ARGMAX && g->nodes[j]->src[0] == ynode) { am = g->nodes[j]; n_am++; }
                if (am && n_am == 1 && !an.skip[(size_t) pos_of(an, am)] && (float *) ynode->data == a.y && ggml_nelements(ynode) == a.M && am->data) {
                    a.argmax_out[0] = (int32_t *) am->data;
                    an.skip[(size_t) pos_of(an, am)] = 1;
                    // a copy of the token into an I32 slot
                    const ggml_tensor * cp = nullptr;
                    for (int j = pos_of(an, am) + 1; j < g->n_nodes; j++)
                        if (g->nodes[j]->op == GGML_OP_CPY && g->nodes[j]->src[0] == am && g->nodes[j]->type == GGML_TYPE_I32 && ggml_nelements(g->nodes[j]) == 1) { cp = g->nodes[j]; break; }
                    if (cp && !an.skip[(size_t) pos_of(an, cp)]) { a.argmax_out[1] = (int32_t *) cp->data; an.skip[(size_t) pos_of(an, cp)] = 1; }
                    unsigned * tk = (unsigned *) em.ws(256 + 2 * 4 * 8192);   // arrival counter | per-workgroup argmax candidates (value, index)
                    HIP_CHECK(hipMemsetAsync(tk, 0, 256, em.c->stream));
                    a.ticket = tk;
                    p->n_fused += 1 + (cp ? 1 : 0);
                }
            }
            if (a.prologue == MV_PLAIN && is_qblock((enum ggml_type) a.wtype) && a.ncols == 1 && !no_attn_prologue) {
                // x is the output of a single-token attention over a short ring (Depth transformer): recompute it in every
                // workgroup of this projection instead of launching it on its own (16 heads x 8 slots is ~nothing)
                for (auto & ag : attn_groups) {
                    const attn_args & at = ag.a;
                    if (ag.emit_pos < 0 || (const float *) at.out != a.x || at.T != 1 || at.D != 64 || at.C > 8 || (int64_t) at.H * at.D != a.K) continue;
                    if (a.K != 1024 || at.H != 16 || ag.emit_pos > grp.emit_pos || uses_of(an, g->nodes[ag.emit_pos]) != 1) continue;
                    p->attn_copies.emplace_back(new attn_args(at));   // owned by the plan, passed by value at launch
                    a.prologue = MV_ATTN;
                    a.attn = p->attn_copies.back().get();
                    ag.emit_pos = -1;
                    break;
                }
            }
            if (a.prologue == MV_RMSNORM && a.wtype == GGML_TYPE_Q4_K && a.ncols == 1 && keep && !(c->flags & 512) && grp.emit_pos == i) {
                // the Temporal layer's in_proj, read by nothing but the layer's single-token attention over a LONG ring: the attention runs as the TAIL of
                // the projection's own launch (inproj_attn_kernel) - emitted where the attention stood; no launch is emitted for the projection itself
                bool merged = false;
                for (auto & ag : attn_groups) {
                    const attn_args & at = ag.a;
                    if (ag.emit_pos < 0 || at.q != a.y || ag.emit_pos < grp.emit_pos) continue;
                    if (!k_inproj_attn_supported(a, at, c->usable_cus)) continue;
                    // every reader of the projection's output must be inside the attention block
                    // (followed through layout-only nodes - a view / reshape / permute / transpose of the output is the output: a reader of such an alias that is
                    // not part of the attention block would run before the deferred projection has written anything)
                    bool private_y = true;
                    std::vector<const ggml_tensor *> aliases { g->nodes[i] };
                    for (int j = i + 1; j < g->n_nodes && private_y; j++) {
                        const ggml_tensor * nj = g->nodes[j];
                        bool reads = false;
                        for (int sidx = 0; sidx < GGML_MAX_SRC; sidx++)
                            if (nj->src[sidx] && std::find(aliases.begin(), aliases.end(), nj->src[sidx]) != aliases.end()) reads = true;
                        if (!reads) continue;
                        const bool layout = nj->op == GGML_OP_VIEW || nj->op == GGML_OP_RESHAPE || nj->op == GGML_OP_PERMUTE || nj->op == GGML_OP_TRANSPOSE;
                        if (layout) { aliases.push_back(nj); continue; }
                        if (std::find(ag.members.begin(), ag.members.end(), j) == ag.members.end()) private_y = false;
                    }
                    if (!private_y || !spin_kernels_allowed(c)) continue;
                    const mv_args am = a; const attn_args aa = at;
                    unsigned * err = c->err_dev;
                    const size_t wn = k_inproj_attn_ws_size(a, at);
                    void * ws = em.ws(wn);
                    HIP_CHECK(hipMemsetAsync(ws, 0, wn, c->stream));
                    at_pos[ag.emit_pos].insert(at_pos[ag.emit_pos].begin(), [=](hipStream_t s) { k_inproj_attn(s, am, aa, ws, err); });
                    ag.emit_pos = -1;
                    p->n_fused += 1; p->n_attn_folded++; c->stats.attention_folds_planned++;
                    merged = true;
                    break;
                }
                if (merged) { if (grp.members.size() > 1) p->n_fused += (int) grp.members.size(); continue; }
            }
            if (!no_pair && a.prologue != MV_GATE_SILU && a.ncols == 1 && a.residual == nullptr && a.res_embed.table == nullptr && a.ticket == nullptr &&
                a.M % 2 == 0 && k_matvec_pair_ok(a.wtype, a.K, a.M / 2) && grp.emit_pos == i) {
                // linear_in of a gated FFN whose only readers are the silu(left) * right pair feeding a long linear_out: let every workgroup take
                // matching rows of both halves and write g itself (the [2 F] intermediate and the gate kernel disappear)
                const ggml_tensor * h = g->nodes[i];
                const int64_t F = a.M / 2;
                const ggml_tensor * l = nullptr, * r = nullptr;
                int nv = 0;
                for (int j = i + 1; j < g->n_nodes && j < i + 12; j++) {
                    const ggml_tensor * v = g->nodes[j];
                    if (v->op == GGML_OP_VIEW && v->src[0] == h) { nv++; if (v->data == h->data) l = v; else if ((const char *) v->data == (const char *) h->data + F * 4) r = v; }
                }
                if (uses_of(an, h) == 2 && nv == 2 && l && r && ggml_nelements(l) == F && ggml_nelements(r) == F && ggml_is_contiguous(l) && ggml_is_contiguous(r) &&
                    uses_of(an, l) == 1 && uses_of(an, r) == 1) {
                    const ggml_tensor * sl = sole_consumer(an, l), * ml = sole_consumer(an, r);
                    if (sl && sl->op == GGML_OP_UNARY && sl->op_params[0] == GGML_UNARY_OP_SILU && uses_of(an, sl) == 1 && ml && ml->op == GGML_OP_MUL &&
                        ml->src[0] == sl && ml->src[1] == r && uses_of(an, ml) == 1) {
                        const ggml_tensor * mo = sole_consumer(an, ml);
                        // the pairing is decided HERE, for both sides: the consumer's own match is run now and must come out as the gated prologue over this
                        // very h with none of its members claimed by another group - otherwise h has to be written and nothing is redirected
                        bool consumer_ok = false;
                        if (mo && mo->op == GGML_OP_MUL_MAT && mo->src[1] == ml && is_qblock(mo->src[0]->type) && mo->src[0]->ne[0] == F && F > 4096 &&
                            !an.skip[(size_t) pos_of(an, mo)]) {
                            mv_group cg;
                            const int mpos = pos_of(an, mo);
                            consumer_ok = match_matvec(an, mpos, cg) && cg.a.prologue == MV_GATE_SILU && (const void *) cg.a.x == h->data && cg.a.K == F && cg.a.ncols == 1;
                            for (int m2 : cg.members) if (m2 < 0 || (m2 != mpos && an.skip[(size_t) m2])) consumer_ok = false;
                        }
                        if (consumer_ok) {
                            float * gbuf = (float *) em.ws((size_t) F * 4);
                            a.pair_F = F;
                            a.y = gbuf;
                            paired_gate[h->data] = gbuf;
                        }
                    }
                }
            }
            if (a.prologue == MV_GATE_SILU && is_qblock((enum ggml_type) a.wtype) && a.K > 4096 && paired_gate.count((const void *) a.x)) {
                // the producer already wrote g = silu(left) * right: plain activation, quantised in this kernel's prologue
                const float * gbuf = paired_gate[(const void *) a.x];
                paired_consumed++;
                a.prologue = MV_PLAIN;
                a.x = gbuf;
                a.x_cs = a.K;
            }
            if (a.prologue == MV_GATE_SILU && is_qblock((enum ggml_type) a.wtype) && a.K > 4096) {
                // long gated rows: quantise the activation once, not once per workgroup
                void * blocks = em.ws((size_t) (a.K / 256) * MV_XBLK_BYTES);
                const float * h = a.x; const int64_t K = a.K;
                const int wt = a.wtype;
                at_pos[grp.emit_pos].push_back([=](hipStream_t s) { k_gate_quant_q8k(s, h, K, blocks, wt); });
                a.prologue = MV_PREQ8K;
                a.x = (const float *) blocks;
            }
            at_pos[grp.emit_pos].push_back(pstep(a));
            if (grp.members.size() > 1) p->n_fused += (int) grp.members.size();
        }
        GGML_ASSERT(paired_consumed == paired_gate.size() && "a linear_in was redirected to the paired gate form but its linear_out did not pick the gate vector up");
    }
    if (fuse) {
        // LayerNorm with weight (and bias) over one row that no mat-vec prologue took (tts: norm_cross in front of the cross-attention's row-view projection):
        // ggml_norm -> ggml_mul(., w) -> [ggml_add(., b)] as one launch with the three launches' float operations
        for (int i = 0; i < g->n_nodes; i++) {
            const ggml_tensor * nn = g->nodes[i];
            if (an.skip[(size_t) i] || (nn->op != GGML_OP_NORM && nn->op != GGML_OP_RMS_NORM) || nn->type != GGML_TYPE_F32 || ggml_nelements(nn) != nn->ne[0] || uses_of(an, nn) != 1) continue;
            if (nn->src[0]->type != GGML_TYPE_F32 || nn->src[0]->nb[0] != 4) continue;
            const ggml_tensor * ml = sole_consumer(an, nn);
            if (!ml || ml->op != GGML_OP_MUL || ml->view_src || pos_of(an, ml) < 0 || an.skip[(size_t) pos_of(an, ml)]) continue;
            const ggml_tensor * wv = ml->src[0] == nn ? ml->src[1] : ml->src[0];
            if (wv == nn || wv->type != GGML_TYPE_F32 || !ggml_is_contiguous(wv) || ggml_nelements(wv) != nn->ne[0] || !wv->data || !ggml_is_contiguous(ml)) continue;
            std::vector<int> members = { i, pos_of(an, ml) };
            const ggml_tensor * last = ml; const float * bp = nullptr;
            const ggml_tensor * ad = uses_of(an, ml) == 1 ? sole_consumer(an, ml) : nullptr;
            if (ad && ad->op == GGML_OP_ADD && !ad->view_src && pos_of(an, ad) >= 0 && !an.skip[(size_t) pos_of(an, ad)] && ggml_is_contiguous(ad)) {
                const ggml_tensor * bv = ad->src[0] == ml ? ad->src[1] : ad->src[0];
                if (bv != ml && bv->type == GGML_TYPE_F32 && ggml_is_contiguous(bv) && ggml_nelements(bv) == nn->ne[0] && bv->data && bv->op == GGML_OP_NONE && ad->src[0] == ml) {
                    bp = (const float *) bv->data; last = ad; members.push_back(pos_of(an, ad));
                }
            }
            // (the weight must be the second factor or the first: the product is the same float either way; the bias must be the second summand: v + b)
            for (int m2 : members) an.skip[(size_t) m2] = 1;
            const tdesc a = make_tdesc(nn->src[0]); const float eps = ggml_get_op_params_f32(nn, 0); const int rms = nn->op == GGML_OP_RMS_NORM;
            const float * wp = (const float *) wv->data; float * out = (float *) last->data;
            at_pos[pos_of(an, last)].push_back([=](hipStream_t s) { k_norm_affine(s, a, eps, rms, wp, bp, out); });
            p->n_fused += (int) members.size();
        }
    }
    for (auto & ag : attn_groups) {
        if (ag.emit_pos < 0) continue;
        const attn_args a = ag.a;
        unsigned * err = c->err_dev;
        if (a.T > 4) {
            // a block of T > 4 new rows (batched prompt prefill): one workgroup per head and group of 4 rows; first every row's K / V
            // goes into the ring, then all groups attend at once (causality is in the mask rows)
            at_pos[ag.emit_pos].insert(at_pos[ag.emit_pos].begin(), [=](hipStream_t s) {
                attn_args b = a;
                b.n_groups = (a.T + 3) / 4;
                b.write_only = 1; k_attn_decode(s, b, nullptr, err);
                b.write_only = 0; k_attn_decode(s, b, nullptr, err);
            });
            continue;
        }
        void * ws = nullptr;
        if (const size_t n = k_attn_split_resident(a, c->usable_cus) ? k_attn_decode_ws_size(a) : 0) { ws = em.ws(n); HIP_CHECK(hipMemsetAsync(ws, 0, n, c->stream)); }   // sequence numbers start at zero
        if (a.T <= 4 && a.n_groups <= 1 && !ws) {
            p->attn_copies.emplace_back(new attn_args(a));
            at_pos[ag.emit_pos].insert(at_pos[ag.emit_pos].begin(), pstep::special_step([=](hipStream_t s) { k_attn_decode(s, a, ws, err); }, 1, p->attn_copies.back().get(), nullptr));
        } else
        at_pos[ag.emit_pos].insert(at_pos[ag.emit_pos].begin(), [=](hipStream_t s) { k_attn_decode(s, a, ws, err); });
    }
    const bool dump = getenv("MI355X_DUMP_PLAN") != nullptr;
    for (int i = 0; i < g->n_nodes; i++) {
        if (dump) {
            const ggml_tensor * n = g->nodes[i];
            const bool layout = n->op == GGML_OP_VIEW || n->op == GGML_OP_RESHAPE || n->op == GGML_OP_PERMUTE || n->op == GGML_OP_TRANSPOSE;
            auto itd = at_pos.find(i);
            if (!layout || itd != at_pos.end())
                fprintf(stderr, "plan %4d %-18s [%5lld %5lld %4lld %2lld] %s%s%s\n", i, ggml_op_name(n->op), (long long) n->ne[0], (long long) n->ne[1], (long long) n->ne[2],
                        (long long) n->ne[3], an.skip[(size_t) i] ? "fused" : (layout ? "-" : "GENERIC"), itd != at_pos.end() ? " <emit group>" : "", n->view_src ? " (alias)" : "");
        }
        if (!an.skip[(size_t) i]) emit_generic(em, g->nodes[i]);
        auto it = at_pos.find(i);
        if (it != at_pos.end()) for (auto & f : it->second) p->steps.push_back(f);
    }
    // Persistent chains: a run of consecutive block mat-vecs each of which waits on its predecessor (the chained Depth transformer,
    // lm.h:446-553: 26 mat-vecs per step, 8 / 16 steps per graph) becomes ONE launch of the chain engine (hip_chain.hip).
    // (only for plans that are kept - cached graphs, profile mode: a one-off plan is launched once and freed at once, a chain's tables would be built,
    // uploaded and torn down per compute)
    if (fuse && keep && !(c->flags & 16) && ((c->flags & 32) || k_chain_default_on()) && spin_kernels_possible(c)) {
        // The codec transformers' RoPE table sits between layer 0's in_proj and its attention in node order: moved in front of that in_proj (whose operands it
        // does not touch), the whole stack is one run (hip_chain_mimi.h takes it from in_proj 0; otherwise the program starts at layer 0's attention).
        for (size_t i = 1; i + 1 < p->steps.size(); i++) {
            pstep & h = p->steps[i];
            const pstep & m = p->steps[i - 1];
            if (!h.hoist_dst || !m.is_mv || m.mv.special || m.mv.wtype != GGML_TYPE_F32 || m.mv.ncols != 2 || !p->steps[i + 1].is_mv) continue;
            auto clash = [&](const float * q, int64_t n_floats) { return q && (const char *) q < h.hoist_dst + h.hoist_bytes && h.hoist_dst < (const char *) (q + n_floats); };
            const mv_args & a = m.mv;
            if (clash(a.x, a.x_cs * (a.ncols - 1) + a.K) || clash(a.y, a.y_cs * (a.ncols - 1) + a.M) || clash(a.x_out, a.K * a.ncols) || clash(a.residual, a.r_cs * (a.ncols - 1) + a.M) ||
                clash(a.alpha, a.K) || clash(a.beta, a.K)) continue;
            std::swap(p->steps[i - 1], p->steps[i]);
        }
        std::vector<pstep> merged;
        size_t i = 0;
        while (i < p->steps.size()) {
            if (p->steps[i].vq) {
                // consecutive levels of one RVQ encode stack (vq.h:97-114): one persistent launch with a candidate hand-off per level (hip_chain.hip, vq_chain_kernel)
                size_t e = i;
                std::vector<vq_level_args> lv;
                while (e < p->steps.size() && p->steps[e].vq && lv.size() < 32 && (lv.empty() || ((const void *) p->steps[e].vq->resid == (const void *) lv.back().resid_out && p->steps[e].vq->resid_stride == 4))) {
                    lv.push_back(*p->steps[e].vq); e++;
                }
                if (lv.size() >= 2 && k_vq_chain_accept(lv.data(), (int) lv.size(), c->usable_cus) && spin_kernels_allowed(c)) {
                    void * ws = em.ws(k_vq_chain_ws_size((int) lv.size()));
                    vq_chain_plan * vc = k_vq_chain_create(c->stream, lv.data(), (int) lv.size(), ws, c->err_dev);
                    p->vq_chains.push_back(vc);
                    p->n_vq_chained += (int) lv.size();
                    merged.push_back(pstep([vc](hipStream_t s) { k_vq_chain_launch(s, vc); }));
                    if (dump) fprintf(stderr, "plan: %zu consecutive RVQ levels -> one launch\n", lv.size());
                    i = e;
                    continue;
                }
                merged.push_back(std::move(p->steps[i])); i++; continue;
            }
            if (!p->steps[i].is_mv) { merged.push_back(std::move(p->steps[i])); i++; continue; }
            size_t e = i;
            while (e < p->steps.size() && p->steps[e].is_mv) e++;
            std::vector<mv_args> run;
            for (size_t k = i; k < e; k++) run.push_back(p->steps[k].mv);
            size_t k = 0;
            while (k < run.size()) {
                int len = k_chain_accept(run.data() + k, (int) (run.size() - k), c->usable_cus, !(c->flags & 1024));
                if (len > 0 && !spin_kernels_allowed(c)) len = 0;
                if (len <= 0) {
                    merged.push_back(std::move(p->steps[i + k])); k++; continue;
                }
                void * ws = em.ws(k_chain_ws_size(run.data() + k, len, c->usable_cus, !(c->flags & 1024)));
                chain_plan * ch = k_chain_create(c->stream, run.data() + k, len, ws, c->err_dev, c->usable_cus, !(c->flags & 1024));
                p->chains.push_back(ch);
                if (k_chain_is_step_program(ch)) p->n_step_programs++;
                merged.push_back(pstep([ch](hipStream_t s) { k_chain_launch(s, ch); }));
                merged.back().chain = ch;
                p->n_chained += len;
                if (dump) fprintf(stderr, "plan: %d consecutive mat-vecs -> one chain launch (%.1f MB of weights)\n", len, (double) k_chain_weight_bytes(ch) / 1e6);
                k += (size_t) len;
            }
            i = e;
        }
        p->steps.swap(merged);
    }
    return p;
}

static void run_steps(hip_ctx * c, plan_t * p) { for (auto & st : p->steps) st.fn(c->stream); }

// flag 8: launch eagerly with start/stop events on every matvec_q4k_kernel dispatch and accumulate
static void run_steps_profiled(hip_ctx * c, plan_t * p) {
    if (!c->prof.recs) {
        c->prof.capacity = 4096;
        c->prof.recs = new mv_profile::rec[(size_t) c->prof.capacity];
        for (int i = 0; i < c->prof.capacity; i++) { HIP_CHECK(hipEventCreate(&c->prof.recs[i].start)); HIP_CHECK(hipEventCreate(&c->prof.recs[i].stop)); }
    }
    c->prof.used = 0;
    k_matvec_set_profile(&c->prof);
    for (auto & st : p->steps) {
        if (!st.chain) { st.fn(c->stream); continue; }
        // a persistent chain launch: one kernel, timed between two stream events
        if (!c->chain_ev[0]) { HIP_CHECK(hipEventCreate(&c->chain_ev[0])); HIP_CHECK(hipEventCreate(&c->chain_ev[1])); }
        HIP_CHECK(hipStreamSynchronize(c->stream));
        HIP_CHECK(hipEventRecord(c->chain_ev[0], c->stream));
        st.fn(c->stream);
        HIP_CHECK(hipEventRecord(c->chain_ev[1], c->stream));
        HIP_CHECK(hipEventSynchronize(c->chain_ev[1]));
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, c->chain_ev[0], c->chain_ev[1]));
        c->prof_chain_seconds += (double) ms * 1e-3; c->prof_chain_launches++; c->prof_chain_bytes += k_chain_weight_bytes(st.chain); c->prof_chain_phases += k_chain_length(st.chain);
    }
    k_matvec_set_profile(nullptr);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < c->prof.used; i++) {
        float ms = 0;
        HIP_CHECK(hipEventElapsedTime(&ms, c->prof.recs[i].start, c->prof.recs[i].stop));
        const int v = c->prof.recs[i].variant == 2 ? 2 : c->prof.recs[i].variant ? 1 : 0;
        if (v != 2) {   // the totals are matvec_q4k_kernel's own (what a profiler lists under that name); inproj_attn_kernel is reported as its own variant
            c->prof_seconds += (double) ms * 1e-3;
            c->prof_launches++;
            c->prof_bytes += c->prof.recs[i].bytes;
        }
        c->prof_seconds_v[v] += (double) ms * 1e-3; c->prof_launches_v[v]++; c->prof_bytes_v[v] += c->prof.recs[i].bytes;
    }
}

static enum ggml_status hip_graph_compute(ggml_backend_t backend, struct ggml_cgraph * g) {
    hip_ctx * c = (hip_ctx *) backend->context;
    ctx_init_lazy(c);
    set_device(c);
    flush_uploads(c);
    c->stats.graphs_computed++;
    if (g->n_nodes == 0) return GGML_STATUS_SUCCESS;

    if (c->flags & 8) {
        plan_t * p = build_plan(c, g);
        run_steps_profiled(c, p);
        plan_free(c, p);
        return GGML_STATUS_SUCCESS;
    }
    const bool cacheable = g->n_nodes >= 32;
    if (!cacheable) {
        plan_t * p = build_plan(c, g, false);
        run_steps(c, p);
        c->stats.kernels_in_last_plan = (int64_t) p->steps.size();
        c->stats.nodes_in_last_plan = p->n_nodes;
        c->stats.fused_nodes_in_last_plan = p->n_fused;
        c->stats.chained_matvecs_in_last_plan = p->n_chained; c->stats.chain_step_programs_in_last_plan = p->n_step_programs; c->stats.vq_levels_chained_in_last_plan = p->n_vq_chained;
        plan_free(c, p);   // workspaces return to the pool; reuse is stream-ordered
        return GGML_STATUS_SUCCESS;
    }

    const uint64_t h = graph_hash(g);
    plan_t * p = nullptr;
    auto it = c->plans.find(g);
    if (it != c->plans.end()) {
        if (it->second->hash == h) { p = it->second; if (p->orphan_seq) { p->orphan_seq = 0; collect_plan_buffers(p, g); } }
        else { HIP_CHECK(hipStreamSynchronize(c->stream)); plan_free(c, it->second); c->plans.erase(it); }
    }
    static const bool time_plan = getenv("MI355X_TIME_PLAN") != nullptr;
    const int64_t tp0 = time_plan ? ggml_time_us() : 0;
    bool fresh = false;
    if (!p) {
        p = build_plan(c, g);
        p->hash = h;
        c->plans[g] = p;
        fresh = true;
    }
    const int64_t tp1 = time_plan ? ggml_time_us() : 0;
    if (!fresh && !p->exec && !(c->flags & 2) && !c->no_capture) {
        // second compute of the same graph: it is a cached graph (the per-frame ones), so capture the launch sequence now; replays cost one
        // hipGraphLaunch. One-shot graphs (scratch contexts, prompt prefill chunks) never pay for a capture.
        HIP_CHECK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        run_steps(c, p);
        HIP_CHECK(hipStreamEndCapture(c->stream, &p->graph));
        HIP_CHECK(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    }
    const int64_t tp2 = time_plan ? ggml_time_us() : 0;
    if (p->exec) { HIP_CHECK(hipGraphLaunch(p->exec, c->stream)); c->stats.graph_replays++; }
    else run_steps(c, p);
    if (time_plan && g->n_nodes > 500)
        fprintf(stderr, "graph %p nodes %d: plan %s %.2f ms, capture %.2f ms, launch %.2f ms (%zu kernels)\n", (void *) g, g->n_nodes, fresh ? "built" : "reused",
                (tp1 - tp0) / 1e3, (tp2 - tp1) / 1e3, (ggml_time_us() - tp2) / 1e3, p->steps.size());
    c->stats.kernels_in_last_plan = (int64_t) p->steps.size();
    c->stats.nodes_in_last_plan = p->n_nodes;
    c->stats.fused_nodes_in_last_plan = p->n_fused;
    c->stats.chained_matvecs_in_last_plan = p->n_chained; c->stats.chain_step_programs_in_last_plan = p->n_step_programs; c->stats.vq_levels_chained_in_last_plan = p->n_vq_chained;
    return GGML_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------------
// backend / device / registry objects
// ---------------------------------------------------------------------------------------------------
static const char * hip_backend_name(ggml_backend_t b) { return ((hip_ctx *) b->context)->name.c_str(); }
static void hip_backend_free(ggml_backend_t b) {
    hip_ctx * c = (hip_ctx *) b->context;
    if (c->stream) {
        set_device(c);
        (void) hipStreamSynchronize(c->stream);
        for (auto & kv : c->plans) plan_free(c, kv.second);
        c->plans.clear();
    }
    if (c->device >= 0 && c->device < 64 && g_spin_owner[c->device] == c) g_spin_owner[c->device] = nullptr;   // (its plans are gone)
    c->in_use = false;
    delete b;
}
static void hip_backend_sync(ggml_backend_t b) {
    hip_ctx * c = (hip_ctx *) b->context;
    if (!c->stream) return;
    set_device(c);
    flush_uploads(c);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    deliver_async_gets(c, c->aget_seq);
}
// ggml_backend_tensor_get_async: a copy queued behind the submitted work into a pinned slot; larger reads take the blocking path
static void hip_get_tensor_async(ggml_backend_t b, const struct ggml_tensor * t, void * data, size_t offset, size_t size) {
    hip_ctx * c = (hip_ctx *) b->context;
    if (size > AGET_SLOT_BYTES) { ggml_backend_tensor_get(t, data, offset, size); return; }
    ctx_init_lazy(c);
    set_device(c);
    flush_uploads(c);
    if (c->aget_pending.size() >= AGET_SLOTS) { HIP_CHECK(hipStreamSynchronize(c->stream)); deliver_async_gets(c, c->aget_seq); }
    const uint64_t seq = ++c->aget_seq;
    const int slot = (int) (seq % AGET_SLOTS);
    HIP_CHECK(hipMemcpyAsync(c->aget_pinned + (size_t) slot * AGET_SLOT_BYTES, (const char *) t->data + offset, size, hipMemcpyDeviceToHost, c->stream));
    c->aget_pending.push_back({ data, size, slot, seq });
}
struct hip_event_ctx { hipEvent_t ev; hip_ctx * c; uint64_t seq; };
static void hip_event_sync(ggml_backend_event_t e) {
    hip_event_ctx * h = (hip_event_ctx *) e->context;
    set_device(h->c);
    HIP_CHECK(hipEventSynchronize(h->ev));
    deliver_async_gets(h->c, h->seq);
    check_device_error(h->c);
}
static void hip_event_free(ggml_backend_event_t e) { hip_event_ctx * h = (hip_event_ctx *) e->context; if (h) { (void) hipEventDestroy(h->ev); delete h; } e->context = NULL; }
static void hip_event_record(ggml_backend_t b, ggml_backend_event_t e) {
    hip_ctx * c = (hip_ctx *) b->context;
    ctx_init_lazy(c);
    set_device(c);
    hip_event_ctx * h = (hip_event_ctx *) e->context;
    if (!h) {
        h = new hip_event_ctx;
        HIP_CHECK(hipEventCreateWithFlags(&h->ev, hipEventDisableTiming));
        e->context = h; e->synchronize = hip_event_sync; e->free_context = hip_event_free;
    }
    flush_uploads(c);
    h->c = c; h->seq = c->aget_seq;
    HIP_CHECK(hipEventRecord(h->ev, c->stream));
}
static void hip_event_wait(ggml_backend_t b, ggml_backend_event_t e) {
    hip_ctx * c = (hip_ctx *) b->context;
    hip_event_ctx * h = (hip_event_ctx *) e->context;
    if (!h) return;                      // never recorded: nothing to wait for
    ctx_init_lazy(c);
    set_device(c);
    flush_uploads(c);
    HIP_CHECK(hipStreamWaitEvent(c->stream, h->ev, 0));
}
static bool hip_supports_op(ggml_backend_t, const struct ggml_tensor * op) {
    switch (op->op) {
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP:
            if (ggml_is_quantized(op->type) || ggml_is_quantized(op->src[0]->type)) {
                if (op->type == op->src[0]->type) return true;
                // load-time (re)quantisation of float rows (src/loader.h:160-187): F32 / F16 / BF16 -> Q8_0 / Q4_0 / Q4_K
                const enum ggml_type st = op->src[0]->type;
                return (st == GGML_TYPE_F32 || st == GGML_TYPE_F16 || st == GGML_TYPE_BF16) &&
                       (op->type == GGML_TYPE_Q8_0 || op->type == GGML_TYPE_Q4_0 || op->type == GGML_TYPE_Q4_K) && op->src[0]->ne[0] % ggml_blck_size(op->type) == 0;
            }
            return true;
        case GGML_OP_MUL_MAT:
            switch (op->src[0]->type) { case GGML_TYPE_F32: case GGML_TYPE_F16: case GGML_TYPE_BF16: case GGML_TYPE_Q4_0: case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_K: return true; default: return false; }
        default: return op->op < GGML_OP_COUNT;
    }
}

static const char * hip_dev_name(ggml_backend_dev_t d) { return ((hip_ctx *) d->context)->name.c_str(); }
static const char * hip_dev_desc(ggml_backend_dev_t d) { return ((hip_ctx *) d->context)->description.c_str(); }
static void hip_dev_memory(ggml_backend_dev_t d, size_t * free, size_t * total) {
    hip_ctx * c = (hip_ctx *) d->context;
    set_device(c);
    if (hipMemGetInfo(free, total) != hipSuccess) { *free = 0; *total = 0; }
}
static enum ggml_backend_dev_type hip_dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static ggml_backend_t hip_dev_init(ggml_backend_dev_t d, const char *) {
    hip_ctx * c = (hip_ctx *) d->context;
    ctx_init_lazy(c);
    // a fresh backend handle starts from default flags and zeroed counters (the device context is shared)
    if (c->flags != 0) { HIP_CHECK(hipStreamSynchronize(c->stream)); for (auto & kv : c->plans) plan_free(c, kv.second); c->plans.clear(); c->flags = 0; }
    c->stats = {};
    auto * b = new ggml_backend;
    b->iface = { hip_backend_name, hip_backend_free, hip_backend_sync, hip_alloc_buffer, hip_graph_compute, hip_supports_op, hip_get_tensor_async, hip_event_record, hip_event_wait };
    b->device = d;
    b->context = c;
    return b;
}

static const char * hip_reg_name(ggml_backend_reg_t) { return "ROCm"; }
static size_t hip_reg_dev_count(ggml_backend_reg_t) { return contexts().size(); }
static ggml_backend_dev_t hip_reg_get_dev(ggml_backend_reg_t, size_t i) { return &contexts()[i]->dev_obj; }
static void * hip_reg_proc(ggml_backend_reg_t, const char *) { return NULL; }   // no thread count on a GPU

extern "C" ggml_backend_reg_t ggml_backend_mi355x_reg(void) {
    static ggml_backend_reg reg = { { hip_reg_name, hip_reg_dev_count, hip_reg_get_dev, hip_reg_proc }, NULL };
    static bool init = false;
    if (!init) {
        init = true;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
        for (int i = 0; i < n; i++) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, i) != hipSuccess) continue;
            hip_ctx * c = new hip_ctx;
            c->device = i;
            c->name = "ROCm" + std::to_string(i);
            c->description = std::string(prop.name) + " (" + prop.gcnArchName + ")";
            c->dev_obj.iface = { hip_dev_name, hip_dev_desc, hip_dev_memory, hip_dev_type, hip_dev_init };
            c->dev_obj.reg = &reg;
            c->dev_obj.context = c;
            contexts().push_back(c);
        }
    }
    return contexts().empty() ? NULL : &reg;
}

static hip_ctx * ctx_of(ggml_backend_t b) {
    GGML_ASSERT(b && b->iface.get_name == hip_backend_name && "not an MI355X backend");
    return (hip_ctx *) b->context;
}
extern "C" void ggml_backend_mi355x_get_stats(ggml_backend_t b, struct ggml_mi355x_stats * stats) { *stats = ctx_of(b)->stats; }
extern "C" void ggml_backend_mi355x_set_capture(ggml_backend_t b, int enabled) {
    if (b && b->iface.get_name == hip_backend_name) ((hip_ctx *) b->context)->no_capture = !enabled;   // any other backend: nothing to do
}
extern "C" void ggml_backend_mi355x_set_flags(ggml_backend_t b, int flags) {
    hip_ctx * c = ctx_of(b);
    if (c->stream) { flush_uploads(c); HIP_CHECK(hipStreamSynchronize(c->stream)); }
    for (auto & kv : c->plans) plan_free(c, kv.second);
    c->plans.clear();
    c->flags = flags;
}
extern "C" void ggml_backend_mi355x_get_kernel_profile(ggml_backend_t b, struct ggml_mi355x_kernel_profile * out) {
    hip_ctx * c = ctx_of(b);
    out->seconds = c->prof_seconds; out->launches = c->prof_launches; out->bytes = c->prof_bytes;
    for (int v = 0; v < 3; v++) { out->variant_seconds[v] = c->prof_seconds_v[v]; out->variant_launches[v] = c->prof_launches_v[v]; out->variant_bytes[v] = c->prof_bytes_v[v]; }
    out->chain_seconds = c->prof_chain_seconds; out->chain_launches = c->prof_chain_launches; out->chain_bytes = c->prof_chain_bytes; out->chain_phases = c->prof_chain_phases;
}
extern "C" void * ggml_backend_mi355x_get_stream(ggml_backend_t b) { hip_ctx * c = ctx_of(b); ctx_init_lazy(c); return (void *) c->stream; }
// makes the backend's HIP device the calling thread's current one (what a caller that drives another HIP library itself - RCCL's ncclCommInitRank binds the
// communicator to the CURRENT device - needs before it; the backend's own entry points do this internally)
extern "C" void ggml_backend_mi355x_make_current(ggml_backend_t b) { hip_ctx * c = ctx_of(b); ctx_init_lazy(c); set_device(c); }
extern "C" ggml_backend_t ggml_backend_mi355x_init_stream(ggml_backend_t base) {
    if (!base || base->iface.get_name != hip_backend_name) return NULL;   // any other backend: the caller keeps using `base`
    hip_ctx * b0 = (hip_ctx *) base->context;
    hip_ctx * c = nullptr;
    int n_same = 0;
    for (hip_ctx * o : stream_contexts()) if (o->device == b0->device) { n_same++; if (!o->in_use && !c) c = o; }
    if (!c) {
        c = new hip_ctx;
        c->device = b0->device;
        c->name = b0->name + "/s" + std::to_string(n_same + 1);
        c->description = b0->description + ", additional stream";
        c->dev_obj.iface = b0->dev_obj.iface;
        c->dev_obj.reg = b0->dev_obj.reg;
        c->dev_obj.context = c;
        stream_contexts().push_back(c);
    }
    c->in_use = true;
    ggml_backend_t b = c->dev_obj.iface.init_backend(&c->dev_obj, NULL);
    c->flags = b0->flags;             // the debug flags in force on `base` (no fusion / no hipGraph / ...) apply to its sibling stream too
    c->no_capture = b0->no_capture;
    return b;
}
