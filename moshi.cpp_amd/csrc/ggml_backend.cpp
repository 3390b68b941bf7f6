// ggml_backend.cpp — backend registry, the host ("CPU") device and the backend-generic entry points of
// include/ggml-backend.h. The MI355X device lives in hip_backend.hip and registers itself here.
#include "ggml_impl.h"
#include "ggml-cpu.h"

#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <unistd.h>

#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------------------
// registry
// ---------------------------------------------------------------------------------------------------
static std::vector<ggml_backend_reg_t> & regs() { static std::vector<ggml_backend_reg_t> v; return v; }
static std::vector<ggml_backend_dev_t> & devs() { static std::vector<ggml_backend_dev_t> v; return v; }

extern "C" void ggml_backend_register(ggml_backend_reg_t reg) {
    if (!reg) return;
    for (auto r : regs()) if (r == reg) return;
    regs().push_back(reg);
    for (size_t i = 0; i < reg->iface.get_device_count(reg); i++) devs().push_back(reg->iface.get_device(reg, i));
}

// GPU devices are listed first so that ggml_backend_dev_get(0) / init_best pick the accelerator
extern "C" void ggml_backend_load_all(void) {
    static bool loaded = false;
    if (loaded) return;
    loaded = true;
    ggml_backend_register(ggml_backend_mi355x_reg());
    ggml_backend_register(ggml_backend_cpu_reg());
}

extern "C" size_t ggml_backend_reg_count(void) { ggml_backend_load_all(); return regs().size(); }
extern "C" ggml_backend_reg_t ggml_backend_reg_get(size_t i) { ggml_backend_load_all(); GGML_ASSERT(i < regs().size()); return regs()[i]; }
extern "C" const char * ggml_backend_reg_name(ggml_backend_reg_t reg) { return reg->iface.get_name(reg); }
extern "C" void * ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char * name) {
    if (!reg || !reg->iface.get_proc_address) return NULL;
    return reg->iface.get_proc_address(reg, name);
}

extern "C" size_t ggml_backend_dev_count(void) { ggml_backend_load_all(); return devs().size(); }
extern "C" ggml_backend_dev_t ggml_backend_dev_get(size_t i) { ggml_backend_load_all(); GGML_ASSERT(i < devs().size()); return devs()[i]; }
extern "C" ggml_backend_dev_t ggml_backend_dev_by_name(const char * name) {
    ggml_backend_load_all();
    for (auto d : devs()) if (strcasecmp(d->iface.get_name(d), name) == 0) return d;
    return NULL;
}
extern "C" ggml_backend_dev_t ggml_backend_dev_by_type(enum ggml_backend_dev_type type) {
    ggml_backend_load_all();
    for (auto d : devs()) if (d->iface.get_type(d) == type) return d;
    return NULL;
}
extern "C" const char * ggml_backend_dev_name(ggml_backend_dev_t d) { return d->iface.get_name(d); }
extern "C" const char * ggml_backend_dev_description(ggml_backend_dev_t d) { return d->iface.get_description(d); }
extern "C" enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t d) { return d->iface.get_type(d); }
extern "C" void ggml_backend_dev_memory(ggml_backend_dev_t d, size_t * free, size_t * total) { d->iface.get_memory(d, free, total); }
extern "C" void ggml_backend_dev_get_props(ggml_backend_dev_t d, struct ggml_backend_dev_props * props) {
    memset(props, 0, sizeof(*props));
    props->name = d->iface.get_name(d);
    props->description = d->iface.get_description(d);
    props->type = d->iface.get_type(d);
    d->iface.get_memory(d, &props->memory_free, &props->memory_total);
    props->caps.host_buffer = props->type == GGML_BACKEND_DEVICE_TYPE_CPU;
}
extern "C" ggml_backend_reg_t ggml_backend_dev_backend_reg(ggml_backend_dev_t d) { return d->reg; }
extern "C" ggml_backend_t ggml_backend_dev_init(ggml_backend_dev_t d, const char * params) { return d->iface.init_backend(d, params); }

extern "C" ggml_backend_t ggml_backend_init_by_name(const char * name, const char * params) {
    ggml_backend_dev_t d = ggml_backend_dev_by_name(name);
    return d ? ggml_backend_dev_init(d, params) : NULL;
}
extern "C" ggml_backend_t ggml_backend_init_by_type(enum ggml_backend_dev_type type, const char * params) {
    ggml_backend_dev_t d = ggml_backend_dev_by_type(type);
    return d ? ggml_backend_dev_init(d, params) : NULL;
}
extern "C" ggml_backend_t ggml_backend_init_best(void) {
    ggml_backend_dev_t d = ggml_backend_dev_by_type(GGML_BACKEND_DEVICE_TYPE_GPU);
    if (!d) d = ggml_backend_dev_by_type(GGML_BACKEND_DEVICE_TYPE_CPU);
    return d ? ggml_backend_dev_init(d, NULL) : NULL;
}

extern "C" const char * ggml_backend_name(ggml_backend_t b) { return b->iface.get_name(b); }
extern "C" void ggml_backend_free(ggml_backend_t b) { if (b) b->iface.free(b); }
extern "C" ggml_backend_dev_t ggml_backend_get_device(ggml_backend_t b) { return b->device; }
extern "C" void ggml_backend_synchronize(ggml_backend_t b) { if (b->iface.synchronize) b->iface.synchronize(b); }
extern "C" void ggml_backend_tensor_get_async(ggml_backend_t b, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    if (b->iface.get_tensor_async) b->iface.get_tensor_async(b, tensor, data, offset, size);
    else ggml_backend_tensor_get(tensor, data, offset, size);     // a backend without streams: done on return
}
extern "C" ggml_backend_event_t ggml_backend_event_new(ggml_backend_dev_t device) {
    ggml_backend_event * e = new ggml_backend_event;
    e->device = device; e->context = NULL; e->synchronize = NULL; e->free_context = NULL;
    return e;
}
extern "C" void ggml_backend_event_free(ggml_backend_event_t e) { if (!e) return; if (e->free_context) e->free_context(e); delete e; }
extern "C" void ggml_backend_event_record(ggml_backend_event_t e, ggml_backend_t b) { if (b->iface.event_record) b->iface.event_record(b, e); }
extern "C" void ggml_backend_event_synchronize(ggml_backend_event_t e) { if (e->synchronize) e->synchronize(e); }
extern "C" void ggml_backend_event_wait(ggml_backend_t b, ggml_backend_event_t e) {
    if (b->iface.event_wait) b->iface.event_wait(b, e); else ggml_backend_event_synchronize(e);   // no streams: the host waits instead
}
extern "C" bool ggml_backend_supports_op(ggml_backend_t b, const struct ggml_tensor * op) {
    return b->iface.supports_op ? b->iface.supports_op(b, op) : false;
}
extern "C" enum ggml_status ggml_backend_graph_compute(ggml_backend_t b, struct ggml_cgraph * g) {
    return b->iface.graph_compute(b, g);
}

// ---------------------------------------------------------------------------------------------------
// buffers and data movement
// ---------------------------------------------------------------------------------------------------
extern "C" ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors(struct ggml_context * ctx, ggml_backend_t backend) {
    GGML_ASSERT(ggml_get_no_alloc(ctx));
    // pass 1: size of everything that needs storage (each tensor aligned on its own)
    size_t total = 0;
    for (struct ggml_tensor * t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t)) {
        if (t->data == NULL && t->view_src == NULL) total += GGML_PAD(ggml_nbytes(t), GGML_TENSOR_ALIGN);
    }
    ggml_backend_buffer_t buffer = backend->iface.alloc_buffer(backend, total);
    if (!buffer) return NULL;
    // pass 2: hand out addresses in context order; views resolve against their (already placed) root
    size_t offs = 0;
    for (struct ggml_tensor * t = ggml_get_first_tensor(ctx); t; t = ggml_get_next_tensor(ctx, t)) {
        if (t->data != NULL) continue;
        if (t->view_src == NULL) {
            t->buffer = buffer;
            t->data = (char *) buffer->base + offs;
            offs += GGML_PAD(ggml_nbytes(t), GGML_TENSOR_ALIGN);
        } else if (t->buffer == NULL) {
            GGML_ASSERT(t->view_src->data != NULL && "view of a tensor that has no storage yet");
            t->buffer = t->view_src->buffer;
            t->data = (char *) t->view_src->data + t->view_offs;
        }
    }
    return buffer;
}

extern "C" void ggml_backend_buffer_free(ggml_backend_buffer_t buffer) { if (buffer) buffer->iface.free_buffer(buffer); }
extern "C" size_t ggml_backend_buffer_get_size(ggml_backend_buffer_t buffer) { return buffer->size; }
extern "C" void * ggml_backend_buffer_get_base(ggml_backend_buffer_t buffer) { return buffer->base; }
extern "C" const char * ggml_backend_buffer_name(ggml_backend_buffer_t buffer) { return buffer->device->iface.get_name(buffer->device); }
extern "C" void ggml_backend_buffer_clear(ggml_backend_buffer_t buffer, uint8_t value) { if (buffer->size) buffer->iface.clear(buffer, value); }
extern "C" bool ggml_backend_buffer_is_host(ggml_backend_buffer_t buffer) { return buffer->is_host; }

static ggml_backend_buffer_t tensor_buffer(const struct ggml_tensor * t) {
    return t->view_src ? t->view_src->buffer : t->buffer;
}

extern "C" void ggml_backend_tensor_set(struct ggml_tensor * tensor, const void * data, size_t offset, size_t size) {
    if (size == 0) return;
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    GGML_ASSERT(buf != NULL && "tensor buffer not set");
    GGML_ASSERT(tensor->data != NULL && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor write out of bounds");
    buf->iface.set_tensor(buf, tensor, data, offset, size);
}
extern "C" void ggml_backend_tensor_get(const struct ggml_tensor * tensor, void * data, size_t offset, size_t size) {
    if (size == 0) return;
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    GGML_ASSERT(buf != NULL && "tensor buffer not set");
    GGML_ASSERT(tensor->data != NULL && "tensor not allocated");
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor) && "tensor read out of bounds");
    buf->iface.get_tensor(buf, tensor, data, offset, size);
}
extern "C" void ggml_backend_tensor_memset(struct ggml_tensor * tensor, uint8_t value, size_t offset, size_t size) {
    if (size == 0) return;
    ggml_backend_buffer_t buf = tensor_buffer(tensor);
    GGML_ASSERT(buf != NULL && tensor->data != NULL);
    GGML_ASSERT(offset + size <= ggml_nbytes(tensor));
    buf->iface.memset_tensor(buf, tensor, value, offset, size);
}
// staged through the host, like the reference's own device<->device path (src/context.h:645-650)
extern "C" void ggml_backend_tensor_copy(struct ggml_tensor * src, struct ggml_tensor * dst) {
    const size_t n = ggml_nbytes(src);
    GGML_ASSERT(n == ggml_nbytes(dst));
    std::vector<uint8_t> tmp(n);
    ggml_backend_tensor_get(src, tmp.data(), 0, n);
    ggml_backend_tensor_set(dst, tmp.data(), 0, n);
}

// ---------------------------------------------------------------------------------------------------
// host ("CPU") device: buffers in host memory; graph execution only through an attached executor
// ---------------------------------------------------------------------------------------------------
static ggml_backend_cpu_graph_compute_t g_cpu_compute = NULL;
struct cpu_backend_ctx { int n_threads; };

extern "C" void ggml_backend_cpu_set_graph_compute(ggml_backend_cpu_graph_compute_t fn) { g_cpu_compute = fn; }

static void cpu_buf_free(ggml_backend_buffer_t b) { free(b->base); delete b; }
static void cpu_buf_memset(ggml_backend_buffer_t, struct ggml_tensor * t, uint8_t v, size_t off, size_t n) { memset((char *) t->data + off, v, n); }
static void cpu_buf_set(ggml_backend_buffer_t, struct ggml_tensor * t, const void * d, size_t off, size_t n) { memcpy((char *) t->data + off, d, n); }
static void cpu_buf_get(ggml_backend_buffer_t, const struct ggml_tensor * t, void * d, size_t off, size_t n) { memcpy(d, (const char *) t->data + off, n); }
static void cpu_buf_clear(ggml_backend_buffer_t b, uint8_t v) { memset(b->base, v, b->size); }

static ggml_backend_dev_t cpu_device();

static ggml_backend_buffer_t cpu_alloc_buffer(ggml_backend_t, size_t size) {
    void * p = NULL;
    if (posix_memalign(&p, GGML_TENSOR_ALIGN, size ? size : GGML_TENSOR_ALIGN) != 0) return NULL;
    auto * b = new ggml_backend_buffer;
    b->iface = { cpu_buf_free, cpu_buf_memset, cpu_buf_set, cpu_buf_get, cpu_buf_clear };
    b->device = cpu_device();
    b->base = p;
    b->size = size;
    b->context = NULL;
    b->is_host = true;
    return b;
}

static const char * cpu_backend_name(ggml_backend_t) { return "CPU"; }
static void cpu_backend_free(ggml_backend_t b) { delete (cpu_backend_ctx *) b->context; delete b; }
// Load-time graphs on host tensors. The reference prepares safetensors weights with tiny graphs on its CPU backend before
// copying them to the accelerator (scratch_cpu, src/moshi.cpp:107): a lone ggml_cast for re-quantisation (src/loader.h:180-187),
// row-range views + cast for the per-step projection split (transformer.h:780-848), and clamp / cont(transpose) / div for the
// codebook centroids (core_vq.h:58-85). Exactly that op set - cpy / cont / dup, clamp, div, layout nodes - is what the host
// device executes by itself, in plain scalar loops. Every other op still fails loudly: the decode hot path has no CPU route.
static bool load_time_graph(const struct ggml_cgraph * g) {
    for (int i = 0; i < g->n_nodes; i++) {
        const struct ggml_tensor * n = g->nodes[i];
        switch (n->op) {
            case GGML_OP_NONE: case GGML_OP_VIEW: case GGML_OP_RESHAPE: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: break;
            case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP:
                if (ggml_nelements(n) != ggml_nelements(n->src[0])) return false;
                if (n->type != n->src[0]->type && !(ggml_is_contiguous(n) && ggml_is_contiguous(n->src[0]))) return false;
                if (n->type == n->src[0]->type && n->type != GGML_TYPE_F32 && !(ggml_is_contiguous(n) && ggml_is_contiguous(n->src[0]))) return false;
                break;
            case GGML_OP_CLAMP:
                if (n->type != GGML_TYPE_F32) return false;
                break;
            case GGML_OP_DIV:
                if (n->type != GGML_TYPE_F32 || n->src[0]->type != GGML_TYPE_F32 || n->src[1]->type != GGML_TYPE_F32 || !ggml_are_same_shape(n, n->src[0])) return false;
                break;
            default: return false;
        }
    }
    return g->n_nodes > 0;
}
static inline char * elem_ptr(const struct ggml_tensor * t, int64_t i0, int64_t i1, int64_t i2, int64_t i3) {
    return (char *) t->data + i0 * (int64_t) t->nb[0] + i1 * (int64_t) t->nb[1] + i2 * (int64_t) t->nb[2] + i3 * (int64_t) t->nb[3];
}
static void host_cast(const struct ggml_tensor * src, struct ggml_tensor * dst) {
    if (src->type == dst->type && ggml_is_contiguous(src) && ggml_is_contiguous(dst)) { memcpy(dst->data, src->data, ggml_nbytes(src)); return; }
    if (src->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32) {   // strided copy in logical element order (cont of a transpose, ...)
        int64_t d[4] = { 0, 0, 0, 0 };
        for (int64_t i3 = 0; i3 < src->ne[3]; i3++) for (int64_t i2 = 0; i2 < src->ne[2]; i2++) for (int64_t i1 = 0; i1 < src->ne[1]; i1++) for (int64_t i0 = 0; i0 < src->ne[0]; i0++) {
            *(float *) elem_ptr(dst, d[0], d[1], d[2], d[3]) = *(const float *) elem_ptr(src, i0, i1, i2, i3);
            if (++d[0] == dst->ne[0]) { d[0] = 0; if (++d[1] == dst->ne[1]) { d[1] = 0; if (++d[2] == dst->ne[2]) { d[2] = 0; ++d[3]; } } }
        }
        return;
    }
    const int64_t K = src->ne[0], rows = ggml_nelements(src) / K;
    GGML_ASSERT(dst->ne[0] == K && "a converting copy keeps the row length");
    std::vector<float> row((size_t) K);
    for (int64_t r = 0; r < rows; r++) {
        const char * sp = (const char *) src->data + (size_t) r * ggml_row_size(src->type, K);
        char * dp = (char *) dst->data + (size_t) r * ggml_row_size(dst->type, K);
        if (src->type == GGML_TYPE_F32) ggml_quantize_row(dst->type, (const float *) sp, dp, K);
        else { ggml_dequantize_row(src->type, sp, row.data(), K); ggml_quantize_row(dst->type, row.data(), dp, K); }
    }
}
static void host_load_time_node(struct ggml_tensor * n) {
    switch (n->op) {
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: host_cast(n->src[0], n); break;
        case GGML_OP_CLAMP: {
            const struct ggml_tensor * a = n->src[0];
            float lo, hi;
            memcpy(&lo, (const char *) n->op_params, 4); memcpy(&hi, (const char *) n->op_params + 4, 4);
            for (int64_t i3 = 0; i3 < n->ne[3]; i3++) for (int64_t i2 = 0; i2 < n->ne[2]; i2++) for (int64_t i1 = 0; i1 < n->ne[1]; i1++) for (int64_t i0 = 0; i0 < n->ne[0]; i0++) {
                const float v = *(const float *) elem_ptr(a, i0, i1, i2, i3);
                *(float *) elem_ptr(n, i0, i1, i2, i3) = v < lo ? lo : (v > hi ? hi : v);
            }
        } break;
        case GGML_OP_DIV: {
            const struct ggml_tensor * a = n->src[0], * b = n->src[1];
            for (int64_t i3 = 0; i3 < n->ne[3]; i3++) for (int64_t i2 = 0; i2 < n->ne[2]; i2++) for (int64_t i1 = 0; i1 < n->ne[1]; i1++) for (int64_t i0 = 0; i0 < n->ne[0]; i0++)
                *(float *) elem_ptr(n, i0, i1, i2, i3) = *(const float *) elem_ptr(a, i0, i1, i2, i3) / *(const float *) elem_ptr(b, i0 % b->ne[0], i1 % b->ne[1], i2 % b->ne[2], i3 % b->ne[3]);
        } break;
        default: break;
    }
}
static enum ggml_status cpu_graph_compute(ggml_backend_t b, struct ggml_cgraph * g) {
    if (!g_cpu_compute && load_time_graph(g)) {
        for (int i = 0; i < g->n_nodes; i++) host_load_time_node(g->nodes[i]);
        return GGML_STATUS_SUCCESS;
    }
    if (!g_cpu_compute) {
        fprintf(stderr, "ggml (mi355x build): the CPU device has no graph executor (only load-time cpy/cast/clamp/div graphs run on the host); "
                        "the decode hot path runs on the MI355X device only. Attach one with ggml_backend_cpu_set_graph_compute() (tests use oracle/).\n");
        return GGML_STATUS_FAILED;
    }
    return g_cpu_compute(g, ((cpu_backend_ctx *) b->context)->n_threads);
}
static bool cpu_supports_op(ggml_backend_t, const struct ggml_tensor *) { return g_cpu_compute != NULL; }
extern "C" enum ggml_status ggml_graph_compute_with_ctx(struct ggml_context *, struct ggml_cgraph * g, int n_threads) {
    if (g_cpu_compute) return g_cpu_compute(g, n_threads);
    if (load_time_graph(g)) { for (int i = 0; i < g->n_nodes; i++) host_load_time_node(g->nodes[i]); return GGML_STATUS_SUCCESS; }
    fprintf(stderr, "ggml (mi355x build): ggml_graph_compute_with_ctx needs a host executor (ggml_backend_cpu_set_graph_compute); the decode hot path runs on the MI355X device only.\n");
    return GGML_STATUS_FAILED;
}

static const char * cpu_dev_name(ggml_backend_dev_t) { return "CPU"; }
static const char * cpu_dev_desc(ggml_backend_dev_t) { return "host memory device"; }
static void cpu_dev_memory(ggml_backend_dev_t, size_t * free, size_t * total) {
    const long pages = sysconf(_SC_PHYS_PAGES), avail = sysconf(_SC_AVPHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
    *total = (size_t) pages * (size_t) psz;
    *free  = (size_t) avail * (size_t) psz;
}
static enum ggml_backend_dev_type cpu_dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_CPU; }
static ggml_backend_t cpu_dev_init(ggml_backend_dev_t dev, const char *) {
    auto * b = new ggml_backend;
    b->iface = { cpu_backend_name, cpu_backend_free, NULL, cpu_alloc_buffer, cpu_graph_compute, cpu_supports_op };
    b->device = dev;
    b->context = new cpu_backend_ctx{ 4 };
    return b;
}

extern "C" void ggml_backend_cpu_set_n_threads(ggml_backend_t b, int n_threads) {
    if (b && ggml_backend_is_cpu(b)) ((cpu_backend_ctx *) b->context)->n_threads = n_threads;
}
extern "C" bool ggml_backend_is_cpu(ggml_backend_t b) { return b && b->iface.get_name == cpu_backend_name; }

static const char * cpu_reg_name(ggml_backend_reg_t) { return "CPU"; }
static size_t cpu_reg_dev_count(ggml_backend_reg_t) { return 1; }
static ggml_backend_dev_t cpu_reg_get_dev(ggml_backend_reg_t, size_t) { return cpu_device(); }
static void * cpu_reg_proc(ggml_backend_reg_t, const char * name) {
    if (strcmp(name, "ggml_backend_set_n_threads") == 0) return (void *) ggml_backend_cpu_set_n_threads;
    if (strcmp(name, "ggml_backend_cpu_set_graph_compute") == 0) return (void *) ggml_backend_cpu_set_graph_compute;
    return NULL;
}

extern "C" ggml_backend_reg_t ggml_backend_cpu_reg(void) {
    static ggml_backend_reg reg = { { cpu_reg_name, cpu_reg_dev_count, cpu_reg_get_dev, cpu_reg_proc }, NULL };
    return &reg;
}
static ggml_backend_dev_t cpu_device() {
    static ggml_backend_device dev = { { cpu_dev_name, cpu_dev_desc, cpu_dev_memory, cpu_dev_type, cpu_dev_init }, ggml_backend_cpu_reg(), NULL };
    return &dev;
}
extern "C" ggml_backend_t ggml_backend_cpu_init(void) { return cpu_dev_init(cpu_device(), NULL); }
