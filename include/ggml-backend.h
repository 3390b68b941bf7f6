// ggml-backend.h — C-ABI drop-in boundary, part 2 of 4 (devices, buffers, graph submission).
//
// Restates the ggml backend API subset the reference calls. Protocol as the reference
// drives it (src/context.h:520-544, 628-653):
//   buffer = ggml_backend_alloc_ctx_tensors(ctx, backend)   every non-view tensor gets its own storage
//   ggml_backend_tensor_set(t, host, off, n)                 uploads (inputs, constants, weights)
//   ggml_backend_graph_compute(backend, graph)               synchronous: results visible on return
//   ggml_backend_tensor_get(t, host, off, n)                 blocking read-back
// All calls come from one thread.
#pragma once

#include "ggml.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ggml_backend_buffer_type * ggml_backend_buffer_type_t;
typedef struct ggml_backend_buffer      * ggml_backend_buffer_t;
typedef struct ggml_backend             * ggml_backend_t;
typedef struct ggml_backend_reg         * ggml_backend_reg_t;
typedef struct ggml_backend_device      * ggml_backend_dev_t;

enum ggml_backend_dev_type {
    GGML_BACKEND_DEVICE_TYPE_CPU,    // src/moshi.cpp:98-110 asserts a CPU-type backend exists
    GGML_BACKEND_DEVICE_TYPE_GPU,
    GGML_BACKEND_DEVICE_TYPE_ACCEL,
};

struct ggml_backend_dev_caps {
    bool async;
    bool host_buffer;
    bool buffer_from_host_ptr;
    bool events;
};

// tools/common_ggml.h:15-19 reads props.memory_free
struct ggml_backend_dev_props {
    const char * name;
    const char * description;
    size_t memory_free;
    size_t memory_total;
    enum ggml_backend_dev_type type;
    struct ggml_backend_dev_caps caps;
};

// ---- registry / devices (tools/common_ggml.h:21-79, tools/common_utils.h:9-19) ---------------------
GGML_API void   ggml_backend_load_all(void);
GGML_API size_t ggml_backend_reg_count(void);
GGML_API ggml_backend_reg_t ggml_backend_reg_get(size_t index);
GGML_API const char * ggml_backend_reg_name(ggml_backend_reg_t reg);
GGML_API void * ggml_backend_reg_get_proc_address(ggml_backend_reg_t reg, const char * name);

GGML_API size_t             ggml_backend_dev_count(void);
GGML_API ggml_backend_dev_t ggml_backend_dev_get(size_t index);
GGML_API ggml_backend_dev_t ggml_backend_dev_by_name(const char * name);
GGML_API ggml_backend_dev_t ggml_backend_dev_by_type(enum ggml_backend_dev_type type);
GGML_API const char *       ggml_backend_dev_name(ggml_backend_dev_t device);
GGML_API const char *       ggml_backend_dev_description(ggml_backend_dev_t device);
GGML_API enum ggml_backend_dev_type ggml_backend_dev_type(ggml_backend_dev_t device);
GGML_API void               ggml_backend_dev_memory(ggml_backend_dev_t device, size_t * free, size_t * total);
GGML_API void               ggml_backend_dev_get_props(ggml_backend_dev_t device, struct ggml_backend_dev_props * props);
GGML_API ggml_backend_reg_t ggml_backend_dev_backend_reg(ggml_backend_dev_t device);
GGML_API ggml_backend_t     ggml_backend_dev_init(ggml_backend_dev_t device, const char * params);

// returns NULL when no such device exists; the tools print and exit(1) (tools/common_ggml.h:30-34)
GGML_API ggml_backend_t ggml_backend_init_by_name(const char * name, const char * params);
GGML_API ggml_backend_t ggml_backend_init_by_type(enum ggml_backend_dev_type type, const char * params);
GGML_API ggml_backend_t ggml_backend_init_best(void);   // first GPU device, else CPU

GGML_API const char *       ggml_backend_name(ggml_backend_t backend);
GGML_API void               ggml_backend_free(ggml_backend_t backend);
GGML_API ggml_backend_dev_t ggml_backend_get_device(ggml_backend_t backend);
GGML_API void               ggml_backend_synchronize(ggml_backend_t backend);

// proc-address "ggml_backend_set_n_threads" (tools/common_ggml.h:35-44)
typedef void (*ggml_backend_set_n_threads_t)(ggml_backend_t backend, int n_threads);

// ---- buffers -------------------------------------------------------------------------------------
// the only allocator the reference uses (src/context.h:522); caller frees the returned buffer
GGML_API ggml_backend_buffer_t ggml_backend_alloc_ctx_tensors(struct ggml_context * ctx, ggml_backend_t backend);
GGML_API void    ggml_backend_buffer_free    (ggml_backend_buffer_t buffer);   // src/context.h:192
GGML_API size_t  ggml_backend_buffer_get_size(ggml_backend_buffer_t buffer);
GGML_API void *  ggml_backend_buffer_get_base(ggml_backend_buffer_t buffer);
GGML_API const char * ggml_backend_buffer_name(ggml_backend_buffer_t buffer);
GGML_API void    ggml_backend_buffer_clear   (ggml_backend_buffer_t buffer, uint8_t value);
GGML_API bool    ggml_backend_buffer_is_host (ggml_backend_buffer_t buffer);

// ---- data movement (src/context.h:308, 643) ------------------------------------------------------
GGML_API void ggml_backend_tensor_set   (      struct ggml_tensor * tensor, const void * data, size_t offset, size_t size);
GGML_API void ggml_backend_tensor_get   (const struct ggml_tensor * tensor,       void * data, size_t offset, size_t size);
// upstream ggml's stream-ordered read-back and events (ggml-backend.h; the reference itself only uses the blocking calls): the copy is queued behind
// the work already submitted to `backend`; `data` is filled once ggml_backend_synchronize(backend) or ggml_backend_event_synchronize of an event
// recorded after it has returned. Used by the run-ahead frame loop (moshi_hot.h) to take a step's tokens without idling the stream.
GGML_API void ggml_backend_tensor_get_async(ggml_backend_t backend, const struct ggml_tensor * tensor, void * data, size_t offset, size_t size);
typedef struct ggml_backend_event * ggml_backend_event_t;
GGML_API ggml_backend_event_t ggml_backend_event_new(ggml_backend_dev_t device);
GGML_API void ggml_backend_event_free(ggml_backend_event_t event);
GGML_API void ggml_backend_event_record(ggml_backend_event_t event, ggml_backend_t backend);
GGML_API void ggml_backend_event_synchronize(ggml_backend_event_t event);
GGML_API void ggml_backend_event_wait(ggml_backend_t backend, ggml_backend_event_t event);   // work submitted to `backend` after this call runs after the event (device-side wait)
GGML_API void ggml_backend_tensor_memset(      struct ggml_tensor * tensor, uint8_t value,     size_t offset, size_t size);
GGML_API void ggml_backend_tensor_copy  (struct ggml_tensor * src, struct ggml_tensor * dst);

// ---- compute (src/context.h:542, 635) ------------------------------------------------------------
GGML_API enum ggml_status ggml_backend_graph_compute(ggml_backend_t backend, struct ggml_cgraph * cgraph);
GGML_API bool ggml_backend_supports_op(ggml_backend_t backend, const struct ggml_tensor * op);

// ---- MI355X backend extras (not part of upstream ggml; optional for callers) ------------------------
// counters for tests / bench: how the last graph was executed
struct ggml_mi355x_stats {
    int64_t graphs_computed;     // ggml_backend_graph_compute calls
    int64_t graph_replays;       // of which were hipGraph replays of a cached plan
    int64_t kernels_in_last_plan;
    int64_t fused_nodes_in_last_plan;
    int64_t nodes_in_last_plan;
    int64_t uploads_batched;     // small tensor_set calls folded into one scatter launch
    int64_t chained_matvecs_in_last_plan;   // block mat-vecs of the last plan that run inside persistent chain launches (hip_chain.hip)
    int64_t attention_folds_planned;   // attention blocks running as the tail of their in_proj launch (inproj_attn_kernel), summed over every plan built so far
    int64_t chain_step_programs_in_last_plan;   // chain launches of the last plan that run as the Depth transformer's compile-time step program (hip_chain_nest.h)
    int64_t vq_levels_chained_in_last_plan;     // RVQ encode levels of the last plan that run inside multi-level launches (vq_chain_kernel)
};
GGML_API void ggml_backend_mi355x_get_stats(ggml_backend_t backend, struct ggml_mi355x_stats * stats);
// accumulated HIP-event timings of the dominant kernel (Q4_K mat-vec), collected while flag 8 is set
struct ggml_mi355x_kernel_profile {
    double  seconds;    // sum of per-dispatch kernel begin -> end times
    int64_t launches;
    int64_t bytes;      // algorithmic bytes: weight bytes each launch streams (rows x row size)
    // the same three by instantiation family of the kernel: [0] LDS-staged tiles (matvec_q4k_kernel<.., WS = 0>: the Temporal matrices and the text head),
    // [1] register streaming (<.., WS = 1>: the Depth transformer's small matrices), [2] inproj_attn_kernel (the Temporal in_proj's tiles with the layer's
    // attention as the launch's tail: bytes = the in_proj weights, seconds = the whole launch)
    double  variant_seconds[3];
    int64_t variant_launches[3];
    int64_t variant_bytes[3];
    // persistent chain launches (matvec_chain_kernel: the chained Depth transformer's mat-vecs as one launch), timed between two stream events
    double  chain_seconds;
    int64_t chain_launches;
    int64_t chain_bytes;      // weight bytes the launch streams
    int64_t chain_phases;     // mat-vecs (phases) inside those launches
};
GGML_API void ggml_backend_mi355x_get_kernel_profile(ggml_backend_t backend, struct ggml_mi355x_kernel_profile * out);
// bit flags, default 0: 1 = disable fusion (one kernel per node), 2 = disable hipGraph capture, 4 = disable upload batching,
// 8 = profile mode (eager launches, per-dispatch HIP events on matvec_q4k_kernel), 16 = no persistent chain launches (one launch per mat-vec),
// 32 = persistent chain launches on even when the MI355X_CHAIN environment variable says 0 (default: on),
// 64 = every codec convolution keeps its im2col launch (default: the launch producing a convolution's input also writes its F16 im2col panel),
// 512 = the Temporal attention stays a launch of its own (default: it runs as the tail of its in_proj launch, inproj_attn_kernel),
// 1024 = chain launches always take the descriptor-driven kernel (default: the Depth transformer's steps run as a compile-time step program)
GGML_API void ggml_backend_mi355x_set_flags(ggml_backend_t backend, int flags);
// hipGraph capture of repeated graphs on / off without touching cached plans (off: a repeated graph still reuses its plan, launched eagerly -
// for sequences of same-shaped one-off graphs such as prompt-prefill chunks, where a capture costs more than it saves). No-op on other backends.
GGML_API void ggml_backend_mi355x_set_capture(ggml_backend_t backend, int enabled);
// HIP stream the backend launches on (void* = hipStream_t) so callers can bracket it with HIP events
GGML_API void * ggml_backend_mi355x_get_stream(ggml_backend_t backend);
// Makes the backend's HIP device current on the calling thread. A caller that talks to another HIP-based library itself on this backend's stream (RCCL:
// ncclCommInitRank binds the communicator to the thread's CURRENT device) calls this first; the backend's own entry points never rely on the current device.
GGML_API void ggml_backend_mi355x_make_current(ggml_backend_t backend);
// A second command stream on the same GPU: a backend handle with its own HIP stream, upload queue and plan cache. Graphs submitted through it run
// concurrently with those of `base` (the codec of neighbouring frames beside the LM step, moshi_hot.h "software-pipelined frame loop"); buffers
// allocated through either handle are ordinary device memory usable by both - ordering between the two streams is the caller's (host round trips
// through ggml_backend_tensor_get / ggml_backend_synchronize). It starts with the flags in force on `base`. Returns NULL when `base` is not an MI355X
// backend. Free with ggml_backend_free.
GGML_API ggml_backend_t ggml_backend_mi355x_init_stream(ggml_backend_t base);

#ifdef __cplusplus
}
#endif
