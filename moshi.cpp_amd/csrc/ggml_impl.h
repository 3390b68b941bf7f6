// ggml_impl.h — internal layouts shared by the core, the backends and the parity oracle.
// Not part of the drop-in boundary: callers only ever see opaque pointers to these.
#pragma once

#include "ggml.h"
#include "ggml-backend.h"

#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

// Computation graph: nodes in execution order (DFS post-order of ggml_build_forward_expand),
// leafs = tensors with op NONE reached while expanding.
struct ggml_cgraph {
    int size;
    int n_nodes;
    int n_leafs;
    struct ggml_tensor ** nodes;
    struct ggml_tensor ** leafs;
    // open-addressing pointer set of already-visited tensors
    size_t visited_size;
    struct ggml_tensor ** visited;
};

// ---- backend object model ---------------------------------------------------------------------------
struct ggml_backend_buffer_i {
    void (*free_buffer)(ggml_backend_buffer_t buffer);
    void (*memset_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor * t, uint8_t value, size_t offset, size_t size);
    void (*set_tensor)(ggml_backend_buffer_t buffer, struct ggml_tensor * t, const void * data, size_t offset, size_t size);
    void (*get_tensor)(ggml_backend_buffer_t buffer, const struct ggml_tensor * t, void * data, size_t offset, size_t size);
    void (*clear)(ggml_backend_buffer_t buffer, uint8_t value);
};

struct ggml_backend_buffer {
    struct ggml_backend_buffer_i iface;
    ggml_backend_dev_t device;
    void * base;       // first byte of the allocation (host pointer or device pointer)
    size_t size;
    void * context;    // backend private
    bool   is_host;
};

struct ggml_backend_i {
    const char * (*get_name)(ggml_backend_t backend);
    void (*free)(ggml_backend_t backend);
    void (*synchronize)(ggml_backend_t backend);
    ggml_backend_buffer_t (*alloc_buffer)(ggml_backend_t backend, size_t size);
    enum ggml_status (*graph_compute)(ggml_backend_t backend, struct ggml_cgraph * cgraph);
    bool (*supports_op)(ggml_backend_t backend, const struct ggml_tensor * op);
    // optional (NULL: blocking fall-backs): stream-ordered read-back completed by synchronize / an event recorded after it
    void (*get_tensor_async)(ggml_backend_t backend, const struct ggml_tensor * t, void * data, size_t offset, size_t size);
    void (*event_record)(ggml_backend_t backend, ggml_backend_event_t event);
    void (*event_wait)(ggml_backend_t backend, ggml_backend_event_t event);
};

struct ggml_backend_event {
    ggml_backend_dev_t device;
    void * context;                                       // backend private, created at the first record
    void (*synchronize)(ggml_backend_event_t event);      // set by the backend that recorded it (NULL: nothing pending)
    void (*free_context)(ggml_backend_event_t event);
};

struct ggml_backend {
    struct ggml_backend_i iface;
    ggml_backend_dev_t device;
    void * context;
};

struct ggml_backend_device_i {
    const char * (*get_name)(ggml_backend_dev_t dev);
    const char * (*get_description)(ggml_backend_dev_t dev);
    void (*get_memory)(ggml_backend_dev_t dev, size_t * free, size_t * total);
    enum ggml_backend_dev_type (*get_type)(ggml_backend_dev_t dev);
    ggml_backend_t (*init_backend)(ggml_backend_dev_t dev, const char * params);
};

struct ggml_backend_device {
    struct ggml_backend_device_i iface;
    ggml_backend_reg_t reg;
    void * context;
};

struct ggml_backend_reg_i {
    const char * (*get_name)(ggml_backend_reg_t reg);
    size_t (*get_device_count)(ggml_backend_reg_t reg);
    ggml_backend_dev_t (*get_device)(ggml_backend_reg_t reg, size_t index);
    void * (*get_proc_address)(ggml_backend_reg_t reg, const char * name);
};

struct ggml_backend_reg {
    struct ggml_backend_reg_i iface;
    void * context;
};

// registries implemented in ggml_backend.cpp; backends call this from their *_reg() accessor
void ggml_backend_register(ggml_backend_reg_t reg);
// implemented by hip_backend.hip; returns NULL when no HIP device is visible
ggml_backend_reg_t ggml_backend_mi355x_reg(void);

// alignment of every tensor inside a backend buffer
#define GGML_TENSOR_ALIGN 256

static inline float ggml_get_op_params_f32(const struct ggml_tensor * t, int i) {
    float v; memcpy(&v, &t->op_params[i], sizeof(float)); return v;
}
static inline void ggml_set_op_params_f32(struct ggml_tensor * t, int i, float v) {
    memcpy(&t->op_params[i], &v, sizeof(float));
}

#ifdef __cplusplus
}
#endif
