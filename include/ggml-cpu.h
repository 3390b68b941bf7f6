// ggml-cpu.h — C-ABI drop-in boundary, part 3 of 4 (the CPU-type device).
//
// The reference requires a CPU-type backend to exist next to the accelerator (src/moshi.cpp:98-110,
// tools/common_ggml.h:46-66): it owns host-side staging buffers and, for safetensors input only, runs
// the load-time ggml_cast graphs (src/loader.h:180-187). In this library the "CPU" device provides
// host buffers and tensor_set/get; it does NOT ship a CPU graph executor — the decode hot path runs on
// the MI355X device or fails loudly. A CPU executor (the parity oracle under oracle/, test
// infrastructure) can be attached with ggml_backend_cpu_set_graph_compute().
#pragma once

#include "ggml.h"
#include "ggml-backend.h"

#ifdef __cplusplus
extern "C" {
#endif

GGML_API ggml_backend_t ggml_backend_cpu_init(void);
GGML_API bool ggml_backend_is_cpu(ggml_backend_t backend);
GGML_API void ggml_backend_cpu_set_n_threads(ggml_backend_t backend_cpu, int n_threads);
// graph evaluation in host memory without a backend (tensors of a no_alloc = false context): only the replay tool uses it
// (src/replay.h:321, 375). Runs the host device's executor: load-time graphs, or whatever ggml_backend_cpu_set_graph_compute attached.
GGML_API enum ggml_status ggml_graph_compute_with_ctx(struct ggml_context * ctx, struct ggml_cgraph * cgraph, int n_threads);
GGML_API ggml_backend_reg_t ggml_backend_cpu_reg(void);

// Attach / detach an external CPU graph executor (NULL detaches). Without one,
// ggml_backend_graph_compute() on the CPU backend prints an error and returns GGML_STATUS_FAILED.
typedef enum ggml_status (*ggml_backend_cpu_graph_compute_t)(struct ggml_cgraph * cgraph, int n_threads);
GGML_API void ggml_backend_cpu_set_graph_compute(ggml_backend_cpu_graph_compute_t fn);

#ifdef __cplusplus
}
#endif
