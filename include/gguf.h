// gguf.h — C-ABI drop-in boundary, part 4 of 4 (GGUF container read/write).
//
// Subset used by the reference's WeightLoader (src/loader.h:85-99, 227-270): open a file in
// no_alloc mode to get tensor metadata in a ggml_context, query name/offset/size per tensor, and
// write a tensor-only GGUF v3 file from a context's tensors.
#pragma once

#include "ggml.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGUF_MAGIC   "GGUF"
#define GGUF_VERSION 3
#define GGUF_DEFAULT_ALIGNMENT 32

struct gguf_context;

struct gguf_init_params {
    bool no_alloc;                 // src/loader.h:87: true (metadata only)
    struct ggml_context ** ctx;    // receives a context holding one tensor per GGUF tensor
};

GGML_API struct gguf_context * gguf_init_empty(void);                                                   // src/loader.h:228
GGML_API struct gguf_context * gguf_init_from_file(const char * fname, struct gguf_init_params params); // src/loader.h:90
GGML_API void gguf_free(struct gguf_context * ctx);

GGML_API uint32_t gguf_get_version    (const struct gguf_context * ctx);
GGML_API size_t   gguf_get_alignment  (const struct gguf_context * ctx);
GGML_API size_t   gguf_get_data_offset(const struct gguf_context * ctx);                // src/loader.h:244

GGML_API int64_t      gguf_get_n_kv(const struct gguf_context * ctx);
GGML_API int64_t      gguf_find_key(const struct gguf_context * ctx, const char * key);
GGML_API const char * gguf_get_key (const struct gguf_context * ctx, int64_t key_id);
GGML_API const char * gguf_get_val_str(const struct gguf_context * ctx, int64_t key_id);
GGML_API uint32_t     gguf_get_val_u32(const struct gguf_context * ctx, int64_t key_id);
GGML_API void gguf_set_val_u32(struct gguf_context * ctx, const char * key, uint32_t val);
GGML_API void gguf_set_val_str(struct gguf_context * ctx, const char * key, const char * val);

GGML_API int64_t        gguf_get_n_tensors    (const struct gguf_context * ctx);              // src/loader.h:245
GGML_API int64_t        gguf_find_tensor      (const struct gguf_context * ctx, const char * name);
GGML_API const char *   gguf_get_tensor_name  (const struct gguf_context * ctx, int64_t tensor_id); // src/loader.h:247
GGML_API enum ggml_type gguf_get_tensor_type  (const struct gguf_context * ctx, int64_t tensor_id);
GGML_API size_t         gguf_get_tensor_offset(const struct gguf_context * ctx, int64_t tensor_id); // src/loader.h:250
GGML_API size_t         gguf_get_tensor_size  (const struct gguf_context * ctx, int64_t tensor_id); // src/loader.h:251

// records metadata and a pointer to tensor->data (host) for a later write (src/loader.h:231)
GGML_API void gguf_add_tensor(struct gguf_context * ctx, const struct ggml_tensor * tensor);
// tensor data is taken from tensor->data when it is host memory, otherwise read back through
// ggml_backend_tensor_get (src/loader.h:232)
GGML_API bool gguf_write_to_file(const struct gguf_context * ctx, const char * fname, bool only_meta);

#ifdef __cplusplus
}
#endif
