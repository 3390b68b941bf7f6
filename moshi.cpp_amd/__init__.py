"""moshi.cpp_amd — ctypes binding of the MI355X ggml drop-in (libggml-mi355x.so) and of the frame-driver harness (libmoshi-hot.so).

The product is the C-ABI library declared in include/{ggml,ggml-backend,ggml-cpu,gguf}.h and
include/moshi_hot.h; this module only loads it and attaches signatures so that tests and bench.py can
drive it the way libmoshi does (src/context.h:520-544 in the reference). There is no Python or CPU
fallback: if the shared library is missing, importing fails loudly.

The directory name contains a dot, so load it with `__graft_entry__.load_package()`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355X_LIB") or os.path.join(_HERE, "libggml-mi355x.so")   # MI355X_LIB: A/B another build (tests/microbench/ab_bench.sh)
HOT_LIB_PATH = os.environ.get("MI355X_HOT_LIB") or os.path.join(os.path.dirname(LIB_PATH), "libmoshi-hot.so")   # harness: include/moshi_hot.h

GGML_MAX_DIMS, GGML_MAX_SRC, GGML_MAX_NAME = 4, 10, 64

# ggml_type ids (include/ggml.h)
F32, F16, Q4_0, Q8_0, Q4_K, Q8_K, I32, I64, BF16 = 0, 1, 2, 8, 12, 15, 26, 27, 30
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q8_0: "q8_0", Q4_K: "q4_K", Q8_K: "q8_K", I32: "i32", I64: "i64", BF16: "bf16"}
DEV_CPU, DEV_GPU = 0, 1


class Tensor(C.Structure):
    pass


Tensor._fields_ = [
    ("type", C.c_int),
    ("buffer", C.c_void_p),
    ("ne", C.c_int64 * GGML_MAX_DIMS),
    ("nb", C.c_size_t * GGML_MAX_DIMS),
    ("op", C.c_int),
    ("op_params", C.c_int32 * 16),
    ("flags", C.c_int32),
    ("src", C.POINTER(Tensor) * GGML_MAX_SRC),
    ("view_src", C.POINTER(Tensor)),
    ("view_offs", C.c_size_t),
    ("data", C.c_void_p),
    ("name", C.c_char * GGML_MAX_NAME),
    ("extra", C.c_void_p),
    ("padding", C.c_char * 8),
]
TP = C.POINTER(Tensor)


class InitParams(C.Structure):
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


class DevProps(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p), ("memory_free", C.c_size_t), ("memory_total", C.c_size_t),
                ("type", C.c_int), ("caps", C.c_bool * 4)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("graphs_computed", "graph_replays", "kernels_in_last_plan",
                                         "fused_nodes_in_last_plan", "nodes_in_last_plan", "uploads_batched", "chained_matvecs_in_last_plan",
                                         "attention_folds_planned", "chain_step_programs_in_last_plan", "vq_levels_chained_in_last_plan")]


class KernelProfile(C.Structure):
    _fields_ = [("seconds", C.c_double), ("launches", C.c_int64), ("bytes", C.c_int64),
                ("variant_seconds", C.c_double * 3), ("variant_launches", C.c_int64 * 3), ("variant_bytes", C.c_int64 * 3),
                ("chain_seconds", C.c_double), ("chain_launches", C.c_int64), ("chain_bytes", C.c_int64), ("chain_phases", C.c_int64)]


class GGUFInitParams(C.Structure):
    _fields_ = [("no_alloc", C.c_bool), ("ctx", C.POINTER(C.c_void_p))]


P, I, L, Z, F, B, S = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_bool, C.c_char_p

# name -> (restype, argtypes); every symbol include/*.h declares is listed here and checked at load time
SIGNATURES = {
    # ggml.h: misc
    "ggml_time_ms": (L, []), "ggml_time_us": (L, []), "ggml_time_init": (None, []),
    "ggml_type_name": (S, [I]), "ggml_op_name": (S, [I]), "ggml_op_desc": (S, [TP]), "ggml_status_to_string": (S, [I]),
    "ggml_blck_size": (L, [I]), "ggml_type_size": (Z, [I]), "ggml_row_size": (Z, [I, L]), "ggml_is_quantized": (B, [I]),
    "ggml_nelements": (L, [TP]), "ggml_nrows": (L, [TP]), "ggml_nbytes": (Z, [TP]), "ggml_element_size": (Z, [TP]),
    "ggml_n_dims": (I, [TP]), "ggml_is_contiguous": (B, [TP]), "ggml_is_transposed": (B, [TP]), "ggml_is_permuted": (B, [TP]),
    "ggml_are_same_shape": (B, [TP, TP]),
    "ggml_fp16_to_fp32": (F, [C.c_uint16]), "ggml_fp32_to_fp16": (C.c_uint16, [F]),
    "ggml_bf16_to_fp32": (F, [C.c_uint16]), "ggml_fp32_to_bf16": (C.c_uint16, [F]),
    # contexts
    "ggml_init": (P, [InitParams]), "ggml_reset": (None, [P]), "ggml_free": (None, [P]), "ggml_used_mem": (Z, [P]),
    "ggml_get_no_alloc": (B, [P]), "ggml_set_no_alloc": (None, [P, B]),
    "ggml_tensor_overhead": (Z, []), "ggml_graph_overhead": (Z, []), "ggml_graph_overhead_custom": (Z, [Z, B]),
    "ggml_new_tensor": (TP, [P, I, I, C.POINTER(L)]), "ggml_new_tensor_1d": (TP, [P, I, L]), "ggml_new_tensor_2d": (TP, [P, I, L, L]),
    "ggml_new_tensor_3d": (TP, [P, I, L, L, L]), "ggml_new_tensor_4d": (TP, [P, I, L, L, L, L]),
    "ggml_dup_tensor": (TP, [P, TP]), "ggml_view_tensor": (TP, [P, TP]),
    "ggml_get_first_tensor": (TP, [P]), "ggml_get_next_tensor": (TP, [P, TP]), "ggml_get_tensor": (TP, [P, S]),
    "ggml_get_name": (S, [TP]), "ggml_set_name": (TP, [TP, S]), "ggml_format_name": (TP, [TP, S]),
    "ggml_set_input": (None, [TP]), "ggml_set_output": (None, [TP]), "ggml_get_unary_op": (I, [TP]),
    # op builders
    "ggml_dup": (TP, [P, TP]), "ggml_add": (TP, [P, TP, TP]), "ggml_add_inplace": (TP, [P, TP, TP]), "ggml_sub": (TP, [P, TP, TP]),
    "ggml_sub_inplace": (TP, [P, TP, TP]), "ggml_mul_inplace": (TP, [P, TP, TP]), "ggml_div_inplace": (TP, [P, TP, TP]),
    "ggml_mul": (TP, [P, TP, TP]), "ggml_div": (TP, [P, TP, TP]), "ggml_neg": (TP, [P, TP]),
    "ggml_scale": (TP, [P, TP, F]), "ggml_scale_inplace": (TP, [P, TP, F]), "ggml_clamp": (TP, [P, TP, F, F]),
    "ggml_sum": (TP, [P, TP]), "ggml_sum_rows": (TP, [P, TP]), "ggml_argmax": (TP, [P, TP]),
    "ggml_argsort": (TP, [P, TP, I]), "ggml_argsort_top_k": (TP, [P, TP, I]), "ggml_top_k": (TP, [P, TP, I]),
    "ggml_arange": (TP, [P, F, F, F]), "ggml_repeat": (TP, [P, TP, TP]), "ggml_repeat_4d": (TP, [P, TP, L, L, L, L]),
    "ggml_concat": (TP, [P, TP, TP, I]), "ggml_pad": (TP, [P, TP, I, I, I, I]),
    "ggml_silu": (TP, [P, TP]), "ggml_gelu": (TP, [P, TP]), "ggml_elu": (TP, [P, TP]),
    "ggml_norm": (TP, [P, TP, F]), "ggml_rms_norm": (TP, [P, TP, F]), "ggml_mul_mat": (TP, [P, TP, TP]),
    "ggml_soft_max": (TP, [P, TP]), "ggml_soft_max_ext": (TP, [P, TP, TP, F, F]),
    "ggml_cast": (TP, [P, TP, I]), "ggml_cpy": (TP, [P, TP, TP]), "ggml_cont": (TP, [P, TP]),
    "ggml_reshape_1d": (TP, [P, TP, L]), "ggml_reshape_2d": (TP, [P, TP, L, L]), "ggml_reshape_3d": (TP, [P, TP, L, L, L]),
    "ggml_reshape_4d": (TP, [P, TP, L, L, L, L]),
    "ggml_view_1d": (TP, [P, TP, L, Z]), "ggml_view_2d": (TP, [P, TP, L, L, Z, Z]), "ggml_view_3d": (TP, [P, TP, L, L, L, Z, Z, Z]),
    "ggml_view_4d": (TP, [P, TP, L, L, L, L, Z, Z, Z, Z]),
    "ggml_permute": (TP, [P, TP, I, I, I, I]), "ggml_transpose": (TP, [P, TP]),
    "ggml_get_rows": (TP, [P, TP, TP]), "ggml_set_rows": (TP, [P, TP, TP, TP]),
    "ggml_im2col": (TP, [P, TP, TP, I, I, I, I, I, I, B, I]), "ggml_conv_1d": (TP, [P, TP, TP, I, I, I]),
    "ggml_conv_transpose_1d": (TP, [P, TP, TP, I, I, I]), "ggml_timestep_embedding": (TP, [P, TP, I, I]),
    # graphs
    "ggml_new_graph": (P, [P]), "ggml_new_graph_custom": (P, [P, Z, B]), "ggml_build_forward_expand": (None, [P, TP]),
    "ggml_graph_clear": (None, [P]), "ggml_graph_size": (I, [P]), "ggml_graph_n_nodes": (I, [P]), "ggml_graph_node": (TP, [P, I]),
    "ggml_graph_nodes": (C.POINTER(TP), [P]), "ggml_graph_n_leafs": (I, [P]), "ggml_graph_leaf": (TP, [P, I]), "ggml_graph_print": (None, [P]),
    "ggml_quantize_row": (None, [I, P, P, L]), "ggml_dequantize_row": (None, [I, P, P, L]),
    # ggml-backend.h
    "ggml_backend_load_all": (None, []), "ggml_backend_reg_count": (Z, []), "ggml_backend_reg_get": (P, [Z]),
    "ggml_backend_reg_name": (S, [P]), "ggml_backend_reg_get_proc_address": (P, [P, S]),
    "ggml_backend_dev_count": (Z, []), "ggml_backend_dev_get": (P, [Z]), "ggml_backend_dev_by_name": (P, [S]),
    "ggml_backend_dev_by_type": (P, [I]), "ggml_backend_dev_name": (S, [P]), "ggml_backend_dev_description": (S, [P]),
    "ggml_backend_dev_type": (I, [P]), "ggml_backend_dev_memory": (None, [P, C.POINTER(Z), C.POINTER(Z)]),
    "ggml_backend_dev_get_props": (None, [P, C.POINTER(DevProps)]), "ggml_backend_dev_backend_reg": (P, [P]),
    "ggml_backend_dev_init": (P, [P, S]),
    "ggml_backend_init_by_name": (P, [S, S]), "ggml_backend_init_by_type": (P, [I, S]), "ggml_backend_init_best": (P, []),
    "ggml_backend_name": (S, [P]), "ggml_backend_free": (None, [P]), "ggml_backend_get_device": (P, [P]),
    "ggml_backend_synchronize": (None, [P]),
    "ggml_backend_alloc_ctx_tensors": (P, [P, P]), "ggml_backend_buffer_free": (None, [P]), "ggml_backend_buffer_get_size": (Z, [P]),
    "ggml_backend_buffer_get_base": (P, [P]), "ggml_backend_buffer_name": (S, [P]), "ggml_backend_buffer_clear": (None, [P, C.c_uint8]),
    "ggml_backend_buffer_is_host": (B, [P]),
    "ggml_backend_tensor_set": (None, [TP, P, Z, Z]), "ggml_backend_tensor_get": (None, [TP, P, Z, Z]),
    "ggml_backend_tensor_memset": (None, [TP, C.c_uint8, Z, Z]), "ggml_backend_tensor_copy": (None, [TP, TP]),
    "ggml_backend_graph_compute": (I, [P, P]), "ggml_backend_supports_op": (B, [P, TP]),
    "ggml_backend_mi355x_get_stats": (None, [P, C.POINTER(Stats)]), "ggml_backend_mi355x_set_flags": (None, [P, I]), "ggml_backend_mi355x_set_capture": (None, [P, I]),
    "ggml_backend_mi355x_get_stream": (P, [P]),
    "ggml_backend_mi355x_make_current": (None, [P]),
    "ggml_backend_mi355x_init_stream": (P, [P]),
    "ggml_backend_tensor_get_async": (None, [P, TP, P, Z, Z]),
    "ggml_backend_event_new": (P, [P]), "ggml_backend_event_free": (None, [P]), "ggml_backend_event_record": (None, [P, P]),
    "ggml_backend_event_synchronize": (None, [P]),
    "ggml_backend_event_wait": (None, [P, P]),
    "ggml_backend_mi355x_get_kernel_profile": (None, [P, C.POINTER(KernelProfile)]),
    # ggml-cpu.h
    "ggml_backend_cpu_init": (P, []), "ggml_backend_is_cpu": (B, [P]), "ggml_backend_cpu_set_n_threads": (None, [P, I]),
    "ggml_graph_compute_with_ctx": (I, [P, P, I]),
    "ggml_backend_cpu_reg": (P, []), "ggml_backend_cpu_set_graph_compute": (None, [P]),
    # gguf.h
    "gguf_init_empty": (P, []), "gguf_init_from_file": (P, [S, GGUFInitParams]), "gguf_free": (None, [P]),
    "gguf_get_version": (C.c_uint32, [P]), "gguf_get_alignment": (Z, [P]), "gguf_get_data_offset": (Z, [P]),
    "gguf_get_n_kv": (L, [P]), "gguf_find_key": (L, [P, S]), "gguf_get_key": (S, [P, L]), "gguf_get_val_str": (S, [P, L]),
    "gguf_get_val_u32": (C.c_uint32, [P, L]), "gguf_set_val_u32": (None, [P, S, C.c_uint32]), "gguf_set_val_str": (None, [P, S, S]),
    "gguf_get_n_tensors": (L, [P]), "gguf_find_tensor": (L, [P, S]), "gguf_get_tensor_name": (S, [P, L]),
    "gguf_get_tensor_type": (I, [P, L]), "gguf_get_tensor_offset": (Z, [P, L]), "gguf_get_tensor_size": (Z, [P, L]),
    "gguf_add_tensor": (None, [P, TP]), "gguf_write_to_file": (B, [P, S, B]),
}

_lib = None


class _Libs:
    """The boundary library plus the harness library behind one attribute namespace (a symbol lives in exactly one of them)."""

    def __init__(self, ggml, hot):
        self.ggml, self.hot = ggml, hot

    def __getattr__(self, name):
        for lib in (self.__dict__["ggml"], self.__dict__["hot"]):
            try:
                fn = getattr(lib, name)
            except AttributeError:
                continue
            setattr(self, name, fn)
            return fn
        raise AttributeError(name)


def load():
    """Load libggml-mi355x.so (+ libmoshi-hot.so) and attach signatures. Raises if a library or a declared symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    for path in (LIB_PATH, HOT_LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with moshi.cpp_amd/build.sh (hipcc --offload-arch=gfx950); "
                               "there is no CPU or Python fallback for the decode hot path")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype, fn.argtypes = res, args
    if missing:
        raise RuntimeError(f"libggml-mi355x.so lacks symbols declared in include/*.h: {missing}")
    stray = [n for n in ("moshi_hot_create", "moshi_hot_sts_frame") if hasattr(lib, n)]
    if stray:
        raise RuntimeError(f"the boundary library must not carry the harness: {stray}")
    hot_lib = C.CDLL(HOT_LIB_PATH, mode=C.RTLD_GLOBAL)
    from . import hot as _hot  # host-side hot-path driver bindings (include/moshi_hot.h)
    _hot.attach(hot_lib)
    _lib = _Libs(lib, hot_lib)
    return _lib


def declared_symbols():
    return sorted(SIGNATURES)
