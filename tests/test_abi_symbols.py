"""CPU: the C-ABI library loads without a GPU and exports every symbol include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import ggml_util as gu

ROOT = gu.ROOT
HEADERS = ["ggml.h", "ggml-backend.h", "ggml-cpu.h", "gguf.h", "moshi_hot.h"]


def declared_in_headers():
    names = set()
    for h in HEADERS:
        src = open(os.path.join(ROOT, "include", h)).read()
        for m in re.finditer(r"GGML_API[^;{]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src):
            names.add(m.group(1))
    names.discard("__attribute__")
    return names


def test_library_loads_and_exports_every_declared_symbol():
    lib = gu.lib()
    missing = [n for n in sorted(declared_in_headers()) if not hasattr(lib, n)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_python_signature_table_covers_the_headers():
    from moshi_cpp_amd import hot
    table = set(gu.pkg.SIGNATURES) | set(hot.SIGNATURES)
    hdr = declared_in_headers()
    assert not (hdr - table - {"ggml_abort"}), f"headers declare symbols the binding table lacks: {sorted(hdr - table)}"
    assert not (table - hdr), f"binding table lists symbols no header declares: {sorted(table - hdr)}"


def test_devices_and_cpu_backend_without_gpu_compute():
    L = gu.lib()
    L.ggml_backend_load_all()
    assert L.ggml_backend_dev_count() >= 1
    cpu = L.ggml_backend_init_by_type(gu.pkg.DEV_CPU, None)
    assert cpu and L.ggml_backend_is_cpu(cpu)
    dev = L.ggml_backend_get_device(cpu)
    props = gu.pkg.DevProps()
    L.ggml_backend_dev_get_props(dev, C.byref(props))
    assert props.name == b"CPU" and props.type == gu.pkg.DEV_CPU and props.memory_total > 0
    reg = L.ggml_backend_dev_backend_reg(dev)
    assert L.ggml_backend_reg_get_proc_address(reg, b"ggml_backend_set_n_threads")
    assert L.ggml_backend_init_by_name(b"no-such-device", None) is None
    L.ggml_backend_free(cpu)


def test_cpu_device_has_no_builtin_executor():
    """The product never computes on the CPU by itself: without the oracle attached graph_compute fails loudly."""
    L = gu.lib()
    L.ggml_backend_cpu_set_graph_compute(None)
    gu._oracle_attached = False
    cpu = L.ggml_backend_init_by_type(gu.pkg.DEV_CPU, None)
    ctx = L.ggml_init(gu.pkg.InitParams(1 << 20, None, True))
    a = L.ggml_new_tensor_1d(ctx, gu.F32, 4)
    y = L.ggml_scale(ctx, a, 2.0)
    g = L.ggml_new_graph(ctx)
    L.ggml_build_forward_expand(g, y)
    buf = L.ggml_backend_alloc_ctx_tensors(ctx, cpu)
    assert L.ggml_backend_graph_compute(cpu, g) == -1   # GGML_STATUS_FAILED

    # ... with one exception: the load-time re-quantisation graph of the reference's weight loader (a lone ggml_cast on host
    # tensors, src/loader.h:180-187) is converted on the host
    w = L.ggml_new_tensor_2d(ctx, 0, 512, 3)           # F32 [512, 3]
    q = L.ggml_cast(ctx, w, 12)                        # -> Q4_K
    back = L.ggml_cast(ctx, q, 0)                      # -> F32 again
    g2 = L.ggml_new_graph(ctx)
    L.ggml_build_forward_expand(g2, back)
    buf2 = L.ggml_backend_alloc_ctx_tensors(ctx, cpu)
    assert buf2
    import numpy as np
    x = (np.random.default_rng(0).standard_normal((3, 512)) * 0.05).astype(np.float32)
    L.ggml_backend_tensor_set(w, x.ctypes.data, 0, x.nbytes)
    assert L.ggml_backend_graph_compute(cpu, g2) == 0
    y = np.zeros_like(x)
    L.ggml_backend_tensor_get(back, y.ctypes.data, 0, y.nbytes)
    assert np.abs(y - x).max() < 0.02 and np.abs(y - x).mean() < 0.004   # 4-bit round trip of N(0, 0.05)
    L.ggml_backend_buffer_free(buf2)

    # ... and the codebook preparation of the Mimi loader (core_vq.h:58-85): embedding_sum / cont(transpose(clamp(cluster_usage)))
    es = L.ggml_new_tensor_2d(ctx, 0, 8, 5)            # [dim 8, 5 centroids]
    cu = L.ggml_new_tensor_2d(ctx, 0, 5, 1)            # [5, 1]
    cl = L.ggml_clamp(ctx, cu, 1e-5, float("inf"))
    emb = L.ggml_div(ctx, es, L.ggml_cont(ctx, L.ggml_transpose(ctx, cl)))
    g3 = L.ggml_new_graph(ctx)
    L.ggml_build_forward_expand(g3, emb)
    buf3 = L.ggml_backend_alloc_ctx_tensors(ctx, cpu)
    a = np.arange(40, dtype=np.float32).reshape(5, 8)
    u = np.array([[2.0, 0.0, 4.0, 1e-9, 0.5]], np.float32)
    L.ggml_backend_tensor_set(es, a.ctypes.data, 0, a.nbytes)
    L.ggml_backend_tensor_set(cu, u.ctypes.data, 0, u.nbytes)
    assert L.ggml_backend_graph_compute(cpu, g3) == 0
    o = np.zeros_like(a)
    L.ggml_backend_tensor_get(emb, o.ctypes.data, 0, o.nbytes)
    np.testing.assert_array_equal(o, a / np.maximum(u.reshape(5, 1), np.float32(1e-5)))
    L.ggml_backend_buffer_free(buf3)
    L.ggml_backend_buffer_free(buf)
    L.ggml_free(ctx)
    L.ggml_backend_free(cpu)
