"""Test helper: build ggml graphs through the C-ABI (the way libmoshi's GraphContext does,
src/context.h:227-545 in the reference), run them on a backend and read results back as numpy arrays.

`Runner("hip")` drives the MI355X backend; `Runner("oracle")` drives the host device with the CPU oracle
(oracle/liboracle.so) attached as its graph executor. The same `build(g)` callback is run on both.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

pkg = load_package()
F32, F16, BF16, I32, I64, Q4_0, Q8_0, Q4_K = pkg.F32, pkg.F16, pkg.BF16, pkg.I32, pkg.I64, pkg.Q4_0, pkg.Q8_0, pkg.Q4_K

_oracle_attached = False


def lib():
    return pkg.load()


def attach_oracle():
    """Install the oracle as the CPU device's graph executor (tests / cpu_baseline only)."""
    global _oracle_attached
    if not _oracle_attached:
        olib = load_oracle().load()
        fn = C.cast(olib.oracle_graph_compute, C.c_void_p)
        lib().ggml_backend_cpu_set_graph_compute(fn)
        _oracle_attached = True


def gpu_available():
    L = lib()
    L.ggml_backend_load_all()
    return bool(L.ggml_backend_dev_by_type(pkg.DEV_GPU))


# ---- numpy <-> ggml element encodings -------------------------------------------------------------------
def f32_to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + (0x7FFF + ((u >> 16) & 1))) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def encode(arr, gtype):
    if gtype == F32:
        return np.ascontiguousarray(arr, dtype=np.float32).tobytes()
    if gtype == F16:
        return np.ascontiguousarray(arr, dtype=np.float16).tobytes()
    if gtype == BF16:
        return f32_to_bf16_bits(arr).tobytes()
    if gtype == I32:
        return np.ascontiguousarray(arr, dtype=np.int32).tobytes()
    if gtype == I64:
        return np.ascontiguousarray(arr, dtype=np.int64).tobytes()
    raise ValueError(gtype)


def decode(raw, gtype, shape):
    if gtype == F32:
        return np.frombuffer(raw, dtype=np.float32).reshape(shape).copy()
    if gtype == F16:
        return np.frombuffer(raw, dtype=np.float16).reshape(shape).astype(np.float32)
    if gtype == BF16:
        return bf16_bits_to_f32(np.frombuffer(raw, dtype=np.uint16)).reshape(shape).copy()
    if gtype == I32:
        return np.frombuffer(raw, dtype=np.int32).reshape(shape).copy()
    raise ValueError(gtype)


# ---- synthetic block-quantised rows (SURVEY.md §8d value recipe) --------------------------------------------
def random_q4_K(rng, rows, k):
    """rows x (k/256) random Q4_K super-blocks as raw bytes (uint8 array [rows, k/256*144])."""
    nb = k // 256
    out = np.zeros((rows, nb, 144), dtype=np.uint8)
    d = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -6).astype(np.float16)
    dmin = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -7).astype(np.float16)
    out[:, :, 0:2] = d.view(np.uint8).reshape(rows, nb, 2)
    out[:, :, 2:4] = dmin.view(np.uint8).reshape(rows, nb, 2)
    sc = rng.integers(1, 64, size=(rows, nb, 8), dtype=np.uint8)
    mn = rng.integers(1, 64, size=(rows, nb, 8), dtype=np.uint8)
    s = np.zeros((rows, nb, 12), dtype=np.uint8)
    for j in range(4):
        s[:, :, j] = (sc[:, :, j] & 63) | ((sc[:, :, j + 4] >> 4) << 6)
        s[:, :, j + 4] = (mn[:, :, j] & 63) | ((mn[:, :, j + 4] >> 4) << 6)
        s[:, :, j + 8] = (sc[:, :, j + 4] & 0xF) | ((mn[:, :, j + 4] & 0xF) << 4)
    out[:, :, 4:16] = s
    out[:, :, 16:144] = rng.integers(0, 256, size=(rows, nb, 128), dtype=np.uint8)
    return out.reshape(rows, nb * 144)


def random_q8_0(rng, rows, k):
    nb = k // 32
    out = np.zeros((rows, nb, 34), dtype=np.uint8)
    d = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -8).astype(np.float16)
    out[:, :, 0:2] = d.view(np.uint8).reshape(rows, nb, 2)
    out[:, :, 2:34] = rng.integers(-127, 128, size=(rows, nb, 32), dtype=np.int8).view(np.uint8)
    return out.reshape(rows, nb * 34)


def random_q4_0(rng, rows, k):
    nb = k // 32
    out = np.zeros((rows, nb, 18), dtype=np.uint8)
    d = (np.abs(rng.standard_normal((rows, nb))) * 2.0 ** -5).astype(np.float16)
    out[:, :, 0:2] = d.view(np.uint8).reshape(rows, nb, 2)
    out[:, :, 2:18] = rng.integers(0, 256, size=(rows, nb, 16), dtype=np.uint8)
    return out.reshape(rows, nb * 18)


def dequantize(raw_rows, gtype, k):
    """Dequantise with the ORACLE's row routine (used as the expectation side of tests)."""
    olib = load_oracle().load()
    rows = raw_rows.shape[0]
    out = np.zeros((rows, k), dtype=np.float32)
    for r in range(rows):
        src = np.ascontiguousarray(raw_rows[r])
        olib.oracle_dequantize_row(gtype, src.ctypes.data, out[r].ctypes.data, k)
    return out


class Graph:
    """One ggml context + graph on one backend."""

    def __init__(self, backend_kind, mem_mb=64, graph_size=8192):
        self.L = lib()
        self.L.ggml_backend_load_all()
        self.kind = backend_kind
        if backend_kind == "hip":
            self.backend = self.L.ggml_backend_init_by_type(pkg.DEV_GPU, None)
            if not self.backend:
                raise RuntimeError("no MI355X device: the HIP backend cannot run (no CPU fallback exists)")
        else:
            attach_oracle()
            self.backend = self.L.ggml_backend_init_by_type(pkg.DEV_CPU, None)
        self.ctx = self.L.ggml_init(pkg.InitParams(mem_mb * 1024 * 1024, None, True))
        self.graph_size = graph_size
        self.uploads = []   # (tensor, bytes)
        self.buffer = None
        self.graph = None
        self._keep = []

    # tensors ---------------------------------------------------------------------------------------------
    def new(self, gtype, *ne):
        ne = list(ne) + [1] * (4 - len(ne))
        return self.L.ggml_new_tensor_4d(self.ctx, gtype, *ne)

    def input(self, arr, gtype=F32, ne=None):
        """numpy array with shape (ne3, ne2, ne1, ne0)-style (last axis fastest) -> tensor with that data."""
        arr = np.asarray(arr)
        if ne is None:
            ne = list(reversed(arr.shape))
        t = self.new(gtype, *ne)
        self.uploads.append((t, encode(arr, gtype)))
        return t

    def input_raw(self, raw, gtype, *ne):
        t = self.new(gtype, *ne)
        self.uploads.append((t, np.ascontiguousarray(raw).tobytes()))
        return t

    def __getattr__(self, name):
        fn = getattr(self.L, "ggml_" + name)

        def call(*args):
            return fn(self.ctx, *args)
        return call

    # execution -------------------------------------------------------------------------------------------
    def build(self, outputs):
        self.graph = self.L.ggml_new_graph_custom(self.ctx, self.graph_size, False)
        for o in outputs:
            self.L.ggml_build_forward_expand(self.graph, o)

    def alloc(self):
        self.buffer = self.L.ggml_backend_alloc_ctx_tensors(self.ctx, self.backend)
        assert self.buffer
        for t, raw in self.uploads:
            assert len(raw) == self.L.ggml_nbytes(t), (len(raw), self.L.ggml_nbytes(t))
            self.L.ggml_backend_tensor_set(t, raw, 0, len(raw))

    def set(self, t, arr, gtype=None):
        raw = encode(arr, t.contents.type if gtype is None else gtype)
        self.L.ggml_backend_tensor_set(t, raw, 0, len(raw))

    def compute(self):
        st = self.L.ggml_backend_graph_compute(self.backend, self.graph)
        assert st == 0, f"graph_compute failed: {st}"

    def get(self, t):
        n = self.L.ggml_nbytes(t)
        tt = t.contents
        assert self.L.ggml_is_contiguous(t), "read-back helper needs a contiguous tensor"
        buf = C.create_string_buffer(n)
        self.L.ggml_backend_tensor_get(t, buf, 0, n)
        shape = tuple(int(tt.ne[i]) for i in (3, 2, 1, 0))
        return decode(buf.raw, tt.type, shape)

    def stats(self):
        s = pkg.Stats()
        self.L.ggml_backend_mi355x_get_stats(self.backend, C.byref(s))
        return s

    def set_flags(self, flags):
        self.L.ggml_backend_mi355x_set_flags(self.backend, flags)

    def free(self):
        if self.buffer:
            self.L.ggml_backend_buffer_free(self.buffer)
        self.L.ggml_free(self.ctx)
        self.L.ggml_backend_free(self.backend)
        self.buffer = None


def run_graph(kind, build_fn, read=None, flags=0, repeat=1, pad_nodes=0):
    """build_fn(g) -> list of output tensors (or (outputs, extra_reads)). Returns list of numpy arrays."""
    g = Graph(kind)
    try:
        if kind == "hip" and flags:
            g.set_flags(flags)
        res = build_fn(g)
        outs, extra = res if isinstance(res, tuple) else (res, [])
        g.build(outs)
        g.alloc()
        for _ in range(repeat):
            g.compute()
        return [g.get(o) for o in list(outs) + list(extra)], (g.stats() if kind == "hip" else None)
    finally:
        g.free()


def compare(build_fn, rtol=1e-5, atol_rel=1e-5, flags=0, exact_int=True):
    """Run on HIP and oracle; F32 outputs must agree within atol_rel * max|ref| (+ rtol); I32 exactly."""
    ref, _ = run_graph("oracle", build_fn)
    got, stats = run_graph("hip", build_fn, flags=flags)
    assert len(ref) == len(got)
    for i, (r, h) in enumerate(zip(ref, got)):
        assert r.shape == h.shape, (i, r.shape, h.shape)
        if r.dtype == np.int32:
            assert np.array_equal(r, h), f"output {i}: int mismatch at {np.argwhere(r != h)[:5]}"
        else:
            fin = np.isfinite(r)
            assert np.array_equal(fin, np.isfinite(h)), f"output {i}: non-finite pattern differs"
            scale = float(np.max(np.abs(r[fin]))) if fin.any() else 1.0
            err = np.abs(r[fin] - h[fin])
            tol = atol_rel * max(scale, 1e-30) + rtol * np.abs(r[fin])
            bad = err > tol
            assert not bad.any(), f"output {i}: max err {err.max():.3e} vs scale {scale:.3e} ({bad.sum()} of {bad.size} beyond tol)"
    return ref, got, stats
