"""Node-by-node, teacher-forced comparison of ONE transformer layer between two executors (oracle vs MI355X backend, or oracle vs a
perturbed oracle on CPU) through moshi_hot_layer_probe (include/moshi_hot.h).

ggml's arithmetic on this path has step functions: mul_mat rounds its activation operand to the weight's vec_dot type (Q8_K for Q4_K,
Q8_0 for Q8_0 / Q4_0, BF16 / F16 for half weights) and set_rows rounds the K / V rows to the BF16 ring. Two correct implementations that
differ by float re-association (1e-7) agree to summation noise on every node UNTIL one of those roundings lands on the other side of a
tie; from there on the layer differs by a quantiser step. This module makes that explicit instead of hiding it in a loose tolerance:

  * every rounding site (a MUL_MAT's src[1], a SET_ROWS' src[0]) is re-quantised on the host (numpy restatement of ggml's reference
    quantisers) from both executors' pre-rounding values, and the elements that round differently are COUNTED (they are ties by
    construction: the two inputs agree to CLEAN_TOL and still round apart);
  * a node is `tainted` iff a flip happened at a site upstream of it inside this layer;
  * clean nodes must agree to CLEAN_TOL (2e-6 of the node's max, the op tests' bar), tainted nodes to TAINT_TOL, flips per site are bounded.
"""
import ctypes as C

import numpy as np

import ggml_util as gu
import hot_util as hu

L = hu.L
pkg = hu.pkg
CLEAN_TOL = 2e-6
TAINT_TOL = 1e-2
VIEW_OPS = {"VIEW", "RESHAPE", "PERMUTE", "TRANSPOSE"}


class Node:
    __slots__ = ("idx", "ptr", "op", "type", "ne", "data", "view_src", "src", "src_type", "contiguous", "values")


def _read(tp):
    t = tp.contents
    nb = L.ggml_nbytes(tp)
    if t.type == pkg.F32:
        a = np.zeros(nb // 4, np.float32)
        L.ggml_backend_tensor_get(tp, a.ctypes.data, 0, nb)
        return a
    if t.type == pkg.BF16:
        raw = np.zeros(nb // 2, np.uint16)
        L.ggml_backend_tensor_get(tp, raw.ctypes.data, 0, nb)
        return gu.bf16_bits_to_f32(raw)
    return None


def probe(model, which, layer, weight_set, x, offset):
    """-> (nodes in execution order, layer output)"""
    nodes = []

    def visit(user, i, tptr):
        tp = C.cast(tptr, pkg.TP)
        t = tp.contents
        n = Node()
        n.idx, n.ptr, n.op, n.type = i, tptr, L.ggml_op_name(t.op).decode(), t.type
        n.ne = tuple(t.ne)
        n.data = t.data
        n.view_src = C.cast(t.view_src, C.c_void_p).value
        n.src = [C.cast(t.src[k], C.c_void_p).value for k in range(pkg.GGML_MAX_SRC)]
        n.src_type = [t.src[k].contents.type if t.src[k] else None for k in range(pkg.GGML_MAX_SRC)]
        n.contiguous = bool(L.ggml_is_contiguous(tp))
        n.values = _read(tp) if (n.contiguous and t.type == pkg.F32 and t.data) else None
        nodes.append(n)

    cb = hu.hot.NODE_VISITOR(visit)
    x = np.ascontiguousarray(x, np.float32)
    dim = x.size
    y = np.zeros(dim, np.float32)
    n = L.moshi_hot_layer_probe(model.m, which, layer, weight_set, x.ctypes.data, offset, y.ctypes.data, C.cast(cb, C.c_void_p), None)
    assert n == len(nodes)
    return nodes, y


# ---- ggml's reference quantisers, restated in numpy (quantize_row_q8_K_ref / quantize_row_q8_0_ref / fp32 -> bf16 / fp16) [ggml-upstream] ----
def q8_K(v):
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 256)
    out = np.zeros(v.shape, np.int32)
    for b in range(v.shape[0]):
        ax = np.abs(v[b])
        j = int(np.argmax(ax))                       # the FIRST element of largest magnitude
        if ax[j] == 0:
            continue
        iscale = np.float32(-127.0) / v[b, j]
        out[b] = np.minimum(127, np.rint((iscale * v[b]).astype(np.float32)).astype(np.int32))
    return out.reshape(-1)


def q8_0(v):
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 32)
    amax = np.abs(v).max(axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    idv = np.where(d != 0, np.float32(1.0) / np.where(d != 0, d, 1), 0).astype(np.float32)
    p = (v * idv[:, None]).astype(np.float32)
    q = (np.sign(p) * np.floor(np.abs(p) + np.float32(0.5))).astype(np.int32).reshape(-1)   # roundf: half away from zero
    # the block scale is stored as F16: when the block maximum differs by a float ulp the stored scale can land on the neighbouring F16 value (2^-11
    # relative) with every quant unchanged - a rounding site of its own, reported as one more "element" per block
    return np.concatenate([q, d.astype(np.float16).view(np.uint16).astype(np.int32)])


def rounded(v, wtype):
    """the integer / bit pattern each element becomes when ggml converts an activation for a weight (or destination) of `wtype`"""
    if wtype == pkg.Q4_K:
        return q8_K(v)
    if wtype in (pkg.Q8_0, pkg.Q4_0):
        return q8_0(v)
    if wtype == pkg.BF16:
        return gu.f32_to_bf16_bits(v).astype(np.int32)
    if wtype == pkg.F16:
        return np.ascontiguousarray(v, np.float32).astype(np.float16).view(np.uint16).astype(np.int32)
    return None                                      # F32: no rounding


def _source_values(by_ptr, ptr):
    """values of the nearest ancestor through layout-only nodes (a rounding site's operand may be a permuted view of a dense node)"""
    for _ in range(8):
        n = by_ptr.get(ptr)
        if n is None:
            return None
        if n.values is not None and not np.isnan(n.values).any():
            return n.values
        if n.op not in VIEW_OPS:
            return None
        ptr = n.src[0]
    return None


def compare_layer(ref, got, where, clean_tol=CLEAN_TOL, taint_tol=TAINT_TOL, max_flip_frac=0.005, taint_in=None, hidden_flips=False, cache_tainted=False):
    """ref / got: node lists of the same layer graph from two executors (got may leave fused-away nodes as NaN).
    taint_in: taint map (node index -> bool) of another comparison of the same layer (the per-node device run), for a run whose rounding
    sites are invisible (fused kernels). hidden_flips: such a run may also flip where the visible one did not - a node beyond clean_tol
    with no known flip upstream is then COUNTED (stats["hidden"]) and treated as a flip from there on, still bounded by taint_tol.
    cache_tainted: the KV ring of this layer already holds rows that rounded differently in an earlier probe (stats["cache_flips"] of that probe):
    the ring is then a tainted input of the attention products.
    Returns a dict of statistics; asserts the per-node bars."""
    assert len(ref) == len(got) and all(a.op == b.op and a.ne == b.ne for a, b in zip(ref, got)), f"{where}: graphs differ"
    ra = {n.ptr: n for n in ref}
    rb = {n.ptr: n for n in got}
    taint = {}
    stats = {"nodes": 0, "clean": 0, "tainted": 0, "sites": 0, "flips": 0, "worst_clean": 0.0, "worst_tainted": 0.0, "taint": {}, "site_flips": [], "hidden": 0, "cache_flips": 0}
    for a, b in zip(ref, got):
        t = any(taint.get(s, False) for s in a.src if s) or bool(a.view_src and taint.get(a.view_src, False))
        if taint_in is not None:
            t = t or taint_in.get(a.idx, False)
        if cache_tainted and a.op == "SET_ROWS":
            t = True
        site = None
        if a.op == "MUL_MAT":
            site = (a.src[1], a.src_type[0], b.src[1])
        elif a.op == "SET_ROWS":
            site = (a.src[0], a.type, b.src[0])
        if site is not None:
            va, vb = _source_values(ra, site[0]), _source_values(rb, site[2])
            if va is not None and vb is not None:
                qa, qb = rounded(va, site[1]), rounded(vb, site[1])
                if qa is not None:
                    nf = int(np.count_nonzero(qa != qb))
                    # only a site whose operand is itself clean measures tie flips; below a flip the operands differ by a quantiser step already
                    sn = ra.get(site[0])
                    src_tainted = taint.get(site[0], False) or bool(taint_in is not None and sn is not None and taint_in.get(sn.idx, False))
                    if not src_tainted:
                        stats["sites"] += 1
                        stats["flips"] += nf
                        stats["site_flips"].append((a.idx, a.op, site[1], nf, va.size))
                        assert nf <= max(2, max_flip_frac * va.size), f"{where} node {a.idx} {a.op}: {nf} of {va.size} clean activation values round differently"
                    if nf:
                        t = True
                        if a.op == "SET_ROWS":
                            stats["cache_flips"] += nf
        taint[a.ptr] = t
        stats["taint"][a.idx] = t
        if a.values is None or b.values is None or a.view_src or np.isnan(b.values).all():
            continue
        assert not np.isnan(b.values).any(), f"{where} node {a.idx} {a.op}: partially written"
        if not np.isfinite(a.values).all():
            assert np.array_equal(np.isfinite(a.values), np.isfinite(b.values)), f"{where} node {a.idx}: infinities differ"
            continue
        e = hu.rel_err(a.values, b.values)
        stats["nodes"] += 1
        if t:
            stats["tainted"] += 1
            stats["worst_tainted"] = max(stats["worst_tainted"], e)
            assert e <= taint_tol, f"{where} node {a.idx} {a.op} {a.ne}: rel err {e:.2e} downstream of a rounding flip"
        elif hidden_flips and clean_tol < e <= taint_tol:
            stats["hidden"] += 1
            taint[a.ptr] = stats["taint"][a.idx] = True
        else:
            stats["clean"] += 1
            stats["worst_clean"] = max(stats["worst_clean"], e)
            assert e <= clean_tol, f"{where} node {a.idx} {a.op} {a.ne}: rel err {e:.2e} with no rounding flip upstream"
    return stats
