"""Replay runner for the reference's capture format.

A capture is `<name>.json` + `<name>.tensors`, written by `graph_dump` (src/ggml_cap.h:1395-1538) and read by `replay_test`
(src/replay.h:52-392, src/replay_ops.h). This module reads the same files and replays them on any ggml backend of this library with
the reference's two passes and its tolerances:

  1. every implemented op on its own: sources become fresh leaves filled from the file, the op is rebuilt from its recorded parameters,
     computed, and compared with the recorded output (views / transposes / permutes / conts and ops without data are skipped);
  2. the whole graph from the `forward_expand` roots, leaves = the recorded `new_tensor` entries.

Comparison (replay_ops.h:92-130, 206-243): I32 exact; F32 |a - b| <= eps * max(|a|, |b| over both arrays), eps = 1e-5 for the CPU
path and 1e-2 for a backend.

JSON layout: {"tensor": {id: [op, [src ids], params, [type, [ne0..3], offset, nbytes], name, group, caller]},
              "groups": {id: [name, parent, [tensor ids], [child ids]]}, "forward_expand": [ids], "nbytes": total}
"""
import ctypes as C
import json

import numpy as np

TYPE_IDS = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8, "q8_1": 9, "q2_K": 10, "q3_K": 11, "q4_K": 12, "q5_K": 13,
            "q6_K": 14, "q8_K": 15, "i8": 24, "i16": 25, "i32": 26, "i64": 27, "f64": 28, "bf16": 30}
SKIP_ALONE = ("view", "transpose", "permute", "cont")      # replay.h:282-287


class ReplayError(Exception):
    pass


class Entry:
    def __init__(self, tid, rec):
        if not isinstance(rec, list) or len(rec) != 7:
            raise ReplayError("unexpected item in tensor")
        self.id = tid
        self.op, self.src, self.params, data, self.name, self.group, self.caller = rec
        tname, ne, self.offset, self.nbytes = data
        if tname not in TYPE_IDS:
            raise ReplayError("unknown type")
        self.type, self.ne = TYPE_IDS[tname], [int(x) for x in ne]
        self.tensor = None

    def can_load(self):
        return self.nbytes != 0


class Capture:
    def __init__(self, basename):
        self.basename = basename
        with open(basename + ".json", "rb") as f:
            doc = json.loads(f.read().decode())
        for key in doc:
            if key not in ("tensor", "groups", "forward_expand", "nbytes"):
                raise ReplayError("unknown key")
        self.tensors = [Entry(tid, rec) for tid, rec in doc.get("tensor", {}).items()]      # file order = creation order
        self.by_id = {e.id: e for e in self.tensors}
        self.groups = {gid: {"name": g[0], "parent": g[1], "tensors": g[2], "children": g[3]} for gid, g in doc.get("groups", {}).items()}
        self.forward_expand = doc.get("forward_expand", [])
        self.nbytes = doc.get("nbytes", 0)
        self.validate()

    def validate(self):
        """the cross-reference checks of replay.h:208-250"""
        for e in self.tensors:
            for s in e.src:
                if s not in self.by_id:
                    raise ReplayError("tensor not found")
            if e.group == "0":
                continue
            if e.group not in self.groups:
                raise ReplayError("group not found")
            if e.id not in self.groups[e.group]["tensors"]:
                raise ReplayError("group to tensor reference error")
        for gid, g in self.groups.items():
            for t in g["tensors"]:
                if t not in self.by_id:
                    raise ReplayError("tensor not found")
                if self.by_id[t].group != gid:
                    raise ReplayError("tensor to group reference error")
            for c in g["children"]:
                if c not in self.groups:
                    raise ReplayError("group not found")
                if self.groups[c]["parent"] != gid:
                    raise ReplayError("group to group reference error")

    def read(self, e):
        with open(self.basename + ".tensors", "rb") as f:
            f.seek(e.offset)
            raw = f.read(e.nbytes)
        if len(raw) != e.nbytes:
            raise ReplayError("tensor data truncated")
        return raw


def build_op(L, ctx, e, src):
    """the alloc() of every replay_op_* class (replay_ops.h:246-640); returns None for ops the reference's runner does not implement"""
    p, op, n = e.params, e.op, len(src)

    def need(k, what):
        if n != k:
            raise ReplayError("src size not %s" % what)

    if op in ("add", "sub", "mul", "div"):
        need(2, "2")
        fn = getattr(L, "ggml_%s%s" % (op, "_inplace" if p else ""))
        return fn(ctx, src[0], src[1])
    if op in ("neg", "sum", "elu", "gelu", "silu", "argmax", "cont", "transpose", "soft_max"):
        need(1, "1")
        return getattr(L, "ggml_" + op)(ctx, src[0])
    if op in ("mul_mat", "cpy", "get_rows"):
        need(2, "2")
        return getattr(L, "ggml_" + op)(ctx, src[0], src[1])
    if op == "repeat_4d":
        need(1, "1")
        if len(p) != 4:
            raise ReplayError("ne array size not 4")
        return L.ggml_repeat_4d(ctx, src[0], *[int(x) for x in p])
    if op == "concat":
        need(2, "2")
        if not 0 <= int(p) < 4:
            raise ReplayError("out of range dim value")
        return L.ggml_concat(ctx, src[0], src[1], int(p))
    if op in ("norm", "rms_norm"):
        need(1, "1")
        return getattr(L, "ggml_" + op)(ctx, src[0], float(p))
    if op == "scale":
        need(1, "1")
        return L.ggml_scale(ctx, src[0], float(p))
    if op == "cast":
        need(1, "1")
        if p not in TYPE_IDS:
            raise ReplayError("unknown type")
        return L.ggml_cast(ctx, src[0], TYPE_IDS[p])
    if op == "reshape":
        need(1, "1")
        if len(p) not in (2, 3, 4):
            raise ReplayError("reshape dimensions wrong")
        return getattr(L, "ggml_reshape_%dd" % len(p))(ctx, src[0], *[int(x) for x in p])
    if op == "view":
        need(1, "1")
        if len(p) not in (2, 4, 6, 8):
            raise ReplayError("view params wrong")
        return getattr(L, "ggml_view_%dd" % (len(p) // 2))(ctx, src[0], *[int(x) for x in p])
    if op == "permute":
        need(1, "1")
        if len(p) != 4:
            raise ReplayError("axis array size not 4")
        return L.ggml_permute(ctx, src[0], *[int(x) for x in p])
    if op == "soft_max_ext":
        if n not in (1, 2):
            raise ReplayError("src size not 1 or 2")
        if len(p) != 2:
            raise ReplayError("params array size not 2")
        return L.ggml_soft_max_ext(ctx, src[0], src[1] if n == 2 else None, float(p[0]), float(p[1]))
    if op == "clamp":
        need(1, "1")
        return L.ggml_clamp(ctx, src[0], float(p[0]), float(p[1]))
    if op in ("conv_1d", "conv_transpose_1d"):
        need(2, "2")
        if len(p) != 3:
            raise ReplayError("params array size not 3")
        return getattr(L, "ggml_" + op)(ctx, src[0], src[1], int(p[0]), int(p[1]), int(p[2]))
    if op == "arange":
        need(0, "0")
        return L.ggml_arange(ctx, float(p[0]), float(p[1]), float(p[2]))
    if op == "top_k":
        need(1, "1")
        return L.ggml_top_k(ctx, src[0], int(p))
    if op == "timestep_embedding":
        need(1, "1")
        return L.ggml_timestep_embedding(ctx, src[0], int(p[0]), int(p[1]))
    return None      # replay_op_base: not implemented (new_tensor and anything unknown)


def compare(e, expected_raw, got_raw, eps):
    """check_results (replay_ops.h:206-243): only F32 and I32 outputs are compared"""
    if e.type == TYPE_IDS["i32"]:
        return np.array_equal(np.frombuffer(expected_raw, np.int32), np.frombuffer(got_raw, np.int32))
    if e.type != TYPE_IDS["f32"]:
        raise ReplayError("only f32 / i32 results can be checked")
    a, b = np.frombuffer(expected_raw, np.float32), np.frombuffer(got_raw, np.float32)
    if a.size == 0:
        return True
    mx = max(float(np.abs(a).max()), float(np.abs(b).max()))
    tol = eps * mx if mx > 0 else eps
    return bool(np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol))


class Runner:
    """pkg = the loaded moshi_cpp_amd package (ctypes signatures attached); backend = a ggml_backend_t of this library"""

    def __init__(self, pkg, backend, eps=None, is_cpu=False):
        self.pkg, self.L, self.backend = pkg, pkg.load(), backend
        self.eps = eps if eps is not None else (1e-5 if is_cpu else 1e-2)     # replay.h:326-333

    def _leaf(self, ctx, e):
        ne = (e.ne + [1, 1, 1, 1])[:4]
        return self.L.ggml_new_tensor_4d(ctx, e.type, *ne)

    def _check_alloc(self, e, t):
        tt = t.contents
        if tt.type != e.type or [int(tt.ne[i]) for i in range(4)] != (e.ne + [1, 1, 1, 1])[:4]:
            raise ReplayError("%s %s: rebuilt op has type %d shape %s, capture says %d %s" % (e.op, e.id, tt.type, [int(tt.ne[i]) for i in range(4)], e.type, e.ne))

    def _run(self, ctx, roots, loads, cap):
        L = self.L
        gf = L.ggml_new_graph(ctx)
        for r in roots:
            L.ggml_build_forward_expand(gf, r)
        buf = L.ggml_backend_alloc_ctx_tensors(ctx, self.backend)
        if not buf:
            raise ReplayError("buffer allocation failed")
        try:
            for e, t in loads:
                raw = cap.read(e)
                L.ggml_backend_tensor_set(t, raw, 0, len(raw))
            if L.ggml_backend_graph_compute(self.backend, gf) != 0:
                raise ReplayError("graph_compute failed")
            out = []
            for r in roots:
                n = L.ggml_nbytes(r)
                b = C.create_string_buffer(n)
                L.ggml_backend_tensor_get(r, b, 0, n)
                out.append(b.raw)
            return out
        finally:
            L.ggml_backend_buffer_free(buf)

    def replay(self, cap, per_op=True, full_graph=True):
        """returns {"tested": n, "skipped": n, "failed": [..], "graph_failed": [..]}"""
        L, pkg = self.L, self.pkg
        res = {"tested": 0, "skipped": 0, "failed": [], "graph_failed": []}
        mem = (len(cap.tensors) + 16) * L.ggml_tensor_overhead() + L.ggml_graph_overhead() + (1 << 20)
        if per_op:
            for idx, e in enumerate(cap.tensors):
                ctx = L.ggml_init(pkg.InitParams(mem, None, True))
                try:
                    srcs = [cap.by_id[s] for s in e.src]
                    probe = [self._leaf(ctx, s) for s in srcs]
                    t = build_op(L, ctx, e, probe) if e.op not in SKIP_ALONE else None
                    if t is None:
                        if e.op != "new_tensor":
                            res["skipped"] += 1
                        continue
                    if not e.can_load() or not all(s.can_load() for s in srcs):
                        res["skipped"] += 1
                        continue
                    if e.type not in (TYPE_IDS["f32"], TYPE_IDS["i32"]):     # the reference asserts here ("TODO", replay_ops.h:209); a result it cannot check is skipped
                        res["skipped"] += 1
                        continue
                    self._check_alloc(e, t)
                    got = self._run(ctx, [t], list(zip(srcs, probe)), cap)[0]
                    if not compare(e, cap.read(e), got, self.eps):
                        res["failed"].append("%s %s %d" % (e.op, e.id, idx))
                    res["tested"] += 1
                finally:
                    L.ggml_free(ctx)
        if full_graph and cap.forward_expand:
            ctx = L.ggml_init(pkg.InitParams(mem, None, True))
            try:
                for e in cap.tensors:
                    e.tensor = None
                loads = []

                def alloc(e):      # replay.h:25-50
                    if e.tensor is not None:
                        return e.tensor
                    src = [alloc(cap.by_id[s]) for s in e.src]
                    if e.op == "new_tensor":
                        t = self._leaf(ctx, e)
                        loads.append((e, t))
                    else:
                        t = build_op(L, ctx, e, src)
                        if t is None:
                            raise ReplayError("op %s cannot be rebuilt" % e.op)
                    L.ggml_set_name(t, e.id.encode())
                    e.tensor = t
                    return t
                roots = [alloc(cap.by_id[t]) for t in cap.forward_expand]
                outs = self._run(ctx, roots, loads, cap)
                for tid, got in zip(cap.forward_expand, outs):
                    e = cap.by_id[tid]
                    if e.can_load() and e.type in (TYPE_IDS["f32"], TYPE_IDS["i32"]) and not compare(e, cap.read(e), got, 1e-2):   # replay.h:381
                        res["graph_failed"].append("%s %s" % (e.op, e.id))
            finally:
                for e in cap.tensors:
                    e.tensor = None
                L.ggml_free(ctx)
        return res
