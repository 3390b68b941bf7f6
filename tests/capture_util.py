"""Test-side writer of the reference's capture layout (graph_dump, src/ggml_cap.h:1395-1538): builds a graph from a node list on a backend,
computes it, and writes `<base>.json` + `<base>.tensors` with every contiguous result. The expected data therefore comes from whatever
backend computed it (the CPU oracle in the tests); the READER under test is moshi_cpp_amd/replay.py."""
import ctypes as C
import json

import numpy as np

import ggml_util as gu
from moshi_cpp_amd import replay

TYPE_NAMES = {v: k for k, v in replay.TYPE_IDS.items()}


def write_capture(base, nodes, backend_kind="oracle"):
    """nodes: list of dicts {"id", "op", "src": [...], "params": json value, optional "type"/"ne"/"raw" for new_tensor}"""
    g = gu.Graph(backend_kind)
    L = g.L
    try:
        t_of, leaves = {}, []
        for n in nodes:
            src = [t_of[s] for s in n["src"]]
            if n["op"] == "new_tensor":
                ne = (list(n["ne"]) + [1, 1, 1, 1])[:4]
                t = L.ggml_new_tensor_4d(g.ctx, replay.TYPE_IDS[n["type"]], *ne)
                leaves.append((t, n["raw"]))
            else:
                e = replay.Entry(n["id"], [n["op"], n["src"], n["params"], ["f32", [1, 1, 1, 1], 0, 0], "", "1", ""])
                t = replay.build_op(L, g.ctx, e, src)
                assert t, n["op"]
            t_of[n["id"]] = t
        consumed = {s for n in nodes for s in n["src"]}
        roots = [n["id"] for n in nodes if n["id"] not in consumed and n["op"] != "new_tensor"]
        g.build([t_of[r] for r in roots])
        g.buffer = L.ggml_backend_alloc_ctx_tensors(g.ctx, g.backend)
        assert g.buffer
        for t, raw in leaves:
            assert len(raw) == L.ggml_nbytes(t), (len(raw), L.ggml_nbytes(t))
            L.ggml_backend_tensor_set(t, raw, 0, len(raw))
        g.compute()
        doc = {"tensor": {}, "groups": {"1": ["test", "0", [n["id"] for n in nodes], []]}, "forward_expand": roots, "nbytes": 0}
        total = 0
        with open(base + ".tensors", "wb") as fbin:
            for n in nodes:
                t = t_of[n["id"]]
                tt = t.contents
                side_effect = n["op"] == "cpy"                                   # written as "0,0" (ggml_cap.h:1430)
                if L.ggml_is_contiguous(t) and not side_effect:
                    nb = L.ggml_nbytes(t)
                    if n["op"] == "new_tensor":
                        fbin.write(n["raw"])                                     # as uploaded: an in-place consumer (clamp) may have rewritten it since
                    else:
                        buf = C.create_string_buffer(nb)
                        L.ggml_backend_tensor_get(t, buf, 0, nb)
                        fbin.write(buf.raw)
                    data = [TYPE_NAMES[tt.type], [int(tt.ne[i]) for i in range(4)], total, nb]
                    total += nb
                else:
                    data = [TYPE_NAMES[tt.type], [int(tt.ne[i]) for i in range(4)], 0, 0]
                doc["tensor"][n["id"]] = [n["op"], n["src"], n["params"], data, n.get("name", ""), "1", n.get("caller", "")]
        doc["nbytes"] = total
        with open(base + ".json", "w") as f:
            json.dump(doc, f)
    finally:
        g.free()


def leaf(tid, arr, gtype="f32"):
    arr = np.asarray(arr)
    return {"id": tid, "op": "new_tensor", "src": [], "params": None, "type": gtype, "ne": list(reversed(arr.shape)), "raw": gu.encode(arr, replay.TYPE_IDS[gtype])}


def leaf_raw(tid, raw, gtype, ne):
    return {"id": tid, "op": "new_tensor", "src": [], "params": None, "type": gtype, "ne": list(ne), "raw": np.ascontiguousarray(raw).tobytes()}


def op(tid, name, src, params=None):
    return {"id": tid, "op": name, "src": list(src), "params": params}
