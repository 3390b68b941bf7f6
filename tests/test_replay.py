"""Replay runner for the reference's capture layout (moshi_cpp_amd/replay.py; src/replay.h, src/replay_ops.h, src/ggml_cap.h:1395-1538):
the reader against a file written literally in graph_dump's formatting, every op class of replay_ops.h round-tripped through a capture
computed by the CPU oracle, the reference's validation errors — and (-m gpu) the same captures replayed on the MI355X backend with the
reference's backend tolerance."""
import os
import struct

import numpy as np
import pytest

import capture_util as cu
import ggml_util as gu
from ggml_util import Q4_K
from moshi_cpp_amd import replay

pkg = gu.pkg


def make_backend(kind):
    L = gu.lib()
    L.ggml_backend_load_all()
    if kind == "hip":
        be = L.ggml_backend_init_by_type(pkg.DEV_GPU, None)
        if not be:
            raise RuntimeError("no MI355X device")
        return be
    gu.attach_oracle()
    return L.ggml_backend_init_by_type(pkg.DEV_CPU, None)


def all_ops_nodes():
    r = np.random.default_rng(11)
    f = lambda *s: r.standard_normal(s).astype(np.float32)
    n = [cu.leaf("a", f(3, 8)), cu.leaf("b", f(3, 8) + 3.0), cu.leaf("row", f(1, 8)),
         cu.op("add", "add", ["a", "b"], False), cu.op("sub", "sub", ["a", "b"], False), cu.op("mul", "mul", ["a", "row"], False), cu.op("div", "div", ["a", "b"], False),
         cu.op("neg", "neg", ["a"]), cu.op("sum", "sum", ["a"]), cu.op("rep", "repeat_4d", ["row"], [8, 3, 2, 1]), cu.op("cat", "concat", ["a", "b"], 1),
         cu.op("elu", "elu", ["a"]), cu.op("gelu", "gelu", ["a"]), cu.op("silu", "silu", ["a"]), cu.op("norm", "norm", ["a"], 1e-5), cu.op("rms", "rms_norm", ["a"], 1e-8),
         cu.leaf("w", f(5, 8)), cu.op("mm", "mul_mat", ["w", "a"]), cu.op("amax", "argmax", ["mm"]), cu.op("scale", "scale", ["a"], 0.25),
         cu.op("cast", "cast", ["a"], "f16"), cu.op("back", "cast", ["cast"], "f32"), cu.op("tr", "transpose", ["a"]), cu.op("cont", "cont", ["tr"]),
         cu.op("resh", "reshape", ["a"], [4, 6]), cu.op("view", "view", ["a"], [4, 3, 32, 16]), cu.op("vcont", "cont", ["view"]), cu.op("perm", "permute", ["resh"], [1, 0, 2, 3]),
         cu.op("sm", "soft_max", ["a"]), cu.leaf("mask", np.where(r.random((3, 8)) < 0.3, -np.inf, 0.0).astype(np.float32)),
         cu.op("sme", "soft_max_ext", ["a", "mask"], [0.5, 0.0]), cu.op("sm1", "soft_max_ext", ["a"], [2.0, 0.0]),
         cu.leaf("idx", np.array([2, 0], np.int32), "i32"), cu.op("rows", "get_rows", ["a", "idx"]), cu.leaf("c_in", f(3, 8)), cu.op("clamp", "clamp", ["c_in"], [-0.5, 0.5]),     # ggml_clamp works in place: its own leaf
         cu.leaf("ker", f(4, 2, 3), "f16"), cu.leaf("sig", f(2, 16)), cu.op("conv", "conv_1d", ["ker", "sig"], [1, 0, 1]),
         cu.leaf("kt", f(2, 4, 3)), cu.op("convt", "conv_transpose_1d", ["kt", "sig"], [2, 0, 1]),
         cu.op("ar", "arange", [], [0.0, 6.0, 1.0]), cu.op("tse", "timestep_embedding", ["ar"], [16, 10000]),
         cu.leaf_raw("wq", gu.random_q4_K(r, 6, 256), "q4_K", [256, 6]), cu.leaf("xq", f(2, 256)), cu.op("mmq", "mul_mat", ["wq", "xq"])]
    return n


def test_reader_takes_a_file_in_graph_dumps_own_formatting(tmp_path):
    # byte-for-byte the way graph_dump prints it (ggml_cap.h:1406-1531): newline-separated entries, ne always four long, group "0" allowed
    base = str(tmp_path / "tiny")
    x = np.array([1.0, -2.0, 3.0, 0.5], np.float32)
    y = (x * np.float32(2.0)).astype(np.float32)
    open(base + ".tensors", "wb").write(x.tobytes() + y.tobytes())
    open(base + ".json", "w").write('{"tensor":{\n"1":["new_tensor",[],null,["f32",[4,1,1,1],0,16],"x","0",""],\n'
                                    '"2":["scale",["1"],2,["f32",[4,1,1,1],16,16],"","0","moshi_test"]},\n"groups":{\n},\n"forward_expand":[\n"2"],\n"nbytes":32}')
    cap = replay.Capture(base)
    assert [e.op for e in cap.tensors] == ["new_tensor", "scale"] and cap.forward_expand == ["2"] and cap.nbytes == 32
    res = replay.Runner(pkg, make_backend("oracle"), is_cpu=True).replay(cap)
    assert res == {"tested": 1, "skipped": 0, "failed": [], "graph_failed": []}
    open(base + ".tensors", "wb").write(x.tobytes() + (y * np.float32(1.001)).astype(np.float32).tobytes())     # 1e-3 off: beyond 1e-5, within a backend's 1e-2
    assert replay.Runner(pkg, make_backend("oracle"), is_cpu=True).replay(cap)["failed"] == ["scale 2 1"]
    assert replay.Runner(pkg, make_backend("oracle"), eps=1e-2).replay(cap)["failed"] == []


def test_every_replay_op_round_trips_through_a_capture_on_the_cpu_device(tmp_path):
    base = str(tmp_path / "ops")
    nodes = all_ops_nodes()
    cu.write_capture(base, nodes)
    cap = replay.Capture(base)
    res = replay.Runner(pkg, make_backend("oracle"), is_cpu=True).replay(cap)
    assert res["failed"] == [] and res["graph_failed"] == []
    n_ops = sum(1 for n in nodes if n["op"] != "new_tensor")
    # alone-skipped: transpose, view, permute, cont x2; children of non-contiguous results (cont of tr / view) are skipped too, like the reference does
    assert res["tested"] + res["skipped"] == n_ops and res["tested"] >= n_ops - 8, res


def test_reference_validation_errors(tmp_path):
    base = str(tmp_path / "bad")
    open(base + ".tensors", "wb").write(b"\0" * 16)
    body = '{"tensor":{"1":["new_tensor",[],null,["%s",[4,1,1,1],0,16],"x","%s",""]},"groups":{%s},"forward_expand":[],"nbytes":16%s}'
    for args, msg in ((("f33", "0", "", ""), "unknown type"), (("f32", "7", "", ""), "group not found"),
                      (("f32", "1", '"1":["g","0",[],[]]', ""), "group to tensor reference error"), (("f32", "0", "", ',"extra":1'), "unknown key")):
        open(base + ".json", "w").write(body % args)
        with pytest.raises(replay.ReplayError, match=msg):
            replay.Capture(base)
    open(base + ".json", "w").write('{"tensor":{"2":["neg",["9"],null,["f32",[4,1,1,1],0,16],"","0",""]},"groups":{},"forward_expand":[],"nbytes":16}')
    with pytest.raises(replay.ReplayError, match="tensor not found"):
        replay.Capture(base)


@pytest.mark.gpu
def test_captures_replay_on_the_device_with_the_reference_backend_tolerance(tmp_path):
    base = str(tmp_path / "ops")
    cu.write_capture(base, all_ops_nodes())           # expected data from the CPU oracle
    cap = replay.Capture(base)
    res = replay.Runner(pkg, make_backend("hip")).replay(cap)                 # eps 1e-2 (replay.h:326-328)
    assert res["failed"] == [] and res["graph_failed"] == [] and res["tested"] > 25, res
    tight = replay.Runner(pkg, make_backend("hip"), eps=1e-4).replay(cap, full_graph=False)
    assert tight["failed"] == [], tight
