"""CPU, build container only (needs /root/reference): the reference's OWN graph builders - src/moshi/modules/transformer.h,
rope.h, gating.h, conv.h, seanet.h, src/moshi/quantization/*.h, src/moshi/models/compression.h, src/torch.h, src/context.h,
compiled from where they lie, nothing copied - are linked against this repository's ggml surface (include/*.h +
libggml-mi355x.so) and RUN next to the moshi_hot driver over the same weight tensors, both on the CPU oracle:

  * Temporal transformer stack, 40 steps over a ring of 24 (wraps): bit-identical outputs every step;
  * Mimi decode (codes -> 1920 samples) and encode (1920 samples -> codes), 6 frames: bit-identical samples, identical codes.

This is the integration INTEGRATION.md describes (libmoshi code calling the unchanged ggml C API), and it pins the graph
construction of moshi_hot.cpp - the driver the GPU parity tests and bench.py use - to the reference's real code rather than to
a reading of it. The programs live in tests/ref_link/."""
import os
import shutil
import subprocess
import tempfile

import pytest

import ggml_util as gu

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def build_and_run(src, args):
    lib_dir = os.path.join(gu.ROOT, "moshi.cpp_amd")
    oracle = os.path.join(gu.ROOT, "oracle", "liboracle.so")
    gu.lib()                      # makes sure the product library is built and loadable
    gu.attach_oracle()            # builds oracle/liboracle.so if needed
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "prog")
        cc = subprocess.run(["g++", "-std=c++20", "-O1", "-w", "-I" + os.path.join(gu.ROOT, "include"), "-I" + os.path.join(REF, "include"), "-I" + REF,
                             os.path.join(HERE, "ref_link", src), "-o", exe, "-L" + lib_dir, "-lmoshi-hot", "-lggml-mi355x", "-ldl", "-Wl,-rpath," + lib_dir],
                            capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-3000:]
        run = subprocess.run([exe, oracle] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    return run.stdout


needs_ref = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None, reason="reference checkout / g++ not available")


@needs_ref
def test_reference_transformer_builders_run_bit_identical_to_driver():
    out = build_and_run("ref_transformer.cpp", [40])
    assert "0 mismatching steps" in out, out


@needs_ref
def test_reference_mimi_builders_run_bit_identical_to_driver():
    out = build_and_run("ref_mimi.cpp", [6])
    assert "0 mismatches" in out, out
