"""CPU, build container only (needs /root/reference): the reference's OWN graph builders - src/moshi/modules/transformer.h,
rope.h, gating.h, conv.h, seanet.h, src/moshi/quantization/*.h, src/moshi/models/compression.h, src/torch.h, src/context.h,
compiled from where they lie, nothing copied - are linked against this repository's ggml surface (include/*.h +
libggml-mi355x.so) and RUN next to the moshi_hot driver over the same weight tensors, both on the CPU oracle:

  * Temporal transformer stack, 40 steps over a ring of 24 (wraps): bit-identical outputs every step;
  * Mimi decode (codes -> 1920 samples) and encode (1920 samples -> codes), 6 frames: bit-identical samples, identical codes;
  * (round 6) the LM frame driver itself - src/moshi/models/lm.h moshi_lmgen_step with its delay ring, embedding sums, text head and the chained
    depformer graph, lm_utils.h, utils/sampling.h - for moshika (dep_q 8) and PersonaPlex (dep_q 16 chained, 8 exposed), greedy and in the reference's
    sampling mode on the same host rand() stream: identical return flags and delayed tokens over 40 frames (tests/ref_link/ref_lm.cpp).

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


def build_and_run(src, args, runs=None):
    lib_dir = os.path.join(gu.ROOT, "moshi.cpp_amd")
    oracle = os.path.join(gu.ROOT, "oracle", "liboracle.so")
    gu.lib()                      # makes sure the product library is built and loadable
    gu.attach_oracle()            # builds oracle/liboracle.so if needed
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "prog")
        # (stub/: an EMPTY sentencepiece_processor.h - include/moshi/moshi.h pulls that third-party header in for its tokenizer API; ref_lm.cpp needs
        # moshi.h only for the plain struct Entry)
        cc = subprocess.run(["g++", "-std=c++20", "-O1", "-w", "-I" + os.path.join(HERE, "ref_link", "stub"), "-I" + os.path.join(gu.ROOT, "include"), "-I" + os.path.join(REF, "include"), "-I" + REF,
                             os.path.join(HERE, "ref_link", src), "-o", exe, "-L" + lib_dir, "-lmoshi-hot", "-lggml-mi355x", "-ldl", "-Wl,-rpath," + lib_dir],
                            capture_output=True, text=True)
        assert cc.returncode == 0, cc.stderr[-3000:]
        outs = []
        for a in (runs or [args]):
            run = subprocess.run([exe, oracle] + [str(x) for x in a], capture_output=True, text=True, timeout=900)
            assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
            outs.append(run.stdout)
    return outs if runs else outs[0]


needs_ref = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None, reason="reference checkout / g++ not available")


@needs_ref
def test_reference_transformer_builders_run_bit_identical_to_driver():
    out = build_and_run("ref_transformer.cpp", [40])
    assert "0 mismatching steps" in out, out


@needs_ref
def test_reference_mimi_builders_run_bit_identical_to_driver():
    out = build_and_run("ref_mimi.cpp", [6])
    assert "0 mismatches" in out, out


@needs_ref
def test_reference_lm_frame_driver_runs_token_identical_to_driver():
    """H0 / H1 / H11 / H12 pinned by the reference's real code: moshi_lmgen_step (lm.h:778-979) beside moshi_hot_lm_step."""
    outs = build_and_run("ref_lm.cpp", None, runs=[[40, pp, smp] for pp in (0, 1) for smp in (0, 1)])
    for (pp, smp), out in zip([(pp, smp) for pp in (0, 1) for smp in (0, 1)], outs):
        assert ("personaplex" if pp else "moshika") in out and ("sampled" if smp else "greedy") in out, out
        assert "39 with output" in out and " 0 mismatching frames" in out, out
        distinct = int(out.split("with output, ")[1].split(" distinct")[0])
        assert distinct > 30, out          # the runs are not degenerate: dozens of different token values went through the delay ring
