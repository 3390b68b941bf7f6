"""bench.py's output contract: ONE JSON line on stdout with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline` and, at N = 1, `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_parses_its_arguments_and_hashes_its_sources():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--serial" in out.stdout and "--shard" in out.stdout and "--gpus" in out.stdout
    sys.path.insert(0, ROOT)
    import bench
    sha = bench.source_sha()
    assert len(sha) == 16 and sha == bench.source_sha()
    with open(os.path.join(ROOT, "profiles", bench.PMC_FILE)) as f:
        pmc = json.load(f)
    assert {"source_sha", "matvec_q4k_kernel"} <= set(pmc) and pmc["matvec_q4k_kernel"]["fetch_bytes_per_launch"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--serial"]])
def test_bench_prints_one_json_line_with_the_contract_fields(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "5", "--no-cpu-baseline", "--no-extras"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "moshika" in d["metric"] and "q4_k" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["frame_loop"].startswith("serial") == bool(extra)
    assert 50 < d["value"] < 2000 and abs(d["ms_per_step"] * d["value"] - 1000) < 5
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["kernel"] == "matvec_q4k_kernel"
    # 97 launches of the LDS-tile family (out_proj, linear_in, linear_out per Temporal layer + the text head); the 32 in_proj launches carry the layer's
    # attention as their tail and are listed as their own variant (inproj_attn_kernel); the Depth transformer's 208 mat-vecs run inside one persistent chain launch
    assert r["launches_per_frame"] == 97 and 0.05 < r["frac"] < 1.0
    merged = [v for k, v in d["roofline_by_variant"].items() if k.startswith("merged")]
    assert len(merged) == 1 and merged[0]["launches_per_frame"] == 32 and 25e6 < merged[0]["algorithmic_bytes_per_launch"] < 32e6
    chain = [v for k, v in d["roofline_by_variant"].items() if k.startswith("persistent_chain")]
    # the Depth step program (375 MB) - and, in the serial loop, where the codec graphs run on the LM's own stream, the two Mimi transformer programs (100 MB each)
    if extra:
        assert len(chain) == 1 and chain[0]["launches_per_frame"] == 3 and 150e6 < chain[0]["algorithmic_bytes_per_launch"] < 250e6
    else:
        assert len(chain) == 1 and chain[0]["launches_per_frame"] == 1 and 300e6 < chain[0]["algorithmic_bytes_per_launch"] < 450e6
    assert 0 <= d["n_fill_avg"] <= 3000 and d["ranks_reporting"] == 1 and d["rccl_world_size"] is None
    if not extra:
        assert d["serial_loop"]["value"] < d["value"]          # the two-stream run-ahead loop beats the serial one on the same model
        assert d["value_serial"] == d["serial_loop"]["value"]  # what an unchanged reference tool gets, at the top level
        # the serial legs run a fixed, warmed-up 125 frames whatever --steps is (6 here): the driver's --steps 20 must not shorten what value_serial rests on
        assert d["serial_loop"]["steps"] == 125 and d["serial_loop"]["handles"] == 1 and d["serial_loop"]["warmup"] >= 30


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["depth", "temporal"])
def test_two_rank_bench_line_on_one_gpu_over_a_gloo_control_plane(shard):
    """The N > 1 JSON the driver would read from an 8-GPU node, exercised on the one GPU there is: two ranks launched the way the driver launches them
    (python -m torch.distributed.run, one process per rank), both on device 0, gloo as the control plane and as the shard's transport (--dist-backend gloo;
    with nccl the same C loop calls RCCL). Strong scaling (ONE stream over all ranks), a `shard` object, both ranks reporting."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--device", "0", "--shard", shard, "--steps", "4", "--warmup", "2",
           "--no-cpu-baseline", "--no-extras"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "strong" and d["unit"] == "frames/s" and d["value"] > 5
    assert d["ranks_reporting"] == 2 and d["control_backend"] == "gloo" and d["rccl_world_size"] is None
    sh = d["shard"]
    assert sh["ranks"] == 2
    if shard == "depth":
        assert sh["messages_per_frame"] == 9 and sh["message_bytes"] > 24 * 1024 and 150e6 < sh["depth_weight_bytes_this_rank"] < 200e6
    else:
        assert sh["all_reduces_per_frame"] == 64 and sh["all_reduce_bytes"] == 16384 and sh["stack_passes"] >= 6
        assert 1.7e9 < sh["temporal_weight_bytes_this_rank"] < 2.2e9
