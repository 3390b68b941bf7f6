"""CPU: the tensor-parallel Temporal stack (column-split in_proj / linear_in, row-split out_proj / linear_out on 256-value
boundaries, heads and KV ring by rank, two all-reduces per layer) on the oracle:
  * with ONE rank its segment graphs are the unsplit stack: bit-identical to the chained layer probes over the regular weights;
  * with TWO gloo ranks, each holding half of every matrix (cut from the same synthetic matrices), the stack output over several stream positions
    (ring fills, RoPE advances) equals the one-rank output to float-summation noise - the integer block dots are unchanged, only the last float
    sums are split - for F32 and for Q4_K weights."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

import hot_util as hu
from ggml_util import F32, Q4_K, Q4_0

L = hu.L
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tp_config(lt):
    cfg = hu.hot.tiny(L, linear_type=lt, embed_type=Q4_0 if lt == Q4_K else F32)
    cfg.ffn_hidden = 1024                       # F / 2 must fall on a 256 boundary
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    return cfg


def test_one_rank_segments_are_the_unsplit_stack():
    import parity_probe as pp
    from moshi_cpp_amd import shard
    cfg = tp_config(Q4_K)
    cfg.tp_world, cfg.tp_rank = 1, 0
    m = hu.Model("oracle", cfg)
    tp = shard.TemporalTP(L, m.m, cfg, 0, 1, None)
    rng = np.random.default_rng(0)
    for pos in range(5):
        x = (rng.standard_normal(cfg.dim) * 3).astype(np.float32)
        got = tp.stack(x)
        want = x
        for layer in range(cfg.num_layers):
            _, want = pp.probe(m, 0, layer, 0, want, pos)
        assert np.array_equal(got, want), f"position {pos}"
    m.free()


WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import hot_util as hu
    from ggml_util import F32, Q4_K, Q4_0
    from moshi_cpp_amd import shard
    import test_temporal_tp_cpu as t
    L = hu.L
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    lt = Q4_K if os.environ["TP_TYPE"] == "q4_k" else F32
    cfg = t.tp_config(lt)
    cfg.tp_world, cfg.tp_rank = world, rank
    m = hu.Model("oracle", cfg)
    tp = shard.TemporalTP(L, m.m, cfg, rank, world, dist)
    ref = None
    if rank == 0:
        c1 = t.tp_config(lt); c1.tp_world, c1.tp_rank = 1, 0
        m1 = hu.Model("oracle", c1)
        ref = shard.TemporalTP(L, m1.m, c1, 0, 1, None)
    rng = np.random.default_rng(0)
    errs = []
    for pos in range(6):
        x = (rng.standard_normal(cfg.dim) * 3).astype(np.float32)
        got = tp.stack(x)
        if rank == 0:
            want = ref.stack(x)
            errs.append(float(np.abs(got - want).max() / np.abs(want).max()))
    if rank == 0:
        wb = L.moshi_hot_weight_bytes(m.m, 0)
        print(json.dumps({"errs": errs, "reductions": tp.reductions, "bytes": wb, "bytes_full": L.moshi_hot_weight_bytes(m1.m, 0)}))
    dist.barrier()
    dist.destroy_process_group()
''') % (ROOT, ROOT)


def run_two_ranks(tmp_path, kind, port):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", TP_TYPE=kind, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port), str(w)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_rank_tp_stack_f32_matches_one_rank(tmp_path):
    out = run_two_ranks(tmp_path, "f32", 29631)
    assert max(out["errs"]) < 1e-5, out
    assert out["reductions"] == 6 * 2 * 2            # positions x layers x two all-reduces per layer
    assert out["bytes"] < 0.62 * out["bytes_full"]   # a rank holds half of every Temporal matrix (+ the replicated head)


def test_two_rank_tp_stack_q4k_matches_one_rank(tmp_path):
    out = run_two_ranks(tmp_path, "q4_k", 29632)
    # the activation rounding (Q8_K, BF16 ring) sits behind the summed partials: a 1e-7 difference in a sum can flip a rounded value, as anywhere else
    assert np.median(out["errs"]) < 1e-5 and max(out["errs"]) < 1e-2, out


FRAME_WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import hot_util as hu
    from ggml_util import F32
    from moshi_cpp_amd import shard
    import test_temporal_tp_cpu as t
    L = hu.L
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = t.tp_config(F32)
    cfg.tp_world, cfg.tp_rank = world, rank
    m = hu.Model("oracle", cfg)
    tp = shard.TemporalTP(L, m.m, cfg, rank, world, dist)
    steps = 8
    if rank != 0:
        served = tp.serve()
        assert served == steps, served
    else:
        tp.install()
        rng = np.random.default_rng(5)
        got = [m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist()) for _ in range(steps)]
        logits = m.read("text_logits", cfg.text_card).copy()
        tp.stop_workers()
        c1 = t.tp_config(F32)                     # the unsplit model through the ordinary Temporal graph
        m1 = hu.Model("oracle", c1)
        rng = np.random.default_rng(5)
        want = [m1.lm_step(rng.integers(0, c1.card, c1.n_q - c1.io_dep_q).tolist()) for _ in range(steps)]
        ref_logits = m1.read("text_logits", c1.text_card)
        print(json.dumps({"same_tokens": got == want, "logit_err": float(np.abs(logits - ref_logits).max() / np.abs(ref_logits).max()),
                          "reductions": tp.reductions, "frames": int(L.moshi_hot_tp_frames(m.m))}))
    dist.barrier()
    dist.destroy_process_group()
''') % (ROOT, ROOT)


def test_two_rank_tensor_parallel_frames_give_the_one_rank_tokens(tmp_path):
    # (f2) as a frame mode. Rank 0 runs whole LM steps whose Temporal half is embedding sum -> broadcast -> 2 L + 1 segments with all-reduces ->
    # head graph, rank 1 serves; the greedy tokens of 8 frames must be the unsplit model's, the last text logits within the stack test's bar.
    w = tmp_path / "frame_worker.py"
    w.write_text(FRAME_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29633", str(w)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["same_tokens"], out
    assert out["logit_err"] < 1e-5, out
    assert out["frames"] == 8 and out["reductions"] == 8 * 2 * 2, out


def test_one_rank_tensor_parallel_frames_are_the_ordinary_frames():
    # frame mode with ONE rank and no transport: the segment graphs are the unsplit stack, so tokens and logits are the Temporal graph's, bit for bit
    from moshi_cpp_amd import shard
    cfg = tp_config(Q4_K)
    cfg.tp_world, cfg.tp_rank = 1, 0
    m = hu.Model("oracle", cfg)
    tp = shard.TemporalTP(L, m.m, cfg, 0, 1, None)
    tp.install()
    m1 = hu.Model("oracle", tp_config(Q4_K))
    rng = np.random.default_rng(2)
    for i in range(6):
        ia = rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist()
        assert m.lm_step(ia) == m1.lm_step(ia), f"frame {i}"
        assert np.array_equal(m.read("text_logits", cfg.text_card), m1.read("text_logits", cfg.text_card)), f"frame {i}"
    m.free(); m1.free()
