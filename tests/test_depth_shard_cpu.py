"""CPU: the Depth-transformer codebook shard (include/moshi_hot.h "Depth codebook shard", moshi.cpp_amd/shard.py) against the
chained single-graph Depth loop it replaces, on the oracle: per-step graphs + K/V-row messages must leave BIT-IDENTICAL tokens (and logits
through them) - first inside one process (world 1: no transport), then over two gloo ranks where rank 1 holds only the odd steps' weights."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np

import hot_util as hu
from ggml_util import Q4_K, Q4_0

L = hu.L
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def frames(m, cfg, n, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        r = m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
        out.append((r, m.last_raw()))
    return out


def test_per_step_graphs_equal_the_chained_graph():
    from moshi_cpp_amd import shard
    for mk in (lambda: hu.hot.tiny(L, linear_type=Q4_K, embed_type=Q4_0), lambda: hu.hot.tiny_personaplex(L, linear_type=Q4_K)):
        cfg = mk()
        cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
        ref = hu.Model("oracle", cfg)
        a = frames(ref, cfg, 8)
        ref.free()
        cfg.dep_shard_world, cfg.dep_shard_rank = 1, 0
        m = hu.Model("oracle", cfg)
        sh = shard.DepthShard(L, m.m, cfg, 0, 1, None)
        sh.install()
        b = frames(m, cfg, 8)
        m.free()
        assert a == b


WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import hot_util as hu
    from ggml_util import Q4_K, Q4_0
    from moshi_cpp_amd import shard
    L = hu.L
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    def config():
        cfg = hu.hot.tiny_personaplex(L, linear_type=Q4_K) if os.environ.get("SHARD_MODEL") == "personaplex" else hu.hot.tiny(L, linear_type=Q4_K, embed_type=Q4_0, dep_q=4, n_q=8)
        cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
        return cfg
    cfg = config()
    cfg.dep_shard_world, cfg.dep_shard_rank, cfg.depth_only = world, rank, int(rank != 0)
    m = hu.Model("oracle", cfg)
    sh = shard.DepthShard(L, m.m, cfg, rank, world, dist)
    if rank == 0:
        sh.install()
        rng = np.random.default_rng(3)
        got = []
        for _ in range(6):
            r = m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
            got.append([list(r), [m.last_raw()[0], list(m.last_raw()[1])]])
        sh.stop_workers()
        ref_cfg = config()
        ref = hu.Model("oracle", ref_cfg)
        rng = np.random.default_rng(3)
        want = []
        for _ in range(6):
            r = ref.lm_step(rng.integers(0, ref_cfg.card, ref_cfg.n_q - ref_cfg.io_dep_q).tolist())
            want.append([list(r), [ref.last_raw()[0], list(ref.last_raw()[1])]])
        w0 = sum(L.moshi_hot_weight_bytes(m.m, p) for p in (1,)); w1 = sum(L.moshi_hot_weight_bytes(ref.m, p) for p in (1,))
        print(json.dumps({"equal": got == want, "hops": sh.hops, "depth_bytes_sharded": w0, "depth_bytes_full": w1, "tokens": got[-1]}))
    else:
        served = sh.serve()
        assert served == 6, served
        assert L.moshi_hot_weight_bytes(m.m, 0) == 0           # no Temporal stack on a Depth-only rank
    dist.barrier()
    dist.destroy_process_group()
''') % (ROOT, ROOT)


def run_two_ranks(tmp_path, model, port):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", SHARD_MODEL=model, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port), str(w)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_two_rank_gloo_shard_is_bit_identical_to_one_rank(tmp_path):
    out = run_two_ranks(tmp_path, "tiny", 29621)
    assert out["equal"], out
    assert out["hops"] == 6 * (1 + 4) + 1                      # per frame: 1 transformer_out broadcast + dep_q step messages; + the stop message
    assert out["depth_bytes_sharded"] < 0.62 * out["depth_bytes_full"]   # rank 0 holds steps 0 and 2 of 4 (+ nothing of the others)


def test_two_rank_gloo_shard_personaplex_ring_wraps(tmp_path):
    # 16 steps over a ring of 8: slots are rewritten inside a frame, on every replica, in step order
    out = run_two_ranks(tmp_path, "personaplex", 29622)
    assert out["equal"], out
