import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
t0 = time.time()
def lap(msg):
    global t0
    print(f"{msg}: {time.time() - t0:.1f}s", flush=True); t0 = time.time()
import numpy as np
import hot_util as hu
from ggml_util import F32
lap("imports")
cfg = hu.hot.tiny(hu.L, linear_type=F32, embed_type=F32)
cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
m = hu.Model("oracle", cfg); lap("oracle create")
rng = np.random.default_rng(3)
for i in range(8): m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
lap("oracle 8 frames"); m.free()
m = hu.Model("hip", cfg); lap("hip create")
for i in range(8): m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist())
lap("hip 8 frames"); m.free()
from moshi_cpp_amd import shard
lap("import shard")
cfg.dep_shard_world, cfg.dep_shard_rank = 1, 0
m = hu.Model("hip", cfg); lap("hip create 2")
sh = shard.DepthShard(hu.L, m.m, cfg, 0, 1, None); lap("DepthShard init (imports torch)")
sh.install()
for i in range(8):
    m.lm_step(rng.integers(0, cfg.card, cfg.n_q - cfg.io_dep_q).tolist()); lap(f"shard frame {i}")
m.free()
