"""Tensor-parallel Temporal stack at moshika widths: microseconds per stack pass (32 layers, 65 segment graphs, 64 all-reduces of 16 KB).
    python tests/microbench/tp_bench.py                                   # one rank: the segment-graph form of the unsplit stack
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tests/microbench/tp_bench.py [--backend gloo --device 0]
"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
ap = argparse.ArgumentParser(); ap.add_argument("--backend", default="nccl"); ap.add_argument("--device", type=int, default=None); ap.add_argument("--passes", type=int, default=40)
args = ap.parse_args()
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
dist = None
if world > 1:
    import torch, torch.distributed as dist
    dev = int(os.environ.get("LOCAL_RANK", 0)) if args.device is None else args.device
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend)
import hot_util as hu
from moshi_cpp_amd import shard
L = hu.L
cfg = hu.hot.moshika(L); cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0; cfg.dep_q = 0; cfg.n_q = 16
cfg.tp_world, cfg.tp_rank = world, rank
L.ggml_backend_load_all()
dev = (int(os.environ.get("LOCAL_RANK", 0)) if args.device is None else args.device) if world > 1 else 0
be = L.ggml_backend_init_by_name(f"ROCm{dev}".encode(), None)
m = L.moshi_hot_create(be, C.byref(cfg), 0)
device = None
if world > 1:
    import torch
    device = torch.device("cuda", dev)
# nccl: RCCL from inside the harness on the backend's stream; gloo (single-GPU dry runs: --backend gloo --device 0): the partial aliased as a device tensor,
# reduced through torch.distributed with the backend synchronised around it
if world > 1 and args.backend == "gloo":
    tp = shard.TemporalTP(L, m, cfg, rank, world, dist, staged_device=device, backend=be)
else:
    tp = shard.TemporalTP(L, m, cfg, rank, world, dist, device=device)
x = (np.random.default_rng(0).standard_normal(cfg.dim) * 4).astype(np.float32)
for _ in range(5):
    tp.stack(x)
L.ggml_backend_synchronize(be)
t0 = time.perf_counter()
for _ in range(args.passes):
    tp.stack(x)
L.ggml_backend_synchronize(be)
dt = (time.perf_counter() - t0) / args.passes
if rank == 0:
    print(f"tensor-parallel Temporal stack, {world} rank(s), backend {args.backend if world > 1 else '-'}: {dt * 1e6:.0f} us per pass ({tp.reductions // (args.passes + 5)} all-reduces), "
          f"Temporal weight bytes on this rank {L.moshi_hot_weight_bytes(m, 0) / 1e9:.2f} GB")
if dist is not None:
    dist.barrier(); dist.destroy_process_group()
