"""CPU, world_size 2, gloo: the N > 1 path of bench.py is "replicas only" (no data-path collective) — what is
distributed is the barrier and the MAX-over-ranks time reduction that turns per-rank timings into the whole-job
frames/s. This spawns two ranks the way the driver launches bench.py and checks that reduction."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dt_local = 2.0 + rank          # rank 1 is the slow one
    dt = bench.reduce_max_time(dist, dt_local, device="cpu")
    fps = bench.whole_job_rate(world, steps=100, seconds=dt)
    if rank == 0:
        print(json.dumps({"dt": dt, "fps": fps, "world": world}))
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_two_rank_time_reduction_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(w)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["world"] == 2 and abs(out["dt"] - 3.0) < 1e-9      # MAX over ranks
    assert abs(out["fps"] - 2 * 100 / 3.0) < 1e-9                 # all ranks' frames / slowest rank's time
