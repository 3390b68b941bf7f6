#!/usr/bin/env python3
"""bench.py — audio frames/s of the moshika-7B q4_k streaming decode hot path on MI355X.

A "step" is one iteration of the reference's `moshi-sts --bench` loop (tools/moshi-sts.cpp:770-808):
mimi_encode(1920 samples) -> Temporal transformer step -> 8 chained Depth steps -> mimi_decode, single
stream, greedy sampling, synthetic weights of the shapes/types a `-q q4_k` load of tools/moshi-config.json
produces (no model files can reach the box), all weights / KV rings / codec state resident in HBM before
the timed region starts.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N`: one process per GPU, each
running an independent stream replica ("replicas only"); barrier + device
synchronise on both sides of the timed region, MAX over ranks, rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # the CPU baseline's OpenMP team must not spin between mat-vecs
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle, load_package  # noqa: E402

PMC_FILE = "r06_pmc_fetch_size.json"   # rocprofv3 --pmc FETCH_SIZE pass of this command (tests/microbench/take_profiles.sh), stamped with the kernel sources' hash
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
SERIAL_FRAMES, SERIAL_WARMUP = 125, 30   # the serial (unchanged-tool) legs: the reference's own --bench length (tools/moshi-sts.cpp:770-808), outside the timed region


def reduce_max_time(dist, dt, device="cuda", group=None):
    """MAX over ranks of the timed-region duration (every rank must have finished its K steps)."""
    if dist is None:
        return dt
    import torch
    if dist.get_backend(group) == "gloo":
        device = "cpu"
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
    return float(tt.item())


def whole_job_rate(world, steps, seconds):
    """Replicas: every rank processed `steps` frames of its own stream; the job rate is all of them over the slowest rank."""
    return world * steps / seconds


def shard_hop_latency(shard, dist, group, L, be, n=200):
    """average time of one step-message broadcast (stream-ordered, synchronised once at the end), owners rotating as in a frame; collective: every rank calls it"""
    if shard.world == 1:
        return 0.0
    for k in range(8):
        shard.hop(k % shard.world)
    L.ggml_backend_synchronize(be)
    t0 = time.perf_counter()
    for k in range(n):
        shard.hop(k % shard.world)
    L.ggml_backend_synchronize(be)
    return round(1e6 * (time.perf_counter() - t0) / n, 2)


def source_sha():
    """hash of the kernel / backend sources the library was built from: profile files under profiles/ carry it, and a counter file taken from
    other sources is refused (its numbers would describe another kernel)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "moshi.cpp_amd", "csrc", "*"))):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def device_sync():
    hip = C.CDLL("libamdhip64.so")
    hip.hipDeviceSynchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=125)    # the reference bench length (README.md:353)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--context-fill", type=int, default=0, help="start the Temporal KV ring at this fill level")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=32, help="OpenMP threads of the CPU baseline (capped by the affinity mask)")
    ap.add_argument("--quant", default="q4_k", choices=["q4_k", "q8_0", "q4_0"], help="linear weight type (the headline metric is q4_k)")
    ap.add_argument("--model", default="moshika", choices=["moshika", "personaplex", "tts_like", "stt_like"],
                    help="moshika = BASELINE.json's metric config (default); personaplex = configs[4]: 16 chained Depth steps, run with --context 2000; "
                         "tts_like / stt_like = the shapes of configs[1] / configs[2] (hot.py), extra lines only")
    ap.add_argument("--context", type=int, default=0, help="Temporal ring capacity (-c of the tools); 0 = the config's 3000")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="control-plane backend for N > 1 (nccl = RCCL; gloo for single-GPU dry runs)")
    ap.add_argument("--device", type=int, default=None, help="override the device index (default LOCAL_RANK); only for dry runs of the N > 1 path on one GPU")
    ap.add_argument("--backend-flags", type=int, default=0, help="ggml_backend_mi355x_set_flags bits (2 = no hipGraph: use under rocprofv3)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra lines (sampled mode, long context, PersonaPlex) that ride in the JSON's `extras` object")
    ap.add_argument("--sampled", action="store_true", help="the reference's own --bench sampling mode (tools/moshi-sts.cpp:106-107: Depth temperature 0.8, "
                    "text temperature 0.7, top-k 250 / 25) instead of greedy; host rand() noise uploaded per compute as src/context.h:456-480 does")
    ap.add_argument("--serial", action="store_true", help="the frame loop on ONE command stream (encode -> LM -> decode of the same frame, each waited for). Default: the same "
                    "graphs software-pipelined over two HIP streams (LM of frame k beside decode of k - 1 and encode of k + 1; identical tokens and PCM)")
    ap.add_argument("--shard", default="none", choices=["none", "depth", "temporal"],
                    help="depth: ONE stream, the Depth transformer's per-codebook weight sets sharded over the ranks (SURVEY.md 8e: step k on rank k %% N, "
                         "K/V rows + token broadcast per step over RCCL); temporal: ONE stream, every Temporal layer split over the ranks (SURVEY.md 8f.2: heads / "
                         "KV ring / FFN units by rank, 2 all-reduces of F32[dim] per layer over RCCL; embeddings, text head, Depth chain and codec on rank 0); "
                         "both strong scaling. Default: independent stream replicas (weak scaling)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the documented launcher as a CHILD (this process has not touched the GPU and never
        # will) and leave with its exit code; the workers re-enter main() with RANK / LOCAL_RANK / WORLD_SIZE set
        import subprocess
        port = os.environ.get("MASTER_PORT", "29533")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    dist = None
    rccl_ranks = None
    emit = lambda line: print(line, flush=True)
    if world > 1:
        import torch
        import torch.distributed as dist
        dev_index = local_rank if args.device is None else args.device
        torch.cuda.set_device(dev_index)
        # One backend for the whole job, decided collectively: every rank first joins a host-side gloo group; RCCL is then tried by
        # all ranks and the outcome is agreed on over gloo (MIN of the success flags), so no rank is ever left alone in an RCCL barrier.
        # (stdout carries ONE JSON line: the gloo transport's connection banners, printed from C++ on fd 1, go to stderr)
        sys.stdout.flush()
        _fd1 = os.dup(1)
        os.dup2(2, 1)
        emit = lambda line: os.write(_fd1, (line + "\n").encode())
        dist.init_process_group("gloo")
        ctl = dist.group.WORLD
        if args.dist_backend == "nccl":
            ok = 1
            try:
                ctl = dist.new_group(backend="nccl", device_id=torch.device("cuda", dev_index))   # RCCL over xGMI
                probe = torch.ones(1, device=f"cuda:{dev_index}")
                dist.all_reduce(probe, group=ctl)                                                    # forces communicator creation now
                torch.cuda.synchronize()
                ok = int(probe.item() == world)
                rccl_ranks = int(probe.item())      # every rank contributed a one over RCCL: the communicator really spans `world` ranks
            except Exception as e:
                print(f"bench.py: RCCL init failed on rank {rank}: {e}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)                                              # over gloo
            if int(flag.item()) == 0:
                raise SystemExit("bench.py: RCCL could not be initialised on every rank (see above); refusing to time a job on a mixed control plane. "
                                 "Use --dist-backend gloo for a host-side dry run.")
        dist_group = ctl
    else:
        dist_group = None

    pkg = load_package()
    L = pkg.load()
    from moshi_cpp_amd import hot

    L.ggml_backend_load_all()
    dev_index = local_rank if args.device is None else args.device
    be = L.ggml_backend_init_by_name(f"ROCm{dev_index}".encode(), None)
    if not be:
        raise SystemExit("bench.py: no MI355X device 'ROCm%d' (the hot path has no CPU fallback)" % dev_index)
    dev = L.ggml_backend_get_device(be)
    dev_desc = L.ggml_backend_dev_description(dev).decode()

    if args.backend_flags:
        L.ggml_backend_mi355x_set_flags(be, args.backend_flags)
    cfg = {"moshika": hot.moshika, "personaplex": hot.personaplex, "tts_like": hot.tts_like, "stt_like": hot.stt_like}[args.model](L)
    if args.context:
        cfg.context = args.context
    if args.model == "tts_like":
        args.quant = "q8_0"
    elif args.quant != "q4_k":
        cfg.linear_type = {"q8_0": 8, "q4_0": 2}[args.quant]   # ggml_type ids
    if args.sampled:
        cfg.temp, cfg.temp_text = 0.8, 0.7
    pipelined = not args.serial and args.shard == "none"
    cfg.codec_stream = int(pipelined)
    # run-ahead: the LM stream never waits for the host (include/moshi_hot.h); tts / stt shapes (text hook, no Depth) overlap their codec half only
    cfg.chain_depth = 2 if pipelined and args.model in ("moshika", "personaplex") else 0
    shard = None
    if args.shard != "none" and args.model not in ("moshika", "personaplex"):
        raise SystemExit("--shard %s: moshika / personaplex only" % args.shard)
    if args.shard == "depth":
        cfg.dep_shard_world, cfg.dep_shard_rank, cfg.depth_only = world, rank, int(rank != 0)
        if rank != 0:
            cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    if args.shard == "temporal":
        cfg.tp_world, cfg.tp_rank = world, rank
        if rank != 0:
            cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    t0 = time.time()
    m = L.moshi_hot_create(be, C.byref(cfg), 0)
    t_load = time.time() - t0
    if args.shard == "depth":
        from moshi_cpp_amd import shard as shard_mod
        import torch
        # the hop loop and the RCCL broadcasts run behind the C-ABI (include/moshi_hot.h "the whole sharded frame"); on a gloo control plane (dry runs on one
        # GPU) the same C loop calls back into torch.distributed over host memory
        use_rccl = dist is None or dist.get_backend(dist_group) != "gloo"
        shard = shard_mod.DepthShard(L, m, cfg, rank, world, dist, group=dist_group, device=torch.device("cuda", dev_index) if use_rccl else None,
                                     staged_device=None if use_rccl else torch.device("cuda", dev_index), backend=be)
        if rank == 0:
            shard.install()
    if args.shard == "temporal":
        from moshi_cpp_amd import shard as shard_mod
        import torch
        # the frame's Temporal half = embedding-sum graph -> broadcast of x -> 2 L + 1 segment graphs with 2 L all-reduces -> head graph, all behind the C-ABI
        # (include/moshi_hot.h "FRAME mode"); workers sit in moshi_hot_tp_serve
        use_rccl = dist is None or dist.get_backend(dist_group) != "gloo"
        shard = shard_mod.TemporalTP(L, m, cfg, rank, world, dist, group=dist_group, device=torch.device("cuda", dev_index) if use_rccl else None,
                                     staged_device=None if use_rccl else torch.device("cuda", dev_index), backend=be)
        if rank == 0:
            shard.install()
    if args.context_fill:
        L.moshi_hot_set_context_fill(m, args.context_fill)

    pcm = np.zeros(1920, np.float32)     # --bench feeds silence (tools/moshi-sts.cpp:764-767)
    out = np.zeros(1920, np.float32)
    txt = C.c_int32()
    aud = (C.c_int32 * 32)()

    if args.model == "tts_like":      # moshi-tts loop (tools/moshi-tts.cpp:757-820): LM step (text from the state machine: here a fixed hook) + Mimi decode
        rng = np.random.default_rng(1)
        cond_sum = (rng.standard_normal(cfg.dim) * 0.1).astype(np.float32)
        cond_cross = rng.standard_normal((cfg.cross_len, cfg.dim)).astype(np.float32)
        L.moshi_hot_set_conditions(m, cond_sum.ctypes.data, cond_cross.ctypes.data)
        hook = hot.TEXT_HOOK(lambda user, offset, sampled: int((offset * 13) % cfg.text_card))
        L.moshi_hot_set_text_hook(m, C.cast(hook, C.c_void_p), None)
        aud64 = (C.c_int32 * 64)()
        none_in = (C.c_int32 * 1)()

        def frame_serial():
            if L.moshi_hot_lm_step_n(m, none_in, 0, C.byref(txt), aud64, None):
                L.moshi_hot_mimi_decode(m, aud64, out.ctypes.data)
            return 1
        frame = frame_serial
        if pipelined:          # decode of frame k - 1 beside the LM step of frame k
            L.moshi_hot_sts_pipeline_begin(m, None)

            def frame():
                return L.moshi_hot_sts_pipeline_frame(m, None, C.byref(txt), aud64, out.ctypes.data)
    elif args.model == "stt_like":    # moshi-stt loop (tools/moshi-stt.cpp:552-719): Mimi encode (32 levels) + LM step with the VAD head
        codes = (C.c_int32 * 64)()
        aud64 = (C.c_int32 * 64)()
        vad = C.c_float()

        def frame_serial():
            L.moshi_hot_mimi_encode(m, pcm.ctypes.data, codes)
            return L.moshi_hot_lm_step_n(m, codes, cfg.n_q, C.byref(txt), aud64, C.byref(vad))
        frame = frame_serial
        if pipelined:          # encode of frame k + 1 beside the LM step (+ VAD head) of frame k
            L.moshi_hot_sts_pipeline_begin(m, pcm.ctypes.data)

            def frame():
                return L.moshi_hot_sts_pipeline_frame(m, pcm.ctypes.data, C.byref(txt), aud64, out.ctypes.data)
    else:
        def frame_serial():
            return L.moshi_hot_sts_frame(m, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)
        frame = frame_serial
        if pipelined:
            # call k: LM step of frame k on the backend's stream; decode of frame k - 1 and encode of frame k + 1 on the codec stream
            # (include/moshi_hot.h "software-pipelined"). One encode, one LM step and one decode per call; results read back before it returns.
            L.moshi_hot_sts_pipeline_begin(m, pcm.ctypes.data)

            def frame():
                return L.moshi_hot_sts_pipeline_frame(m, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)

    def barrier():
        L.ggml_backend_synchronize(be)
        device_sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier(group=dist_group)

    if shard is not None and rank != 0:
        # a Depth-only rank: serve the owner's frames (warm-up, timed region, phase pass) until it says stop; the owner's clock is the job's
        shard.serve()
        if args.shard == "depth":
            hop_us = shard_hop_latency(shard, dist, dist_group, L, be)
        reduce_max_time(dist, 0.0, group=dist_group)
        L.moshi_hot_free(m)
        dist.barrier()
        dist.destroy_process_group()
        return
    for _ in range(args.warmup):
        frame()
    tokens = []
    if shard is None:
        barrier()
    else:
        L.ggml_backend_synchronize(be); device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
        tokens.append(txt.value)
    if shard is None:
        barrier()
    else:
        L.ggml_backend_synchronize(be); device_sync()      # every step message of the last frame has been received here: the chain is complete
    dt = time.perf_counter() - t0
    if shard is None:
        dt = reduce_max_time(dist, dt, group=dist_group)
    fps = whole_job_rate(world, args.steps, dt) if shard is None else args.steps / dt   # shard: ONE stream over all ranks (strong scaling)
    offset_end = L.moshi_hot_offset(m)

    # algorithmic HBM bytes per frame: every weight byte once + the filled KV slots
    wb = [L.moshi_hot_weight_bytes(m, p) for p in range(5)]
    if args.shard == "temporal" and world == 1:
        wb[0] -= L.moshi_hot_weight_bytes(m, 5)   # a one-rank tensor-parallel model holds the unsplit stack AND its (whole) slices; a frame streams one of the two
    n_fill_avg = min(cfg.context, args.context_fill + args.warmup + args.steps / 2.0)
    kv_bytes = 2 * cfg.num_layers * n_fill_avg * cfg.dim * 2
    emb_rows = (cfg.n_q + 1) * cfg.dim * 18 / 32 + cfg.dep_q * cfg.dep_dim * 18 / 32
    frame_bytes = wb[0] + wb[1] + wb[2] + wb[3] + kv_bytes + emb_rows

    st = pkg.Stats()
    L.ggml_backend_mi355x_get_stats(be, C.byref(st))

    result = {
        "metric": "audio frames/sec (12.5 Hz target) %s %s decode" % ({"moshika": "moshika-7B", "personaplex": "personaplex-7B", "tts_like": "tts-1.6b-shaped",
                                                                        "stt_like": "stt-1b-shaped"}[args.model], args.quant),
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "weak" if shard is None else "strong",
        "vs_baseline": None, "dtype": "q4_K weights x q8_K activations (int8 dot, f32 accumulate); bf16 KV; f32 elsewhere",
        "data": "synthetic",
        "config": {"workload": ({"tts_like": "moshi-tts loop: Temporal step (cross-attention, demux) + %d Depth steps + mimi decode (32 levels), ",
                                 "stt_like": "moshi-stt loop: mimi encode (32 levels) + Temporal step + VAD head (%d Depth steps), "}
                                .get(args.model, "moshi-sts --bench loop: mimi encode + Temporal step + %d Depth steps + mimi decode, ") % cfg.dep_q) +
                               "%s %s, 1 audio stream per GPU, %s, ctx capacity %d" % (args.model, args.quant, "sampled (temp 0.8 / 0.7, top-k 250 / 25)" if args.sampled else "greedy", cfg.context),
                   "context_fill_start": args.context_fill,
                   "frame_loop": ("software-pipelined over 2 HIP streams: the codec half of the neighbouring frame runs beside the LM step (--serial: one after the other)"
                                  if pipelined and not cfg.chain_depth else "") or "software-pipelined over 2 HIP streams with run-ahead: the LM step of frame k is queued behind the one of frame k-1 (sampled tokens reach "
                                 "the next Temporal graph through device memory; the host reads them one step late), mimi decode of frame k-1 and mimi encode of frame k+1 "
                                 "run beside it; same graphs, inputs, states and outputs as the serial loop (--serial), one LM step + one encode + one decode per step" if pipelined else "serial: encode -> LM -> decode of one frame per step",
                   "parallelism": ("Depth codebook shard: step k on rank k %% %d, replicated 8-slot ring, 1 + dep_q broadcasts per frame" % world) if args.shard == "depth"
                                  else ("tensor-parallel Temporal stack over %d rank(s): heads, KV ring and FFN units by rank, 1 broadcast + %d all-reduces of F32[%d] per frame; "
                                        "embeddings, text head, Depth chain and codec on rank 0" % (world, 2 * cfg.num_layers, cfg.dim)) if args.shard == "temporal"
                                  else "independent stream replica per GPU" if world > 1 else "1 GPU",
                   "device": dev_desc},
        "realtime_factor": round(fps / (world if shard is None else 1) / 12.5, 1),
        "frame_bytes": int(frame_bytes),
        "hbm_gbps_whole_frame": round(frame_bytes * fps / (world if shard is None else 1) / 1e9, 1),
        "hbm_frac_whole_frame": round(frame_bytes * fps / (world if shard is None else 1) / 1e9 / HBM_PEAK_GBPS, 4),
        "load_seconds": round(t_load, 1),
        "graph_replays": int(st.graph_replays), "uploads_batched": int(st.uploads_batched),
        "n_fill_avg": round(n_fill_avg, 1),   # live Temporal ring slots averaged over the timed steps (the KV term of frame_bytes)
        # control plane of an N > 1 job: the size of the communicator the barrier / MAX-time reduction ran on, read back from it (a sum of ones over RCCL)
        "rccl_world_size": rccl_ranks if world > 1 and args.dist_backend == "nccl" else None,
        "ranks_reporting": world, "control_backend": (args.dist_backend if world > 1 else None),
        "chained_matvecs_in_last_plan": int(st.chained_matvecs_in_last_plan),
    }

    if rank == 0:
        # where the frame goes (separate pass, synchronising around each phase; not part of the timed region)
        L.moshi_hot_set_timing(m, 1)
        for _ in range(10):
            (frame_serial if pipelined else frame)()
        ph = (C.c_double * 4)()
        L.moshi_hot_get_timing(m, ph)
        L.moshi_hot_set_timing(m, 0)
        result["phase_us"] = {"mimi_encode": round(ph[0], 1), "temporal": round(ph[1], 1), "depth": round(ph[2], 1), "mimi_decode": round(ph[3], 1)}
        # algorithmic bytes of each phase (every weight byte once + the live KV rows) over its synchronised wall time
        pb = {"temporal": wb[0] + kv_bytes + (cfg.n_q + 1) * cfg.dim * 18 / 32, "depth": wb[1] + cfg.dep_q * cfg.dep_dim * 18 / 32, "mimi_encode": wb[2], "mimi_decode": wb[3]}
        result["phase_roofline"] = {k: {"bytes": int(pb[k]), "GB/s": round(pb[k] / (result["phase_us"][k] * 1e-6) / 1e9, 1) if result["phase_us"][k] else None,
                                        "frac": round(pb[k] / (result["phase_us"][k] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if result["phase_us"][k] else None} for k in pb}
        result["source_sha"] = source_sha()
        if pipelined:
            # the same model stepped serially (one frame's encode -> LM -> decode, each waited for): what the pipelining buys
            # Both serial legs run a FIXED number of frames whatever --steps is (the driver's --steps 20 gave 60 ms of timing: BENCH_r05 read 331 where longer runs
            # read 355) and are warmed up for long enough to bring the clocks back after the seconds of host-only work in front of them (model creation).
            ns, nwarm = SERIAL_FRAMES, SERIAL_WARMUP
            for _ in range(nwarm):
                frame_serial()
            L.ggml_backend_synchronize(be); device_sync()       # (this rank only: not a collective)
            t1 = time.perf_counter()
            for _ in range(ns):
                frame_serial()
            L.ggml_backend_synchronize(be); device_sync()
            result["serial_loop"] = {"value": round(ns / (time.perf_counter() - t1), 2), "unit": "frames/s", "steps": ns}
            # the figure an UNCHANGED reference tool gets (tools/moshi-sts.cpp:770-808 keeps its serial call order); `value` needs the restructured caller
            result["value_serial"] = result["serial_loop"]["value"]
            if args.model in ("moshika", "personaplex") and world == 1:
                # ... measured the way such a tool runs: ONE backend handle for every graph (a second instance of the model without the codec stream). The model
                # above keeps its codec graphs on the second command stream, where the codec's transformers stay 40 launches each; on the LM's own handle they are
                # one persistent launch each (DESIGN.md section 4 "Step programs").
                cs, cd = cfg.codec_stream, cfg.chain_depth
                cfg.codec_stream, cfg.chain_depth = 0, 0
                m1 = L.moshi_hot_create(be, C.byref(cfg), 0)
                cfg.codec_stream, cfg.chain_depth = cs, cd
                if args.context_fill:
                    L.moshi_hot_set_context_fill(m1, args.context_fill)
                for _ in range(nwarm):
                    L.moshi_hot_sts_frame(m1, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)
                L.ggml_backend_synchronize(be); device_sync()
                t1 = time.perf_counter()
                for _ in range(ns):
                    L.moshi_hot_sts_frame(m1, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)
                L.ggml_backend_synchronize(be); device_sync()
                result["serial_loop"]["same_model_codec_on_second_stream"] = result["serial_loop"]["value"]
                result["serial_loop"]["value"] = round(ns / (time.perf_counter() - t1), 2)
                result["serial_loop"]["handles"] = 1
                result["serial_loop"]["warmup"] = nwarm
                result["value_serial"] = result["serial_loop"]["value"]
                result["serial_loop"]["frames_stepped_by_the_one_handle_model"] = int(L.moshi_hot_offset(m1))
                L.moshi_hot_free(m1)

    if shard is not None and args.shard == "temporal":
        shard.stop_workers()
        reduce_max_time(dist, dt, group=dist_group)
        result["shard"] = {"kind": "tensor-parallel Temporal stack (SURVEY.md 8f.2)", "ranks": world, "all_reduces_per_frame": 2 * cfg.num_layers,
                           "all_reduce_bytes": int(cfg.dim * 4), "broadcasts_per_frame": 1, "broadcast_bytes": int((cfg.dim + 8) * 4), "transport": shard.transport,
                           "stack_passes": int(L.moshi_hot_tp_frames(m)), "temporal_weight_bytes_this_rank": int(L.moshi_hot_weight_bytes(m, 0))}
        args.no_roofline = True
        args.no_cpu_baseline = True
    elif shard is not None:
        shard.stop_workers()
        hop_us = shard_hop_latency(shard, dist, dist_group, L, be)
        reduce_max_time(dist, dt, group=dist_group)
        result["shard"] = {"kind": "depth codebook shard (SURVEY.md 8e)", "ranks": world, "messages_per_frame": 1 + cfg.dep_q,
                           "message_bytes": int(shard.msg_floats.value * 4), "broadcast_us": hop_us, "transport": shard.transport,
                           "depth_weight_bytes_this_rank": int(L.moshi_hot_weight_bytes(m, 1))}
        args.no_roofline = True          # flag 8 re-plans graphs eagerly; the roofline line belongs to the default (unsharded) run
        args.no_cpu_baseline = True

    if rank == 0 and not args.no_roofline:
        # dominant kernel = matvec_q4k_kernel (3.77 of 4.4 GB per frame). Its launches are timed in situ with HIP
        # start/stop events attached to each dispatch (flag 8: same plan, launched eagerly on the backend stream).
        L.ggml_backend_mi355x_set_flags(be, 8)
        for _ in range(3):
            (frame_serial if pipelined else frame)()
        L.ggml_backend_synchronize(be)
        kp = pkg.KernelProfile()
        L.ggml_backend_mi355x_get_kernel_profile(be, C.byref(kp))
        L.ggml_backend_mi355x_set_flags(be, args.backend_flags)
        # HBM bytes per launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE on this same command,
        # corrected as MI355X_MICROARCH.md prescribes); counters cannot be read from inside the timed process
        traffic, traffic_note = None, "no counter file for these sources"
        try:
            with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
                pmc = json.load(f)
            if pmc.get("source_sha") == source_sha():
                traffic, traffic_note = pmc["matvec_q4k_kernel"]["fetch_bytes_per_launch"], "profiles/%s (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction)" % PMC_FILE
            else:
                traffic_note = "profiles/%s was taken from other sources (%s, now %s): refused" % (PMC_FILE, pmc.get("source_sha"), source_sha())
        except Exception:
            pass
        if kp.launches:
            gbps = kp.bytes / kp.seconds / 1e9
            result["roofline"] = {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                                  "kernel": "matvec_q4k_kernel", "launches_per_frame": int(kp.launches // 3),
                                  "avg_launch_us": round(1e6 * kp.seconds / kp.launches, 3),
                                  "algorithmic_bytes_per_launch": int(kp.bytes // kp.launches)}
            # the same measurement split by instantiation family (rocprofv3 lists them as separate kernels): the large matrices stream, the small ones wait
            names = ("lds_staged_tiles: matvec_q4k_kernel<.., WS=0> (Temporal out_proj / linear_in / linear_out, text head)", "register_streaming: matvec_q4k_kernel<.., WS=1> (small matrices outside a chain)",
                     "merged: inproj_attn_kernel (Temporal in_proj tiles + RoPE + ring write + attention in one launch)")
            result["roofline_by_variant"] = {names[v]: {"achieved": round(kp.variant_bytes[v] / kp.variant_seconds[v] / 1e9, 1), "unit": "GB/s",
                                                        "frac": round(kp.variant_bytes[v] / kp.variant_seconds[v] / 1e9 / HBM_PEAK_GBPS, 4),
                                                        "launches_per_frame": int(kp.variant_launches[v] // 3), "avg_launch_us": round(1e6 * kp.variant_seconds[v] / kp.variant_launches[v], 3),
                                                        "algorithmic_bytes_per_launch": int(kp.variant_bytes[v] // kp.variant_launches[v])}
                                             for v in range(3) if kp.variant_launches[v]}
            if kp.chain_launches:
                # the chained Depth transformer (lm.h:446-553) as persistent launches: weights streamed once per launch, ~208 dependent phases inside it
                result["roofline_by_variant"]["persistent_chain: step programs (depth_nest_kernel: the Depth transformer; serial loop also the codec's two mimi_tr_kernel), %d phases per launch" % (kp.chain_phases // kp.chain_launches)] = {
                    "achieved": round(kp.chain_bytes / kp.chain_seconds / 1e9, 1), "unit": "GB/s", "frac": round(kp.chain_bytes / kp.chain_seconds / 1e9 / HBM_PEAK_GBPS, 4),
                    "launches_per_frame": int(kp.chain_launches // 3), "avg_launch_us": round(1e6 * kp.chain_seconds / kp.chain_launches, 3),
                    "algorithmic_bytes_per_launch": int(kp.chain_bytes // kp.chain_launches), "us_per_phase": round(1e6 * kp.chain_seconds / kp.chain_phases, 3)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the same frame loop on the host cores through the CPU oracle (a port of ggml's CPU semantics; the reference's
        # own ggml CPU backend cannot be built here or on the box). Bounded sample: ~10-30 s of CPU work.
        try:
            olib = load_oracle().load()
            L.ggml_backend_cpu_set_graph_compute(C.cast(olib.oracle_graph_compute, C.c_void_p))
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            def cpu_rate(cores, min_frames, budget_s):
                cbe = L.ggml_backend_init_by_type(pkg.DEV_CPU, None)
                L.ggml_backend_cpu_set_n_threads(cbe, cores)
                cm = L.moshi_hot_create(cbe, C.byref(cfg), 0)
                cpcm, cout = np.zeros(1920, np.float32), np.zeros(1920, np.float32)
                ctxt, caud = C.c_int32(), (C.c_int32 * 32)()
                # frame 0 builds the graphs and produces nothing (max_delay = 1): untimed warm-up
                L.moshi_hot_sts_frame(cm, cpcm.ctypes.data, C.byref(ctxt), caud, cout.ctypes.data)
                t0 = time.perf_counter()
                n = 0
                while n < min_frames or (time.perf_counter() - t0 < budget_s and n < 16):
                    L.moshi_hot_sts_frame(cm, cpcm.ctypes.data, C.byref(ctxt), caud, cout.ctypes.data)
                    n += 1
                    if time.perf_counter() - t0 > 3 * budget_s:
                        break
                cdt = time.perf_counter() - t0
                L.moshi_hot_free(cm)
                return n, cdt
            cores = max(1, min(avail, args.cpu_threads))
            n, cdt = cpu_rate(cores, args.cpu_frames, 10)
            result["cpu_baseline"] = {"value": round(n / cdt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
                                      "sample": f"{n} frames ({cdt:.1f} s) of the same moshika-7B q4_k sts loop after 1 warm-up frame, "
                                                "oracle/liboracle.so (scalar ggml-CPU semantics, OpenMP over mat-vec rows)"}
            if cores > 8 and args.cpu_threads == 32:
                # N = 8 threads sits beside the reference README's CPU figures (README.md:388-396)
                n8, cdt8 = cpu_rate(8, 1, 6)
                result["cpu_baseline"]["at_8_threads"] = {"value": round(n8 / cdt8, 4), "unit": "frames/s", "cores": 8, "sample": f"{n8} frames ({cdt8:.1f} s)"}
        except Exception as e:  # the baseline is auxiliary; never lose the GPU number over it
            result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}

    if rank == 0 and m is not None:
        # warm-up + timed + the phase / serial / roofline passes, on both model instances (profile summaries divide by it)
        result["frames_stepped_total"] = int(L.moshi_hot_offset(m)) + int(result.get("serial_loop", {}).get("frames_stepped_by_the_one_handle_model", 0))
    if rank == 0 and world == 1 and shard is None and not args.no_extras and args.model == "moshika" and args.quant == "q4_k" and not args.sampled and not args.context_fill:
        # not the headline: the same loop (a) with the reference's --bench sampling defaults, (b) from a nearly full ring, (c) at BASELINE configs[4]
        def quick(make_cfg, fill=0, steps=40):
            c2 = make_cfg()
            c2.codec_stream = int(pipelined)
            c2.chain_depth = 2 if pipelined else 0
            mm = L.moshi_hot_create(be, C.byref(c2), 0)
            if fill:
                L.moshi_hot_set_context_fill(mm, fill)
            if pipelined:
                L.moshi_hot_sts_pipeline_begin(mm, pcm.ctypes.data)
                step = lambda: L.moshi_hot_sts_pipeline_frame(mm, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)
            else:
                step = lambda: L.moshi_hot_sts_frame(mm, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)
            for _ in range(6):
                step()
            L.ggml_backend_synchronize(be)
            t1 = time.perf_counter()
            for _ in range(steps):
                step()
            L.ggml_backend_synchronize(be)
            r = steps / (time.perf_counter() - t1)
            L.moshi_hot_free(mm)
            return round(r, 2)

        def sampled_cfg():
            c2 = hot.moshika(L); c2.temp, c2.temp_text = 0.8, 0.7
            return c2

        def pp_cfg():
            c2 = hot.personaplex(L); c2.context = 2000
            return c2
        L.moshi_hot_free(m)
        m = None
        try:
            result["extras"] = {
                "sampled_temp_0.8_0.7_frames_per_s": quick(sampled_cfg),
                "context_fill_2800_of_3000_frames_per_s": quick(lambda: hot.moshika(L), fill=2800),
                "personaplex_ctx2000_fill_1900_frames_per_s": quick(pp_cfg, fill=1900),
            }
            # the shapes of BASELINE.json configs[1] / configs[2] (hot.tts_like / hot.stt_like), each in a child process of its own (their loops differ: text hook +
            # conditions + decode only; encode + VAD head only)
            import subprocess
            for name in ("tts_like", "stt_like"):
                cmd = [sys.executable, os.path.abspath(__file__), "--model", name, "--steps", "40", "--warmup", "6", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
                if args.serial:
                    cmd.append("--serial")
                child = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                line = [l for l in child.stdout.splitlines() if l.startswith("{")]
                result["extras"][name + "_frames_per_s"] = json.loads(line[-1])["value"] if child.returncode == 0 and line else None
            # the long-context line with its own roofline: this command from a ring holding 2 800 of 3 000 slots, in a child of its own (its phase pass and its
            # event-timed launches belong to THAT fill). A session longer than 4 minutes lives there. K / V bytes = 2 x layers x live slots x dim x 2 B per frame.
            cmd = [sys.executable, os.path.abspath(__file__), "--context-fill", "2800", "--steps", "40", "--warmup", "6", "--no-cpu-baseline", "--no-extras"]
            child = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            line = [l for l in child.stdout.splitlines() if l.startswith("{")]
            if child.returncode == 0 and line:
                d = json.loads(line[-1])
                merged = lambda r: next((v for k, v in r.get("roofline_by_variant", {}).items() if k.startswith("merged")), None)
                ml, ms = merged(d), merged(result)
                kvb = 2 * cfg.num_layers * d["n_fill_avg"] * cfg.dim * 2
                lc = {"frames_per_s": d["value"], "n_fill_avg": d["n_fill_avg"], "phase_us_temporal": d["phase_us"]["temporal"], "phase_us_temporal_at_the_headline_fill": result["phase_us"]["temporal"],
                      "kv_bytes_per_frame": int(kvb)}
                if ml and ms:
                    # inproj_attn_kernel carries the layer's attention as its tail: what the 28.3 MB in_proj costs alone is the headline fill's figure (a handful of live slots)
                    att = (ml["avg_launch_us"] - ms["avg_launch_us"]) * 1e-6
                    lc.update({"merged_inproj_attn_launch_us": ml["avg_launch_us"], "merged_inproj_attn_launch_us_at_the_headline_fill": ms["avg_launch_us"],
                               "kv_GBps_over_the_added_launch_time": round(kvb / cfg.num_layers / att / 1e9, 1) if att > 0 else None,
                               "kv_frac_of_hbm_peak": round(kvb / cfg.num_layers / att / 1e9 / HBM_PEAK_GBPS, 4) if att > 0 else None})
                dt_t = (d["phase_us"]["temporal"] - result["phase_us"]["temporal"]) * 1e-6
                lc["kv_GBps_over_the_added_temporal_time"] = round(kvb / dt_t / 1e9, 1) if dt_t > 0 else None
                result["extras"]["context_fill_2800"] = lc
        except Exception as e:
            result["extras"] = {"error": str(e)}
    if m is not None:
        L.moshi_hot_free(m)
    if rank == 0:
        emit(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
