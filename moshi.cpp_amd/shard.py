"""Depth-transformer codebook shard across ranks (reference: the chained Depth loop of src/moshi/models/lm.h:505-527 over
the per-step weight sets of lm_default.h:136-146,187-216).

Rank r holds the Depth weights of the steps k with k % world == r (include/moshi_hot.h, dep_shard_*). Per frame:
  1. the Temporal owner (rank 0) broadcasts `transformer_out` (F32[dim] + a "more frames" flag);
  2. for k = 0 .. dep_q-1: the owner of step k runs it (6 layers + head + sample) and broadcasts ONE message - its new K / V ring rows of all
     layers (2 x 6 x 1024 BF16 values, as the ring stores them: 24 KB) and the sampled token; every other rank writes them into its replica of the 8-slot ring / token vector;
  3. rank 0 reads the dep_q tokens.
The collectives are `torch.distributed.broadcast` on tensors that ALIAS the C side's message storage (device memory on the MI355X backend, so
RCCL moves them GPU to GPU over xGMI; host memory on the CPU device, for the gloo test): torch is transport plumbing only, the C-ABI carries no
torch type. There is no other data-path collective. The chain stays serial, so this is a strong-scaling (latency) experiment, reported as such.

EXPERIMENTAL on devices: the RCCL legs have only run under gloo (host memory, tests/test_depth_shard_cpu.py, tests/test_temporal_tp_cpu.py) and as a
one-GPU dry run - no multi-GPU box has been available to this project. Stream ordering on the device relies on ProcessGroupNCCL's wait() making the
current (external = the backend's) stream wait for the collective; MI355X_SHARD_HOST_SYNC=1 adds a host-side stream synchronisation after every
collective for a first bring-up on real hardware.
"""
import ctypes as C
import os

_HOST_SYNC = os.environ.get("MI355X_SHARD_HOST_SYNC", "0") not in ("", "0")


class _DeviceBlock:
    """exposes a raw device pointer to torch.as_tensor (zero copy)"""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def _alias(torch, ptr, n_floats, device):
    if device is None:                                                  # host memory (CPU device + gloo)
        buf = (C.c_float * n_floats).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.float32)
    return torch.as_tensor(_DeviceBlock(ptr, n_floats), device=device)  # device memory (MI355X backend + RCCL)


class DepthShard:
    def __init__(self, L, model, cfg, rank, world, dist, group=None, device=None, stream_ptr=None):
        self.L, self.m, self.cfg, self.rank, self.world, self.dist, self.group, self.torch = L, model, cfg, rank, world, dist, group, None
        self.msg = self.tout = self.stream = None
        self.msg_floats = C.c_int64()
        msg_ptr = L.moshi_hot_depth_shard_msg(model, C.byref(self.msg_floats))
        if world > 1:                                  # torch is transport plumbing only: a single rank never imports it
            import torch
            self.torch = torch
            n = C.c_int64()
            self.msg = _alias(torch, msg_ptr, self.msg_floats.value, device)
            self.tout = _alias(torch, L.moshi_hot_depth_shard_tout(model, C.byref(n)), n.value, device)
            # collectives are enqueued behind the backend's own stream (and the backend's next kernels behind them): no host synchronisation per hop
            self.stream = torch.cuda.ExternalStream(stream_ptr, device=device) if (device is not None and stream_ptr) else None
        self.hops = 0

    def _bcast(self, t, src):
        if self.world == 1:
            return
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                self.dist.broadcast(t, src=src, group=self.group)
            if _HOST_SYNC:
                self.stream.synchronize()
        else:
            self.dist.broadcast(t, src=src, group=self.group)
        self.hops += 1

    def _chain(self):
        L, m = self.L, self.m
        for k in range(self.cfg.dep_q):
            owner = k % self.world
            if owner == self.rank:
                L.moshi_hot_depth_shard_step(m, k)
            self._bcast(self.msg, owner)
            if owner != self.rank:
                L.moshi_hot_depth_shard_import(m, k)

    # ---- Temporal owner (rank 0): the hook moshi_hot_lm_step_n calls instead of the local chained Depth graph ---------------------------
    def depth_hook(self, user, text_token, audio):
        self.L.moshi_hot_depth_shard_begin_export(self.m, text_token, 1)
        self._bcast(self.tout, 0)
        self._chain()
        out = (C.c_int32 * self.cfg.dep_q)()
        self.L.moshi_hot_depth_shard_tokens(self.m, out, self.cfg.dep_q)
        for i in range(self.cfg.dep_q):
            audio[i] = out[i]

    def install(self):
        from . import hot
        self._cb = hot.DEPTH_HOOK(self.depth_hook)
        self.L.moshi_hot_set_depth_hook(self.m, C.cast(self._cb, C.c_void_p), None)

    def stop_workers(self):
        self.L.moshi_hot_depth_shard_begin_export(self.m, 0, 0)
        self._bcast(self.tout, 0)

    # ---- every other rank ----------------------------------------------------------------------------------------------------------------
    def serve(self):
        """returns the number of frames served when the owner says stop"""
        frames = 0
        while True:
            self._bcast(self.tout, 0)
            if not self.L.moshi_hot_depth_shard_begin_import(self.m):
                return frames
            self._chain()
            frames += 1


class TemporalTP:
    """Tensor-parallel Temporal stack (include/moshi_hot.h "tensor-parallel Temporal stack"): every rank holds its slices of each
    layer and runs the 2 L + 1 segment graphs; between segments the F32[dim] partial is summed over the ranks in place (`all_reduce` on a tensor
    aliasing the C side's message: RCCL over xGMI on devices, gloo on the host). Two all-reduces of 16 KB per layer."""

    def __init__(self, L, model, cfg, rank, world, dist, group=None, device=None, stream_ptr=None):
        self.L, self.m, self.cfg, self.rank, self.world, self.dist, self.group = L, model, cfg, rank, world, dist, group
        self.msg = self.stream = self.torch = None
        n = C.c_int64()
        ptr = L.moshi_hot_tp_msg(model, C.byref(n))
        if world > 1:
            import torch
            self.torch = torch
            self.msg = _alias(torch, ptr, n.value, device)
            self.stream = torch.cuda.ExternalStream(stream_ptr, device=device) if (device is not None and stream_ptr) else None
        self.reductions = 0

    def stack(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, np.float32)
        L, m = self.L, self.m
        L.moshi_hot_tp_begin(m, x.ctypes.data)
        last = 2 * self.cfg.num_layers
        for i in range(last + 1):
            L.moshi_hot_tp_segment(m, i)
            if i < last and self.world > 1:
                if self.stream is not None:
                    with self.torch.cuda.stream(self.stream):
                        self.dist.all_reduce(self.msg, group=self.group)
                    if _HOST_SYNC:
                        self.stream.synchronize()
                else:
                    self.dist.all_reduce(self.msg, group=self.group)
                self.reductions += 1
        out = np.zeros(self.cfg.dim, np.float32)
        L.moshi_hot_tp_end(m, out.ctypes.data)
        return out
