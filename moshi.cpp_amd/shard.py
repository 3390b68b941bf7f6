"""Depth-transformer codebook shard across ranks (reference: the chained Depth loop of src/moshi/models/lm.h:505-527 over
the per-step weight sets of lm_default.h:136-146,187-216).

Rank r holds the Depth weights of the steps k with k % world == r (include/moshi_hot.h, dep_shard_*). Per frame:
  1. the Temporal owner (rank 0) broadcasts `transformer_out` (F32[dim] + a "more frames" flag);
  2. for k = 0 .. dep_q-1: the owner of step k runs it (6 layers + head + sample) and broadcasts ONE message - its new K / V ring rows of all
     layers (2 x 6 x 1024 BF16 values, as the ring stores them: 24 KB) and the sampled token; every other rank writes them into its replica of the 8-slot ring / token vector;
  3. rank 0 reads the dep_q tokens.
The whole frame - the hop loop included - runs BEHIND THE C-ABI (moshi_hot_depth_shard_install / _serve, include/moshi_hot.h): no interpreter on the
critical path. On devices the broadcasts are ncclBroadcast calls the harness makes itself (librccl.so opened at run time) on the backend's own HIP stream,
i.e. stream-ordered between the step's kernels and the import's - no host synchronisation, no event, no torch stream semantics involved. This module
only carries RCCL's 128-byte unique id from rank 0 to the other ranks (over the caller's torch.distributed control group) and starts the calls.
On the CPU device (tests/test_depth_shard_cpu.py) the transport is a function this module supplies: torch.distributed.broadcast over gloo on a tensor
aliasing the message's host storage - the same C loop, a different wire.
There is no other data-path collective. The chain stays serial, so this is a strong-scaling (latency) experiment, reported as such. The RCCL leg has run
with a world of ONE rank on the test box (tests/test_hip_frame.py); no multi-GPU box has been available to this project.
"""
import ctypes as C


def _rccl_bootstrap(L, model, rank, world, dist, group, device):
    """RCCL from inside the harness: rank 0 makes the 128-byte unique id, the caller's torch.distributed control group carries it to the other ranks, every
    rank then joins the communicator behind the C-ABI (moshi_hot_depth_shard_rccl_init makes the backend's device current first). A C++ caller does the same
    with its own wire (INTEGRATION.md "Multi-GPU bootstrap without torch")."""
    import torch
    idbuf = C.create_string_buffer(128)
    if rank == 0:
        rc = L.moshi_hot_depth_shard_rccl_unique_id(idbuf)
        if rc != 0:
            raise RuntimeError("ncclGetUniqueId failed (%d)" % rc)
    if world > 1:
        t = torch.frombuffer(bytearray(idbuf.raw), dtype=torch.uint8).clone()
        if dist.get_backend(group) != "gloo":
            t = t.to(device)
        dist.broadcast(t, src=0, group=group)
        idbuf = C.create_string_buffer(bytes(t.cpu().numpy().tobytes()), 128)
    rc = L.moshi_hot_depth_shard_rccl_init(model, rank, world, idbuf)
    if rc != 0:
        raise RuntimeError("ncclCommInitRank failed (%d)" % rc)


class DepthShard:
    def __init__(self, L, model, cfg, rank, world, dist, group=None, device=None, stream_ptr=None, staged_device=None, backend=None):
        """device: the torch device of this rank's GPU -> RCCL from inside the harness. device=None: the caller-supplied transport; the messages are host
        memory (CPU device) unless staged_device / backend are given (a GPU backend under a gloo control plane: single-GPU dry runs - the message is
        aliased as a device tensor and the backend is synchronised around every hop)."""
        from . import hot
        self.L, self.m, self.cfg, self.rank, self.world, self.dist, self.group = L, model, cfg, rank, world, dist, group
        self.msg_floats = C.c_int64()
        L.moshi_hot_depth_shard_msg(model, C.byref(self.msg_floats))
        self.transport = "none"
        self._cb = None
        if world > 1 or device is not None:   # (a single rank WITH a device still sets RCCL up: every hop is then a real, if trivial, ncclBroadcast)
            if device is not None:
                _rccl_bootstrap(L, model, rank, world, dist, group, device)
                self.transport = "rccl (ncclBroadcast called by the harness on the backend's stream)"
            else:
                import torch

                def bcast(user, data, nbytes, root):
                    if staged_device is not None:
                        L.ggml_backend_synchronize(backend)
                        dist.broadcast(_alias(torch, data, nbytes // 4, staged_device), src=root, group=group)
                        torch.cuda.synchronize()
                        return
                    buf = (C.c_uint8 * nbytes).from_address(data)
                    dist.broadcast(torch.frombuffer(buf, dtype=torch.uint8), src=root, group=group)
                self._cb = hot.BCAST_FN(bcast)
                L.moshi_hot_depth_shard_set_transport(model, C.cast(self._cb, C.c_void_p), None)
                self.transport = "caller-supplied function (torch.distributed over host memory)"

    def install(self):
        """rank 0: the sharded frame becomes the Depth half of every LM step"""
        self.L.moshi_hot_depth_shard_install(self.m)

    def stop_workers(self):
        self.L.moshi_hot_depth_shard_stop(self.m)

    def serve(self):
        """every other rank: returns the number of frames served when the owner says stop"""
        return int(self.L.moshi_hot_depth_shard_serve(self.m))

    @property
    def hops(self):
        return int(self.L.moshi_hot_depth_shard_hops(self.m))

    def hop(self, root):
        """one broadcast of the step message (latency probe; collective)"""
        self.L.moshi_hot_depth_shard_broadcast(self.m, 0, root)


class _DeviceBlock:
    """exposes a raw device pointer to torch.as_tensor (zero copy)"""

    def __init__(self, ptr, n_floats):
        self.__cuda_array_interface__ = {"shape": (n_floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def _alias(torch, ptr, n_floats, device):
    if device is None:                                                  # host memory (CPU device + gloo)
        buf = (C.c_float * n_floats).from_address(ptr)
        return torch.frombuffer(buf, dtype=torch.float32)
    return torch.as_tensor(_DeviceBlock(ptr, n_floats), device=device)  # device memory (single-GPU dry runs under a gloo control plane)


class TemporalTP:
    """Tensor-parallel Temporal stack (include/moshi_hot.h "tensor-parallel Temporal stack"): every rank holds its slices of each
    layer and runs the 2 L + 1 segment graphs; between segments the F32[dim] partial is summed over the ranks in place. The loop and the all-reduces run
    behind the C-ABI (moshi_hot_tp_stack): ncclAllReduce on the backend's stream on devices (the model's communicator, set up by DepthShard / by
    moshi_hot_depth_shard_rccl_init), or - host memory, the CPU device - a function this class supplies (torch.distributed.all_reduce over gloo)."""

    def __init__(self, L, model, cfg, rank, world, dist, group=None, device=None, stream_ptr=None, staged_device=None, backend=None, have_comm=False):
        """device: the torch device of this rank's GPU -> ncclAllReduce from inside the harness on the model's communicator (set up here unless have_comm says a
        DepthShard of the same model already did). device=None: a caller-supplied transport - host memory on the CPU device, or (staged_device / backend given: a
        GPU backend under a gloo control plane, single-GPU dry runs) the partial aliased as a device tensor with the backend synchronised around every reduction."""
        from . import hot
        self.L, self.m, self.cfg, self.rank, self.world = L, model, cfg, rank, world
        self._cb = None
        self.transport = "none"
        if device is not None:
            if not have_comm:
                _rccl_bootstrap(L, model, rank, world, dist, group, device)
            self.transport = "rccl (ncclAllReduce called by the harness on the backend's stream)"
        elif world > 1:
            import torch

            def allreduce(user, data, n):
                if staged_device is not None:
                    L.ggml_backend_synchronize(backend)
                    dist.all_reduce(_alias(torch, data, n, staged_device), group=group)
                    torch.cuda.synchronize()
                    return
                dist.all_reduce(_alias(torch, data, n, None), group=group)
            self._cb = hot.ALLREDUCE_FN(allreduce)
            L.moshi_hot_tp_set_transport(model, C.cast(self._cb, C.c_void_p), None)

            def bcast(user, data, nbytes, root):   # frame mode: the stack input travels from rank 0 to the others
                if staged_device is not None:
                    L.ggml_backend_synchronize(backend)
                    dist.broadcast(_alias(torch, data, nbytes // 4, staged_device), src=root, group=group)
                    torch.cuda.synchronize()
                    return
                buf = (C.c_uint8 * nbytes).from_address(data)
                dist.broadcast(torch.frombuffer(buf, dtype=torch.uint8), src=root, group=group)
            self._bcb = hot.BCAST_FN(bcast)
            L.moshi_hot_depth_shard_set_transport(model, C.cast(self._bcb, C.c_void_p), None)
            self.transport = "caller-supplied function (torch.distributed)"

    def install(self):
        """rank 0: the tensor-parallel stack becomes the Temporal half of every LM step (include/moshi_hot.h "FRAME mode")"""
        self.L.moshi_hot_tp_install(self.m)

    def serve(self):
        """every other rank: frames served until rank 0 stops"""
        return int(self.L.moshi_hot_tp_serve(self.m))

    def stop_workers(self):
        self.L.moshi_hot_tp_stop(self.m)

    @property
    def reductions(self):
        return int(self.L.moshi_hot_tp_reductions(self.m))

    def stack(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(self.cfg.dim, np.float32)
        self.L.moshi_hot_tp_stack(self.m, x.ctypes.data, out.ctypes.data)
        return out
