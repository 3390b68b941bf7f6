"""Helpers to drive the host-side hot-path driver (include/moshi_hot.h) on a backend."""
import ctypes as C

import numpy as np

import ggml_util as gu

L = gu.lib()
pkg = gu.pkg
from moshi_cpp_amd import hot  # noqa: E402


def make_backend(kind, threads=8):
    L.ggml_backend_load_all()
    if kind == "hip":
        be = L.ggml_backend_init_by_type(pkg.DEV_GPU, None)
        if not be:
            raise RuntimeError("no MI355X device (the hot path has no CPU fallback)")
        return be
    gu.attach_oracle()
    be = L.ggml_backend_init_by_type(pkg.DEV_CPU, None)
    L.ggml_backend_cpu_set_n_threads(be, threads)
    return be


class Model:
    def __init__(self, kind, cfg, seed=0, flags=0):
        self.cfg = cfg
        self.be = make_backend(kind)
        if kind == "hip" and flags:
            L.ggml_backend_mi355x_set_flags(self.be, flags)
        self.kind = kind
        self.m = L.moshi_hot_create(self.be, C.byref(cfg), seed)
        assert self.m

    def sts_frame(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.float32)
        txt = C.c_int32(-7)
        aud = (C.c_int32 * 32)()
        out = np.zeros(1920, np.float32)
        r = L.moshi_hot_sts_frame(self.m, pcm.ctypes.data, C.byref(txt), aud, out.ctypes.data)
        return r, txt.value, list(aud)[:self.cfg.dep_q], out

    def host_ring(self):
        """the host-side delay ring of the LM generator, [rows, n_q + 1] int32"""
        n = L.moshi_hot_host_ring(self.m, None, 0)
        buf = np.zeros(n, np.int32)
        assert L.moshi_hot_host_ring(self.m, buf.ctypes.data, n) == n
        return buf.reshape(-1, self.cfg.n_q + 1)

    def sts_pipeline(self, frames):
        """the software-pipelined loop over `frames` (moshi_hot_sts_pipeline_*): -> per frame (produced, text, audio tokens, pcm) like sts_frame"""
        frames = [np.ascontiguousarray(f, np.float32) for f in frames]
        res = [[0, -7, [0] * self.cfg.dep_q] for _ in frames]
        pcm_of = {}
        L.moshi_hot_sts_pipeline_begin(self.m, frames[0].ctypes.data)
        lag = 0
        for k in range(len(frames)):
            txt = C.c_int32(-7)
            aud = (C.c_int32 * 32)()
            prev = np.zeros(1920, np.float32)
            nxt = frames[k + 1].ctypes.data if k + 1 < len(frames) else None
            r = L.moshi_hot_sts_pipeline_frame(self.m, nxt, C.byref(txt), aud, prev.ctypes.data)
            lag = 1 if r & 4 else 0          # run-ahead: the tokens are frame k - 1's
            if r & 2:
                pcm_of[k - 1] = prev
            if k - lag >= 0 and (r & 1 or not lag):
                res[k - lag] = [r & 1, txt.value, list(aud)[:self.cfg.dep_q]]
        last = np.zeros(1920, np.float32)
        txt = C.c_int32(-7)
        aud = (C.c_int32 * 32)()
        r = L.moshi_hot_sts_pipeline_end(self.m, C.byref(txt), aud, last.ctypes.data)
        if r & 1:
            res[len(frames) - 1] = [1, txt.value, list(aud)[:self.cfg.dep_q]]
        if r & 2:
            pcm_of[len(frames) - 1] = last
        return [tuple(r) + (pcm_of.get(k, np.zeros(1920, np.float32)),) for k, r in enumerate(res)]

    def lm_step(self, in_audio):
        ia = (C.c_int32 * 32)(*in_audio)
        txt = C.c_int32(-7)
        aud = (C.c_int32 * 32)()
        r = L.moshi_hot_lm_step(self.m, ia, C.byref(txt), aud)
        return r, txt.value, list(aud)[:self.cfg.io_dep_q]

    def lm_step_n(self, tokens, vad=False):
        """moshi_lmgen_step in full: len(tokens) == n_q + 1 is a "provided" (prompt) frame; vad=True also returns the VAD head's probability."""
        ia = (C.c_int32 * 64)(*tokens)
        txt = C.c_int32(-7)
        aud = (C.c_int32 * 64)()
        v = C.c_float(-1.0)
        r = L.moshi_hot_lm_step_n(self.m, ia, len(tokens), C.byref(txt), aud, C.byref(v) if vad else None)
        out = (r, txt.value, list(aud)[:self.cfg.io_dep_q])
        return out + (v.value,) if vad else out

    def lm_step_embedding(self, emb):
        emb = np.ascontiguousarray(emb, np.float32)
        assert emb.size == self.cfg.dim
        L.moshi_hot_lm_step_embedding(self.m, emb.ctypes.data)

    def prefill(self, frames, chunk=0):
        """frames: list of (n_q + 1)-token lists (text first)"""
        flat = np.ascontiguousarray(np.array(frames, np.int32).reshape(-1))
        L.moshi_hot_prefill(self.m, flat.ctypes.data, len(frames), chunk)

    def system_prompts(self, text_prompt, batched=False, chunk=0):
        tp = (C.c_int32 * max(1, len(text_prompt)))(*text_prompt)
        if batched:
            L.moshi_hot_personaplex_system_prompts_batched(self.m, tp, len(text_prompt), chunk)
        else:
            L.moshi_hot_personaplex_system_prompts(self.m, tp, len(text_prompt))

    def last_raw(self):
        txt = C.c_int32()
        aud = (C.c_int32 * 32)()
        L.moshi_hot_last_raw_tokens(self.m, C.byref(txt), aud)
        return txt.value, list(aud)[:self.cfg.dep_q]

    def rings(self, which=0):
        """every K / V ring of a stack as raw bytes: [(layer, kv, bytes)]"""
        out = []
        layer = 0
        while True:
            n = L.moshi_hot_ring_bytes(self.m, which, layer, 0, None, 0, 0)
            if n < 0:
                return out
            for kv in (0, 1):
                buf = np.zeros(n, np.uint8)
                assert L.moshi_hot_ring_bytes(self.m, which, layer, kv, buf.ctypes.data, n, 0) == n
                out.append((layer, kv, buf))
            layer += 1

    def set_rings(self, rings, which=0):
        for layer, kv, buf in rings:
            assert L.moshi_hot_ring_bytes(self.m, which, layer, kv, buf.ctypes.data, buf.nbytes, 1) == buf.nbytes

    def force_last(self, txt, aud):
        L.moshi_hot_force_last(self.m, txt, (C.c_int32 * 32)(*aud))

    def mimi_decode(self, codes):
        cc = (C.c_int32 * 32)(*codes)
        out = np.zeros(1920, np.float32)
        L.moshi_hot_mimi_decode(self.m, cc, out.ctypes.data)
        return out

    def mimi_encode(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.float32)
        codes = (C.c_int32 * 32)()
        L.moshi_hot_mimi_encode(self.m, pcm.ctypes.data, codes)
        return list(codes)[:self.cfg.mimi_n_q]

    def read(self, what, n):
        out = np.zeros(n, np.float32)
        assert L.moshi_hot_read_last(self.m, what.encode(), out.ctypes.data, n) == 0
        return out

    def stats(self):
        s = pkg.Stats()
        L.ggml_backend_mi355x_get_stats(self.be, C.byref(s))
        return s

    def free(self):
        L.moshi_hot_free(self.m)
        L.ggml_backend_free(self.be)


def rel_err(ref, got):
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    return float(np.max(np.abs(ref - got))) / scale


def set_conditions(m, cfg, seed=4):
    """tts conditions: deterministic sum F32[dim] and cross F32[dim, cross_len]; returns them"""
    rng = np.random.default_rng(seed)
    s = (rng.standard_normal(cfg.dim) * 0.1).astype(np.float32) if cfg.condition_sum else None
    x = rng.standard_normal((cfg.cross_len, cfg.dim)).astype(np.float32) if cfg.cross_attention else None
    L.moshi_hot_set_conditions(m.m, s.ctypes.data if s is not None else None, x.ctypes.data if x is not None else None)
    return s, x


def set_text_hook(m, fn):
    """fn(offset, sampled) -> text token; keeps the ctypes thunk alive on the model"""
    m._hook = hot.TEXT_HOOK(lambda user, offset, sampled: fn(offset, sampled))
    L.moshi_hot_set_text_hook(m.m, C.cast(m._hook, C.c_void_p), None)


# ---- the .mimi container of tools/mimi-encode.cpp:171-195 / mimi-decode.cpp:139-187: "MIMI", int32 n_q, then frames of n_q int16 codes ----
def write_mimi(path, frames):
    import struct
    n_q = len(frames[0])
    with open(path, "wb") as f:
        f.write(b"MIMI")
        f.write(struct.pack("<i", n_q))
        for fr in frames:
            assert len(fr) == n_q
            f.write(np.asarray(fr, np.int16).tobytes())


def read_mimi(path):
    import struct
    with open(path, "rb") as f:
        if f.read(4) != b"MIMI":
            raise ValueError("invalid mimi input file")
        (n_q,) = struct.unpack("<i", f.read(4))
        if n_q < 1 or n_q > 32:
            raise ValueError("n_q in mimi file out of range %d" % n_q)
        data = f.read()
        raw = np.frombuffer(data[:len(data) // 2 * 2], np.int16)
    n = raw.size // n_q                      # a trailing partial frame is dropped, as fread(..., n_q*2, 1, f) == 1 does
    return n_q, raw[:n * n_q].reshape(n, n_q).astype(np.int32)


def decode_mimi_file(kind, path, cfg_fn):
    """the main loop of tools/mimi-decode.cpp:186-196: one mimi_decode per frame of codes"""
    n_q, frames = read_mimi(path)
    cfg = cfg_fn(n_q)
    m = Model(kind, cfg)
    pcm = [m.mimi_decode(fr.tolist()) for fr in frames]
    m.free()
    return n_q, np.concatenate(pcm) if pcm else np.zeros(0, np.float32)
