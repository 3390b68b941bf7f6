"""Run only the LM step (Temporal + Depth) for N frames: python lm_only.py [frames] [flags]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
pkg = load_package()
L = pkg.load()
from moshi_cpp_amd import hot  # noqa: E402

L.ggml_backend_load_all()
be = L.ggml_backend_init_by_name(b"ROCm0", None)
L.ggml_backend_mi355x_set_flags(be, flags)
cfg = hot.moshika(L)
cfg.enable_mimi_encoder = 0
cfg.enable_mimi_decoder = 0
m = L.moshi_hot_create(be, C.byref(cfg), 0)
codes = (C.c_int32 * 32)()
txt = C.c_int32()
out = (C.c_int32 * 32)()
t0 = None
for i in range(frames + 3):
    if i == 3:
        L.ggml_backend_synchronize(be); t0 = time.perf_counter()
    L.moshi_hot_lm_step(m, codes, C.byref(txt), out)
L.ggml_backend_synchronize(be)
print("lm us/frame", 1e6 * (time.perf_counter() - t0) / frames)
L.moshi_hot_free(m)
