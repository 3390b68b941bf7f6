"""Run only the Mimi encoder or decoder graph for N frames (for rocprofv3 kernel traces): python mimi_only.py enc|dec [frames] [flags]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

which = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 2
pkg = load_package()
L = pkg.load()
from moshi_cpp_amd import hot  # noqa: E402

L.ggml_backend_load_all()
be = L.ggml_backend_init_by_name(b"ROCm0", None)
L.ggml_backend_mi355x_set_flags(be, flags)
cfg = hot.moshika(L)
cfg.enable_lm = 0
m = L.moshi_hot_create(be, C.byref(cfg), 0)
pcm = np.zeros(1920, np.float32)
codes = (C.c_int32 * 32)()
import time
t0 = None
for i in range(frames + 3):
    if i == 3:
        L.ggml_backend_synchronize(be); t0 = time.perf_counter()
    if which == "enc":
        L.moshi_hot_mimi_encode(m, pcm.ctypes.data, codes)
    else:
        L.moshi_hot_mimi_decode(m, codes, pcm.ctypes.data)
L.ggml_backend_synchronize(be)
print(which, "us/frame", 1e6 * (time.perf_counter() - t0) / frames)
L.moshi_hot_free(m)
