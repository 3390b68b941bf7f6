import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, ctypes as C, ggml_util as gu
from ggml_util import *
r = np.random.default_rng(0 * 100 + 2)
K, M = 1024, 24
x = (r.standard_normal((M, K)) * np.exp(r.standard_normal((M, K)))).astype(np.float32)
x[1, 256:512] = 0.0; x[2, :256] = 2.5; x[3, 32:64] = 0.0; x[4] *= 1e-6; x[5, 256:288] = -1.0; x[6, ::7] = 0.0
g = gu.Graph("hip"); t = g.cast(g.input(x, F32), Q4_0); g.build([t]); g.alloc(); g.compute()
n = g.L.ggml_nbytes(t); raw = C.create_string_buffer(n); g.L.ggml_backend_tensor_get(t, raw, 0, n); a = np.frombuffer(raw.raw, np.uint8).copy(); g.free()
host = np.zeros(a.size, np.uint8); rb = host.size // M
for i in range(M):
    row = np.ascontiguousarray(x[i]); gu.lib().ggml_quantize_row(Q4_0, row.ctypes.data, host[i*rb:(i+1)*rb].ctypes.data, K)
A = a.reshape(-1, 18); H = host.reshape(-1, 18)
bad = np.nonzero((A != H).any(1))[0]
print(len(bad), "blocks differ of", A.shape[0], bad[:20], "rows", sorted(set((bad // 32).tolist())))
for b in bad[:4]:
    print(b, A[b], H[b], x.reshape(-1, 32)[b][:8])
