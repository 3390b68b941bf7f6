"""CPU: how far the reference arithmetic itself moves under a one-ulp perturbation at moshika's widths — the noise floor the full-config
device parity bars (tests/test_hip_frame.py::test_full_moshika_q4k_config_lm_steps_teacher_forced) are set against.

ggml's mul_mat quantises its activation to Q8_K (ggml-cpu.c, vec_dot_type of Q4_K) and the attention rows are stored as BF16
(moshi's cache type): both are step functions. Nudging ONE f32 norm vector of layer 0 by one ulp (a 1.2e-7 relative change, the size of
a float re-association) flips a few of those rounded values per layer; at dim 4096 that already moves the layer output by ~1e-3 of its
maximum. Two correct implementations with different summation orders (ggml's own AVX2 / AVX-512 / NEON kernels included) can agree no
better than this, so the device-vs-oracle bar at the full configuration cannot be the 1e-5 the 512-wide models allow."""
import ctypes as C

import numpy as np

import hot_util as hu

L = hu.L


def run(cfg, inputs, nudge):
    m = hu.Model("oracle", cfg, seed=0)
    if nudge:
        t = C.cast(L.moshi_hot_weight(m.m, b"lm.transformer.layers.0.norm1.alpha"), hu.pkg.TP)
        assert t
        w = np.zeros(cfg.dim, np.float32)
        L.ggml_backend_tensor_get(t, w.ctypes.data, 0, w.nbytes)
        w = np.nextafter(w, np.float32(np.inf), dtype=np.float32)
        L.ggml_backend_tensor_set(t, w.ctypes.data, 0, w.nbytes)
    out = []
    for ia in inputs:
        m.lm_step(ia)
        out.append((m.read("stack_out", cfg.dim).copy(), m.read("text_logits", cfg.text_card).copy()))
    m.free()
    return out


def test_one_ulp_nudge_moves_a_full_width_layer_by_a_rounding_flip():
    cfg = hu.hot.moshika(L)
    cfg.enable_mimi_encoder = cfg.enable_mimi_decoder = 0
    cfg.num_layers = 1                      # full widths, one Temporal layer: a few seconds of CPU
    rng = np.random.default_rng(5)
    inputs = [rng.integers(0, cfg.card, cfg.n_q - cfg.dep_q).tolist() for _ in range(2)]
    base, ulp = run(cfg, inputs, False), run(cfg, inputs, True)
    again = run(cfg, inputs, False)
    for (a, la), (b, lb), (c, lc) in zip(base, ulp, again):
        assert np.array_equal(a, c) and np.array_equal(la, lc)          # the oracle is deterministic ...
        e = hu.rel_err(a, b)
        assert 1e-5 < e < 2e-2, f"stack_out moved by {e:.2e}"            # ... and a 1.2e-7 nudge is amplified >100x by the quantisers (measured 1.6e-3)
        assert hu.rel_err(la, lb) < 5e-2
